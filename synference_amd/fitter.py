"""``SBI_Fitter`` surface of the reference for the amortised-posterior flow path, HIP backend.

Same method names, argument meaning and error behaviour as ref: src/synference/sbi_runner.py for
  run_single_sbi (4392-4435), create_priors (3442-3450), split_dataset (3407-3440),
  sample_posterior (6350-6474), log_prob (7162-7198), fit_catalogue sampling + quantile section
  (2948-2989, 3230-3282), evaluate_model's flow-derived metrics (6484-6735).
What is NOT here (out of scope): library generation, feature engineering from
raw fluxes, noise models, Optuna search, MDN / likelihood / ratio engines, MCMC / VI samplers,
plotting.  Asking for them raises ``ValueError`` rather than falling through to another backend.

The per-galaxy Python loops of the reference (6438-6442, 7193-7196) collapse into one catalogue
call on the GPU; return types and shapes are the reference's ((N,S,D) float64 ndarray; (N,) float64).
"""
from __future__ import annotations

import logging
import os
import time
import warnings
from typing import Callable, List, Optional, Union

import numpy as np
import torch

from .estimator import SUPPORTED_MODELS, load_nde_hip
from .priors import CustomIndependentUniform, prior_from_parameters
from .runner import HIPRunner, NumpyLoader

logger = logging.getLogger("synference_amd")

# the reference's default model directory is f"{code_path}/models/" with code_path = the directory above the package
# (sbi_runner.py:87-89, 4413)
DEFAULT_MODEL_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "models") + "/"


def _warn_unknown(fn: str, unknown: dict) -> None:
    """Keywords outside the mirrored signature are named in a warning instead of vanishing."""
    if unknown:
        warnings.warn(f"{fn}: unknown keyword argument(s) {sorted(unknown)} ignored", stacklevel=3)


class StandardScaler:
    """The two-method subset of sklearn.preprocessing.StandardScaler that ``prior_method="manual"`` uses (``feature_scalar`` /
    ``target_scalar`` defaults of run_single_sbi, sbi_runner.py:4416-4417, 4664-4671), with sklearn's attribute names
    (``mean_``, ``scale_``: population std, zeros replaced by 1) so that the product does not import scikit-learn; any class with
    fit / transform / inverse_transform can be passed instead."""

    def fit(self, X, y=None):
        X = np.asarray(X, dtype=np.float64)
        self.mean_ = X.mean(axis=0)
        self.scale_ = X.std(axis=0)
        self.scale_[self.scale_ == 0.0] = 1.0
        return self

    def transform(self, X):
        return (np.asarray(X, dtype=np.float64) - self.mean_) / self.scale_

    def fit_transform(self, X, y=None):
        return self.fit(X).transform(X)

    def inverse_transform(self, X):
        return np.asarray(X, dtype=np.float64) * self.scale_ + self.mean_


class SBI_Fitter:
    device = "cuda"

    @property
    def _timestamp(self):
        """Current date and time as a string (ref: sbi_runner.py:7640-7643)."""
        return time.strftime("%Y%m%d_%H%M%S")

    def __init__(self, name: str, parameter_names: list, raw_observation_names: list = None,
                 raw_observation_grid: np.ndarray = None, parameter_array: np.ndarray = None,
                 parameter_units: list = None, raw_observation_units: list = None, simulator: callable = None,
                 feature_array: np.ndarray = None, feature_names: list = None, feature_units: list = None,
                 library_path: str = None, supplementary_parameters: np.ndarray = None,
                 supplementary_parameter_names: list = None, supplementary_parameter_units: list = None,
                 device: str = "cuda", observation_type: str = "photometry") -> None:
        if simulator is not None:
            raise ValueError("online simulators are outside the accelerated path: the HIP backend trains on a library "
                             "(init_from_hdf5) or on a feature array + parameter array")
        self.library_path = library_path
        self.raw_observation_grid = None if raw_observation_grid is None else np.asarray(raw_observation_grid)
        self.raw_observation_units = raw_observation_units
        self.supplementary_parameters = supplementary_parameters
        self.supplementary_parameter_names = supplementary_parameter_names
        self.supplementary_parameter_units = supplementary_parameter_units
        self.name = name
        self.parameter_names = list(parameter_names)
        self.fitted_parameter_names = list(parameter_names)
        self.simple_fitted_parameter_names = [str(p).split("/")[-1] for p in parameter_names]
        self.fitted_parameter_units = parameter_units
        self.raw_observation_names = raw_observation_names
        self.feature_names = feature_names if feature_names is not None else raw_observation_names
        self.feature_units = feature_units
        self.device = device
        self.observation_type = observation_type
        self.feature_array = None
        self.fitted_parameter_array = None
        self.has_features = False
        if feature_array is not None:
            # reference: feature_array.astype(np.float32), (N, C) (sbi_runner.py:2150)
            self.feature_array = np.ascontiguousarray(np.asarray(feature_array).astype(np.float32))
            self.has_features = True
        self.parameter_array = None
        self.parameter_units = parameter_units
        if parameter_array is not None:
            self.parameter_array = np.asarray(parameter_array)
            self.fitted_parameter_array = np.asarray(parameter_array)
            if self.fitted_parameter_array.shape[1] != len(self.parameter_names):
                raise ValueError("parameter_array must be (N, len(parameter_names))")
        self.has_simulator = False
        self._feature_scalar = None      # prior_method="manual" (sbi_runner.py:287-288)
        self._target_scalar = None
        self.posteriors = None
        self.stats = None
        self._prior = None
        self._train_indices = self._test_indices = None
        self._X_train = self._y_train = self._X_test = self._y_test = None
        self.fitted_model_name = None

    # ---------------------------------------------------------------------------------------
    @classmethod
    def init_from_hdf5(cls, model_name: str, hdf5_path: str, return_output: bool = False, **kwargs):
        """ref: sbi_runner.py:309-405 -- a fitter on top of a library file (``Grid/Photometry`` (C,N) in nJy,
        ``Grid/Parameters`` (D,N), root attributes ``FilterCodes`` / ``ParameterNames`` / units), read by
        ``synference_amd.library.load_library_from_hdf5`` (no h5py needed)."""
        from .library import load_library_from_hdf5
        output = load_library_from_hdf5(hdf5_path)
        if return_output:
            return output
        if "photometry" in output:
            grid, otype = output["photometry"], "photometry"
        elif "spectra" in output:
            grid, otype = output["spectra"], "spectra"
        else:
            raise ValueError("HDF5 file must contain 'photometry' or 'spectra' data.")
        return cls(name=model_name, raw_observation_grid=grid, raw_observation_names=list(output["filter_codes"]),
                   parameter_array=output["parameters"].T, parameter_names=list(output["parameter_names"]),
                   parameter_units=None if output["parameter_units"] is None else list(output["parameter_units"]),
                   raw_observation_units=output["photometry_units"], library_path=hdf5_path,
                   supplementary_parameters=output.get("supplementary_parameters"),
                   supplementary_parameter_names=list(output.get("supplementary_parameter_names", [])),
                   supplementary_parameter_units=list(output.get("supplementary_parameter_units", [])),
                   observation_type=otype, **kwargs)

    def create_feature_array(self, flux_units: str = "AB", extra_features: list = None, **kwargs):
        """ref: sbi_runner.py:1065-1094 -- the simple wrapper: no noise, every filter of the library."""
        if self.observation_type == "photometry":
            return self.create_feature_array_from_raw_photometry(normed_flux_units=flux_units, extra_features=extra_features,
                                                                 **kwargs)
        if self.observation_type == "spectra":
            raise ValueError("feature arrays from raw spectra are outside the HIP path")
        raise ValueError(f"Observation type {self.observation_type} not supported. Please use 'photometry' or 'spectra'.")

    def create_feature_array_from_raw_photometry(self, normalize_method: Optional[str] = None, extra_features: list = None,
                                                 normed_flux_units: str = "AB", normalization_unit: str = "AB",
                                                 verbose: bool = True, scatter_fluxes: Union[int, bool] = False,
                                                 empirical_noise_models=None, depths=None,
                                                 include_errors_in_feature_array: bool = False,
                                                 min_flux_pc_error: float = 0.0, simulate_missing_fluxes: bool = False,
                                                 norm_mag_limit: float = 50.0, remove_nan_inf: bool = True,
                                                 parameters_to_remove: Optional[list] = None,
                                                 photometry_to_remove: Optional[list] = None,
                                                 parameters_to_add: Optional[list] = None, drop_dropouts: bool = False,
                                                 drop_dropout_fraction: float = 1.0, max_rows: int = -1,
                                                 parameter_transformations: Optional[dict] = None,
                                                 asinh_softening_parameters=None, seed: int = 0, **unknown):
        """The AB-magnitude branch of the reference's feature engineering (ref: sbi_runner.py:1429-2222) with the
        arithmetic on the device: the library's (C, N) fluxes in nJy become the (N', F) float32 feature array.

        Same argument names and meaning as the reference for what is supported:
          * ``photometry_to_remove`` (1566-1589), ``scatter_fluxes`` + ``depths`` (array, or dict by filter name; same
            unit as the grid) + ``min_flux_pc_error`` (1629-1655, ``_apply_depths`` 580-691; ``sf_scatter_depths``),
            ``include_errors_in_feature_array`` (magnitude errors 2.5 e / (ln 10 f), 1698-1702, columns ``unc_<filter>``);
          * nJy -> AB with negative fluxes and anything fainter set to ``norm_mag_limit`` (1704-1716, 1927-1932;
            ``sf_flux_to_abmag``);
          * ``normalize_method`` = a filter name (1783-1834): that band is removed, the others become colours relative
            to it, and the band itself -- the UNSCATTERED library flux, as ``normalization_unit`` "AB" or "log10 nJy"
            -- is the last column ``norm_<filter>_<unit>`` (2002-2027);
          * ``remove_nan_inf``, ``drop_dropouts`` / ``drop_dropout_fraction``, ``max_rows`` (2084-2140), and the
            parameter bookkeeping of ``update_parameter_array`` (476-578: ``parameters_to_remove``,
            ``parameters_to_add`` from the supplementary parameters, repetition per scatter copy, deleted rows,
            ``parameter_transformations``).
          * ``normed_flux_units="asinh"`` (1591-1627, 1718-1732; utils.py:647-704; ``sf_flux_to_asinh``): softening
            ``asinh_softening_parameters`` per filter in the grid's unit -- an array, a dict by filter name, or
            ``"SNR_<k>"`` = k x depth / 5 when scattering with depths; no magnitude-limit clip there, as in the reference.
        Outside the accelerated path (``ValueError``): other flux units, extra feature expressions, empirical noise
        models, simulated missing fluxes, normalisation by a supplementary parameter or of asinh magnitudes.  ``seed``
        replaces numpy's global generator for the scatter noise and the ``max_rows`` draw."""
        _warn_unknown("create_feature_array_from_raw_photometry", unknown)
        if self.raw_observation_grid is None:
            raise ValueError("no raw observation grid: build the fitter with init_from_hdf5 or pass feature_array")
        if extra_features or normed_flux_units not in ("AB", "asinh") or empirical_noise_models is not None or simulate_missing_fluxes:
            raise ValueError("only normed_flux_units='AB' / 'asinh' without extra features, empirical noise models or "
                             "simulated missing fluxes is on the HIP path")
        asinh = normed_flux_units == "asinh"
        if asinh and normalize_method is not None:
            raise ValueError("normalisation of asinh magnitudes is outside the HIP path")
        if asinh:
            assert asinh_softening_parameters is not None, "asinh_softening_parameters must be provided for asinh normalization."
        if not torch.cuda.is_available():
            raise RuntimeError("create_feature_array_from_raw_photometry runs on the GPU (no CPU fallback)")
        from .features import flux_to_abmag, flux_to_asinh, scatter_depths
        names = [str(n_) for n_ in self.raw_observation_names]
        grid = np.asarray(self.raw_observation_grid)                                    # (C, N)
        photometry_to_remove = list(photometry_to_remove or [])
        if photometry_to_remove:
            rm = [i for i, n_ in enumerate(names) if n_ in photometry_to_remove]
            if not rm:
                raise ValueError(f"No matching photometry filters found in the raw photometry names: {photometry_to_remove}")
            grid = np.delete(grid, rm, axis=0)
            names = [n_ for i, n_ in enumerate(names) if i not in rm]
            if not names:
                raise ValueError("No photometry filters left after removing the specified ones.")
        n_sc = int(scatter_fluxes) if scatter_fluxes else 0
        flux = torch.as_tensor(np.ascontiguousarray(grid.T), dtype=torch.float32).cuda()  # (N, C) on the device
        err = None
        if n_sc:
            if depths is None:
                raise ValueError("If scattering fluxes, depths or empirical noise models must be provided.")
            if isinstance(depths, dict):
                depths = np.asarray([depths[n_] for n_ in names], dtype=np.float32)
            self.phot_depths, self.min_flux_pc_error = depths, min_flux_pc_error
            flux, err = scatter_depths(flux, depths, n_sc, 5.0, min_flux_pc_error, seed=seed, return_errors=True)
        if asinh:
            if isinstance(asinh_softening_parameters, str):
                assert asinh_softening_parameters.startswith("SNR_"), "If a string, asinh_softening_parameters must start with 'SNR_'."
                assert n_sc and depths is not None, ("If setting asinh_softening_parameters from noise models, "
                                                     "depths or empirical_noise_models must be provided.")
                fb = float(asinh_softening_parameters.split("_")[-1]) * np.asarray(depths, dtype=np.float64).reshape(-1) / 5.0
            elif isinstance(asinh_softening_parameters, dict):
                fb = np.asarray([asinh_softening_parameters[n_] for n_ in names], dtype=np.float64)
            else:
                fb = np.asarray(asinh_softening_parameters, dtype=np.float64).reshape(-1)
            if fb.size not in (1, len(names)):
                raise AssertionError("asinh_softening_parameter must be a list of the same length as the number of photometry filters.")
            fb = np.broadcast_to(fb, (len(names),)).astype(np.float32)
            mag, mag_err = flux_to_asinh(flux, fb, err) if err is not None else (flux_to_asinh(flux, fb, None), None)
        else:
            mag, mag_err = flux_to_abmag(flux, err, norm_mag_limit) if err is not None else (flux_to_abmag(flux, None, norm_mag_limit), None)
        norm_col, norm_name = None, None
        if normalize_method is not None:
            if normalize_method not in names:
                raise NotImplementedError("Normalization method not implemented.\n                    Please use a filter name for normalization.")
            j = names.index(normalize_method)
            keep = [i for i in range(len(names)) if i != j]
            ref = mag[:, j:j + 1]
            orig = torch.as_tensor(np.ascontiguousarray(grid[j]), dtype=torch.float64).cuda()   # unscattered library flux (nJy)
            if n_sc:
                orig = orig.repeat_interleave(n_sc)
            mag = mag[:, keep] - ref                                                   # colours (norm_func = np.subtract)
            if mag_err is not None:
                mag_err = mag_err[:, keep]
            if normalization_unit == "AB":
                norm_col = -2.5 * torch.log10(orig * 1e-3) + 23.9
            elif normalization_unit.startswith("log10 "):
                if normalization_unit.split(" ")[1] != "nJy":
                    raise ValueError("normalization_unit: 'AB' or 'log10 nJy' on the HIP path")
                norm_col = torch.log10(orig)
                norm_col[torch.isinf(norm_col)] = 0.0
            else:
                raise ValueError("normalization_unit: 'AB' or 'log10 nJy' on the HIP path")
            zero_norm = (ref.reshape(-1) == 0)
            names = [names[i] for i in keep]
            norm_name = f"norm_{normalize_method}_{normalization_unit}"
        if not asinh:
            mag = torch.where(mag > norm_mag_limit, torch.full_like(mag, norm_mag_limit), mag)   # 1927-1932 (AB only)
        cols, feature_names, units = [mag], list(names), [normed_flux_units] * len(names)
        error_names = [f"unc_{n_}" for n_ in names] if mag_err is not None else []
        if mag_err is not None and include_errors_in_feature_array:
            cols.append(mag_err)
            feature_names += error_names
            units += [normed_flux_units] * len(error_names)
        if norm_col is not None:
            cols.append(norm_col.float().reshape(-1, 1))
            feature_names.append(norm_name)
            units.append(normalization_unit)
        feat = torch.cat(cols, dim=1)
        delete = torch.zeros(feat.shape[0], dtype=torch.bool, device=feat.device)
        if norm_col is not None:
            delete |= zero_norm                                                        # 1917-1925
        if remove_nan_inf:
            bad = ~torch.isfinite(feat).all(dim=1)
            if verbose and int(bad.sum()):
                logger.warning(f"Warning: Deleting {int(bad.sum())} rows with NaN or Inf\n                    values in the feature array.")
            delete |= bad
        if drop_dropouts:
            nb = len(names)
            drop = (feat[:, :nb].abs() >= norm_mag_limit).sum(dim=1) >= nb * drop_dropout_fraction
            if verbose and int(drop.sum()):
                logger.warning(f"Warning: Dropping {int(drop.sum())} dropouts where more than\n                    "
                               f"{drop_dropout_fraction * 100}% of bands are at the norm_mag_limit.")
            delete |= drop
        delete_rows = torch.nonzero(delete).reshape(-1).cpu().numpy()
        if max_rows > 0 and feat.shape[0] - len(delete_rows) > max_rows:
            options = np.setdiff1d(np.arange(feat.shape[0]), delete_rows)
            chosen = np.random.default_rng(seed).choice(options, size=max_rows, replace=False)
            delete_rows = np.concatenate([delete_rows, np.setdiff1d(options, chosen)])
        keep_rows = np.setdiff1d(np.arange(feat.shape[0]), delete_rows)
        if keep_rows.size == 0:
            raise ValueError("All rows in the feature array were deleted. Please check the input parameters.")
        feat = feat[torch.as_tensor(keep_rows, device=feat.device)]
        self.feature_array = np.ascontiguousarray(feat.cpu().numpy().astype(np.float32))
        self.feature_names, self.feature_units, self.has_features = feature_names, units, True
        self.feature_array_flags = dict(normalize_method=normalize_method, extra_features=extra_features,
                                        normed_flux_units=normed_flux_units, normalization_unit=normalization_unit,
                                        scatter_fluxes=scatter_fluxes, depths=depths,
                                        include_errors_in_feature_array=include_errors_in_feature_array,
                                        min_flux_pc_error=min_flux_pc_error, norm_mag_limit=norm_mag_limit,
                                        remove_nan_inf=remove_nan_inf, parameters_to_remove=parameters_to_remove,
                                        photometry_to_remove=photometry_to_remove, parameters_to_add=parameters_to_add,
                                        drop_dropouts=drop_dropouts, drop_dropout_fraction=drop_dropout_fraction,
                                        raw_observation_names=names, error_names=error_names, norm_name=norm_name)
        self.update_parameter_array(parameters_to_remove=list(parameters_to_remove or []), delete_rows=np.sort(delete_rows),
                                    n_scatters=max(n_sc, 1), parameters_to_add=list(parameters_to_add or []),
                                    parameter_transformations=parameter_transformations)
        return self.feature_array, self.feature_names

    def update_parameter_array(self, parameters_to_remove: list = [], delete_rows=[], n_scatters: int = 1,
                               parameters_to_add: list = [], parameter_transformations: dict = None) -> None:
        """ref: sbi_runner.py:476-578 -- the fitted parameter array that goes with the feature array: columns removed /
        added (from the supplementary parameters), rows repeated per scatter copy, deleted rows dropped, optional
        per-parameter transformations (the column is renamed ``<fn>_<name>``)."""
        if getattr(self, "parameter_array", None) is None:
            raise ValueError("no parameter array: build the fitter from a library or pass parameter_array")
        arr = np.array(self.parameter_array, copy=True)
        pnames = [str(n_) for n_ in self.parameter_names]
        punits = None if self.parameter_units is None else list(self.parameter_units)
        for prm in np.unique(list(parameters_to_remove)):
            if prm in pnames:
                i = pnames.index(prm)
                arr = np.delete(arr, i, axis=1)
                pnames.pop(i)
                if punits is not None:
                    punits.pop(i)
        for prm in parameters_to_add:
            sn = [str(n_) for n_ in (self.supplementary_parameter_names or [])]
            if prm not in sn:
                raise ValueError(f"Can't add {prm} to parameter array - not found in supplementary parameters. "
                                 f"Available parameters: {self.supplementary_parameter_names}")
            i = sn.index(prm)
            arr = np.column_stack((arr, np.asarray(self.supplementary_parameters)[i]))
            pnames.append(prm)
            if punits is not None and self.supplementary_parameter_units is not None:
                punits.append(self.supplementary_parameter_units[i])
        if n_scatters > 1:
            arr = np.repeat(arr, n_scatters, axis=0)
        if len(delete_rows) > 0:
            arr = np.delete(arr, np.asarray(delete_rows, dtype=np.int64), axis=0)
        if parameter_transformations is not None:
            for prm, fn in parameter_transformations.items():
                if prm not in pnames:
                    raise ValueError(f"Parameter {prm} not found in fitted parameter names for transformation.")
                i = pnames.index(prm)
                arr[:, i] = fn(arr[:, i])
                pnames[i] = f"{fn.__name__}_{prm}"
                if punits is not None and punits[i] is not None and str(punits[i]) != "dimensionless":
                    punits[i] = f"{fn.__name__}({punits[i]})"
        self.fitted_parameter_array = arr
        self.fitted_parameter_names = pnames
        self.simple_fitted_parameter_names = [n_.split("/")[-1] for n_ in pnames]
        self.fitted_parameter_units = punits

    def split_dataset(self, train_fraction: float = 0.8, random_seed: int = None, verbose: bool = True) -> tuple:
        if random_seed is not None:
            np.random.seed(random_seed)
        if not self.has_features:
            raise ValueError("Feature array not created. Please create the feature array first.")
        n = self.feature_array.shape[0]
        idx = np.arange(n)
        np.random.shuffle(idx)
        k = int(n * train_fraction)
        return idx[:k], idx[k:]

    def create_priors(self, override_prior_ranges: dict = {}, prior=CustomIndependentUniform, verbose: bool = True,
                      debug_sample_acceptance: bool = False, extend_prior_range_pc: float = 0.0,
                      set_self: bool = False):
        if not self.has_features:
            raise ValueError("Feature array not created and no simulator.\n"
                             "                Please create the feature array first.")
        if self.fitted_parameter_array is None:
            raise ValueError("Parameter grid not created. Please create the parameter grid first.")
        p = prior_from_parameters(self.fitted_parameter_array, self.fitted_parameter_names,
                                  override_prior_ranges, extend_prior_range_pc, device="cpu")
        if verbose:
            logger.info("Prior ranges:")
            for n_, lo, hi in zip(self.fitted_parameter_names, p.low.tolist(), p.high.tolist()):
                logger.info(f"{n_}: {lo:.2f} - {hi:.2f}")
        if set_self:
            self._prior = p
        return p

    # ---------------------------------------------------------------------------------------
    def run_single_sbi(self, train_test_fraction: float = 0.8, random_seed: Optional[int] = None,
                       backend: str = "sbi", engine: Union[str, List[str]] = "NPE",
                       train_indices: Optional[np.ndarray] = None, test_indices: Optional[np.ndarray] = None,
                       n_nets: int = 1, model_type: Union[str, List[str]] = "mdn",
                       hidden_features: Union[int, List[int]] = 50, num_components: Union[int, List[int]] = 4,
                       num_transforms: Union[int, List[int]] = 4, training_batch_size: int = 64,
                       learning_rate: float = 1e-4, validation_fraction: float = 0.2, stop_after_epochs: int = 15,
                       clip_max_norm: float = 5.0, additional_model_args: dict = {}, save_model: bool = True,
                       verbose: bool = True, prior_method: str = "ili", out_dir: str = DEFAULT_MODEL_DIR, plot: bool = True,
                       name_append: str = "timestamp", feature_scalar: Callable = StandardScaler,
                       target_scalar: Callable = StandardScaler, set_self: bool = True, learning_type: str = "offline",
                       simulator: Optional[Callable] = None, num_simulations: int = 1000, num_online_rounds: int = 5,
                       initial_training_from_library: bool = False, override_prior_ranges: dict = {},
                       online_training_xobs: Optional[np.ndarray] = None, load_existing_model: bool = True,
                       use_existing_indices: bool = True, evaluate_model: bool = True, save_method: str = "joblib",
                       num_posterior_draws_per_sample: int = 1000, embedding_net: Optional[torch.nn.Module] = None,
                       custom_config_yaml: Optional[str] = None, sql_db_path: Optional[str] = None, *,
                       max_num_epochs: Optional[int] = None, optimizer_choice: str = "Adam", **unknown) -> tuple:
        """Train an ensemble of n_nets flows; returns (posteriors, stats) like the reference.

        ``custom_config_yaml`` (ref: sbi_runner.py:4570-4597, custom_runner.py:226-244, 298-365): a YAML file whose
        ``train_args`` has ``skip_optimization: True`` and ``fixed_params`` (``model_choice``, ``optimizer_choice``,
        ``learning_rate``, ``training_batch_size``, ``stop_after_epochs``, ``clip_max_norm`` and the model's own
        ``<model>_hidden_features`` / ``<model>_num_transforms`` / ``<model>_num_bins`` ...) trains ONE model with those
        values, ``validation_fraction`` from ``train_args`` (default 0.1); the Optuna search branch is out of scope.

        Positional order, names and defaults are the reference's (sbi_runner.py:4392-4435); ``max_num_epochs`` and
        ``optimizer_choice`` are keyword-only additions.  What the defaults mean here: ``backend="sbi"`` (alias ``"hip"``) = the
        nflows-style MAF / NSF that ili's sbi backend builds, on the HIP engine; ``"lampe"`` = its zuko-style NSF;
        ``model_type="mdn"`` -- the reference's default -- is NOT on the HIP path and raises, so a caller names "maf" or "nsf";
        ``plot=True`` is accepted and skipped with one log line (plotting is out of scope); ``evaluate_model=True`` runs
        ``evaluate_model`` on the test split.  ``out_dir`` gets the fitter's name appended (4541), an existing
        ``{out_dir}/{name}_{name_append}_params.pkl`` is loaded instead of training when ``load_existing_model`` (4546-4563:
        returns None when it is False), ``use_existing_indices`` re-uses a stored split that covers the feature array
        (4617-4636).  ``prior_method="manual"`` (4664-4690): ``feature_scalar()`` / ``target_scalar()`` are fit on the training
        rows, the flow is trained on the SCALED arrays and the box prior is min / max -+ 3 sigma of the scaled parameters --
        like the reference, nothing un-scales the draws afterwards: ``self._feature_scalar`` / ``self._target_scalar`` are the
        caller's tools for that.  Online learning (``simulator`` ...), ``sql_db_path`` and unknown keywords are refused or
        warned about, never silently dropped."""
        _warn_unknown("run_single_sbi", unknown)
        if simulator is not None or learning_type == "online" or initial_training_from_library or online_training_xobs is not None:
            raise ValueError("only learning_type='offline' (amortised NPE on a fixed library) is on the HIP path; simulator / "
                             "online rounds are not built")
        if sql_db_path is not None:
            raise ValueError("sql_db_path (the Optuna study database) belongs to the hyper-parameter search, which is outside the HIP path")
        if plot:
            logger.info("run_single_sbi(plot=True): plotting is outside the HIP path and is skipped")
        if custom_config_yaml is not None:
            import yaml
            with open(custom_config_yaml) as fh:
                ta = yaml.safe_load(fh)["train_args"]
            if not ta.get("skip_optimization", False):
                raise ValueError("custom_config_yaml without skip_optimization asks for the Optuna hyper-parameter search, "
                                 "which is outside the HIP path: set train_args.skip_optimization and fixed_params")
            fp = ta.get("fixed_params")
            if not fp or "model_choice" not in fp:  # custom_runner.py:229-233
                raise ValueError("`skip_optimization` is True, but `fixed_params` (including 'model_choice') "
                                 "are not defined in `train_args`.")
            model_type = fp["model_choice"]
            n_nets = 1
            margs = {k.split("_", 1)[1]: v for k, v in fp.items() if k.startswith(model_type + "_")}
            hidden_features = int(margs.pop("hidden_features", hidden_features))
            num_transforms = int(margs.pop("num_transforms", num_transforms))
            additional_model_args = {**additional_model_args, **margs}
            optimizer_choice = fp.get("optimizer_choice", "Adam")
            learning_rate = fp["learning_rate"]
            training_batch_size = fp.get("training_batch_size", 32)
            stop_after_epochs = fp.get("stop_after_epochs", 20)
            clip_max_norm = fp.get("clip_max_norm", 5.0)
            validation_fraction = ta.get("validation_fraction", 0.1)
        if backend == "lampe":   # the reference's second backend name (sbi_runner.py:5123-5125): its NSF, on the HIP engine
            additional_model_args = dict(additional_model_args or {}, backend="lampe")
            backend = "hip"
        if backend == "sbi":     # ili's sbi backend = nflows-style flows: what the HIP engine implements
            backend = "hip"
        if backend != "hip":
            raise ValueError(f"backend '{backend}' is not available in synference_amd: use backend='sbi' (alias 'hip': nflows-style "
                             "MAF / NSF) or 'lampe' (the zuko-style autoregressive NSF of the reference's lampe backend)")
        if learning_type != "offline":
            raise ValueError("only learning_type='offline' (amortised NPE) is on the HIP path")
        if prior_method not in ("ili", "manual"):
            raise ValueError("Invalid prior method. Use 'manual' or 'ili'.")   # sbi_runner.py:4700-4701
        if not self.has_features:
            raise ValueError("Feature array not created. Please create the feature array first.")
        if self.fitted_parameter_array is None:
            raise ValueError("Parameter grid not created. Please create the parameter grid first.")
        run_out_dir = None
        if out_dir is not None:
            run_out_dir = os.path.join(os.path.abspath(out_dir), self.name)      # sbi_runner.py:4541
        stamp = self._timestamp if name_append == "timestamp" else str(name_append)
        if run_out_dir is not None and save_model and os.path.exists(f"{run_out_dir}/{self.name}_{stamp}_params.pkl"):
            if load_existing_model:   # sbi_runner.py:4546-4557
                logger.info(f"Loading existing model from {run_out_dir}/{self.name}_{stamp}_params.pkl")
                posteriors, stats, _params = self.load_model_from_pkl(f"{run_out_dir}/{self.name}_{stamp}_posterior.pkl", set_self=set_self)
                return posteriors, stats
            logger.info("Model with same name already exists. Please change the name of this model or delete the existing one.")
            return None
        engines = [engine] * n_nets if isinstance(engine, str) else list(engine)
        models = [model_type] * n_nets if isinstance(model_type, str) else list(model_type)
        hf = [hidden_features] * n_nets if isinstance(hidden_features, int) else list(hidden_features)
        nt = [num_transforms] * n_nets if isinstance(num_transforms, int) else list(num_transforms)
        for m in models:
            if m not in SUPPORTED_MODELS:
                raise ValueError(f"model_type '{m}' is not on the HIP path; supported: {SUPPORTED_MODELS}")
        if train_indices is None:   # sbi_runner.py:4617-4636
            have = getattr(self, "_train_indices", None) is not None and getattr(self, "_test_indices", None) is not None
            if (not have or not use_existing_indices or
                    len(self._train_indices) + len(self._test_indices) != self.feature_array.shape[0]):
                train_indices, test_indices = self.split_dataset(train_test_fraction, random_seed, verbose)
            else:
                logger.info("Using existing train and test indices.")
                train_indices, test_indices = self._train_indices, self._test_indices
        X_train = self.feature_array[train_indices]
        y_train = self.fitted_parameter_array[train_indices]
        X_test = self.feature_array[test_indices] if test_indices is not None else None
        y_test = self.fitted_parameter_array[test_indices] if test_indices is not None else None
        if prior_method == "manual":   # sbi_runner.py:4664-4690
            self._feature_scalar, self._target_scalar = feature_scalar(), target_scalar()
            X_train = np.asarray(self._feature_scalar.fit(X_train).transform(X_train), dtype=np.float32)
            y_train = np.asarray(self._target_scalar.fit(y_train).transform(y_train))
            if X_test is not None:
                X_test = np.asarray(self._feature_scalar.transform(X_test), dtype=np.float32)
                y_test = np.asarray(self._target_scalar.transform(y_test))
            y_std, y_min, y_max = np.std(y_train, axis=0), np.min(y_train, axis=0), np.max(y_train, axis=0)
            from .priors import CustomIndependentUniform
            prior = CustomIndependentUniform(low=torch.tensor(y_min - 3 * y_std, dtype=torch.float32),
                                             high=torch.tensor(y_max + 3 * y_std, dtype=torch.float32),
                                             name_list=self.fitted_parameter_names, device="cpu")
        else:
            # the prior box spans the WHOLE parameter array (train + test rows), as in the reference's create_priors
            # (sbi_runner.py:3519-3520)
            prior = self.create_priors(override_prior_ranges, verbose=verbose)
        nets = []
        for i in range(n_nets):
            args = dict(hidden_features=hf[i], num_transforms=nt[i])
            args.update(additional_model_args)
            nets.append(load_nde_hip(engines[i], model=models[i], embedding_net=embedding_net, **args))
        train_args = dict(training_batch_size=training_batch_size, learning_rate=learning_rate,
                          validation_fraction=validation_fraction, stop_after_epochs=stop_after_epochs,
                          clip_max_norm=clip_max_norm, log_every=1 if verbose else 0,
                          optimizer_choice=optimizer_choice)
        if max_num_epochs is not None:
            train_args["max_num_epochs"] = max_num_epochs
        run_name = f"{self.name}_{stamp}_"
        trainer = HIPRunner.load(backend="hip", engine=engines[0], prior=prior, nets=nets, train_args=train_args,
                                 out_dir=(run_out_dir if save_model else None), device=self.device, name=run_name)
        t0 = time.time()
        try:
            posteriors, stats = trainer(NumpyLoader(X_train, y_train), seed=random_seed)
        except Exception as e:  # sbi_runner.py:4940-4941
            raise RuntimeError(f"Error during SBI training: {e}") from e
        self.training_time = time.time() - t0
        if set_self:
            self.posteriors, self.stats, self._prior = posteriors, stats, prior
            self._train_indices, self._test_indices = train_indices, test_indices
            self._X_train, self._y_train, self._X_test, self._y_test = X_train, y_train, X_test, y_test
            self.fitted_model_name = run_name
        if save_model and run_out_dir is not None:  # sbi_runner.py:4973-5014: the fitter's state next to the posterior pickle
            prior_was = self._prior
            self._prior = prior
            try:
                self.save_state(out_dir=run_out_dir, name_append=stamp, save_method=save_method, has_grid=True, engine=engine, learning_type=learning_type,
                                ensemble_model_types=list(models), ensemble_model_args=[dict(hidden_features=hf[i],
                                num_transforms=nt[i], **additional_model_args) for i in range(n_nets)], n_nets=n_nets,
                                train_args=train_args, stats=stats, training_time=self.training_time,
                                train_fraction=train_test_fraction, test_indices=test_indices, train_indices=train_indices)
            finally:
                self._prior = prior_was if not set_self else prior
        if evaluate_model and X_test is not None:
            self.evaluate_model(posteriors=posteriors, X_test=X_test, y_test=y_test,
                                num_samples=num_posterior_draws_per_sample)
        return posteriors, stats

    # ---------------------------------------------------------------------------------------
    def sample_posterior(self, X_test: np.ndarray = None, sample_method: str = "direct", sample_kwargs: dict = {},
                         posteriors: object = None, num_samples: int = 1000, timeout_seconds_per_test=30,
                         log_times=False, seed: Optional[int] = None, **kwargs) -> np.ndarray:
        """(num_objects, num_samples, num_parameters) float64; NaN rows where sampling failed
        (ref: sbi_runner.py:6436, 6458-6460); a single observation returns (num_samples, num_parameters)."""
        if posteriors is None:
            posteriors = self.posteriors
        if X_test is None:
            if getattr(self, "_X_test", None) is not None:
                X_test = self._X_test
            else:
                raise ValueError("X_test must be provided or set in the object.")
        if sample_method != "direct":
            raise ValueError("Invalid sample method for the HIP backend. Use 'direct'.")
        X = np.asarray(X_test, dtype=np.float32)
        single = X.ndim == 1 or (X.ndim == 2 and X.shape[0] == 1)
        if X.ndim == 1:
            X = X[None, :]
        # Under an initialised process group (one process per GPU) the catalogue is sharded: rank r samples rows
        # [r N / W, (r+1) N / W) -- rows are independent (ref: sbi_runner.py:6438-6442), weights are replicated, there is no
        # data-path collective -- and the blocks are gathered afterwards.  ``gather`` (keyword): "rank0" (default) -- rank 0
        # returns the whole (N, S, D) array, every other rank its own block (rows ``self.last_shard_rows``): ONE copy of the
        # array exists (configs[4]: 3.2 GB; all ranks holding it would move 8 x that over xGMI); "all" -- every rank returns
        # the whole array; "none" -- every rank keeps its block.  The random streams are keyed by the row's position in the
        # catalogue: the gathered result is the single-process one, bit for bit.
        from .hostio import to_host_f64
        from .posterior import all_gather_rows, broadcast_seed, dist_world, gather_rows, shard_bounds
        rank, world = dist_world()
        if world > 1 and len(X) >= world and kwargs.get("shard", True):
            if seed is None:
                seed = posteriors._next_seed(None)
            seed = broadcast_seed(seed)
            b = shard_bounds(len(X), world)
            tmo = float(timeout_seconds_per_test) * (b[rank + 1] - b[rank]) if timeout_seconds_per_test else None
            t0 = time.time()
            local = posteriors.sample_catalogue(torch.as_tensor(X[b[rank]:b[rank + 1]]), num_samples, seed,
                                                timeout_seconds=tmo, row_offset=b[rank])
            if log_times:
                per = (time.time() - t0) / max(1, b[rank + 1] - b[rank])
                self.last_times_per_object = np.full(b[rank + 1] - b[rank], per)
                self.last_time_per_object = float(per)
            self.last_shard_rows = (int(b[rank]), int(b[rank + 1]))
            how = kwargs.get("gather", "rank0")
            if how not in ("rank0", "all", "none"):
                raise ValueError("gather must be 'rank0', 'all' or 'none'")
            full = all_gather_rows(local, b) if how == "all" else (gather_rows(local, b, dst=0) if how == "rank0" else None)
            return to_host_f64(full if full is not None else local)
        # log_times: the reference times every object (sbi_runner.py:6438-6469: median and 16th-84th percentile of the
        # per-object wall time); the catalogue call is timed in chunks instead and each chunk's time is shared equally
        # by its objects, so the three statistics keep their meaning without a per-galaxy host loop
        # without log_times ONE catalogue call, then the native hand-over.  (Measured: cutting the catalogue in 2 .. 6 chunks so
        # that chunk k crosses the bus while chunk k + 1 is drawn costs more than it hides -- every chunk pays the sampler's
        # tail: 3.6 / 3.9 / 4.3 / 4.8 ms for 1 / 2 / 3 / 4 chunks of the cfg2 catalogue.  SF_API_CHUNKS=n forces n.)
        auto_chunks = int(os.environ.get("SF_API_CHUNKS", "0")) or 1
        n_chunks = min(len(X), 16) if log_times else max(1, min(len(X), auto_chunks))
        if seed is None and n_chunks > 1:
            seed = posteriors._next_seed(None)
        bounds = np.linspace(0, len(X), n_chunks + 1).astype(int)
        # (uninitialised: every chunk either fills its rows or, on failure, sets them to NaN below)
        from .hostio import pinned_result, result_array
        # ONE catalogue call on a one-member flow posterior: the sampler writes the float64 host container itself (pinned memory
        # mapped into the device's address space; sf_flow_set_sample_output_f64) -- the draws cross PCIe while the kernel runs
        # instead of in a copy + widening pass afterwards (cfg2 catalogue: 2.9 ms against 2.6 + 1.0)
        direct = None
        if n_chunks == 1 and not log_times and os.environ.get("SF_API_DIRECT", "1") != "0":
            members = getattr(posteriors, "posteriors", [posteriors])
            try:
                if len(members) == 1 and members[0].posterior_estimator.flow.supports_f64_out():
                    direct = pinned_result((len(X), num_samples, len(self.fitted_parameter_names)))
            except AttributeError:
                direct = None
        if direct is not None:
            pin_t, samples = direct
        else:
            samples = result_array((len(X), num_samples, len(self.fitted_parameter_names)))
        times = []
        pending = []
        for ci in range(n_chunks):
            a, b = int(bounds[ci]), int(bounds[ci + 1])
            if b <= a:
                continue
            t0 = time.time()
            try:
                # the reference's per-object timeout (sbi_runner.py:6358) becomes the wall-clock ceiling of the chunk
                tmo = float(timeout_seconds_per_test) * (b - a) if timeout_seconds_per_test else None
                # (same seed, rows keyed by their position: the draws do not depend on the chunking)
                if direct is not None:
                    posteriors.sample_catalogue(torch.as_tensor(X[a:b]), num_samples, seed, timeout_seconds=tmo, row_offset=a,
                                                out=pin_t[a:b])
                    torch.cuda.synchronize()     # the host array is complete when the stream is
                    times.extend([(time.time() - t0) / (b - a)] * (b - a))
                    continue
                s = posteriors.sample_catalogue(torch.as_tensor(X[a:b]), num_samples, seed, timeout_seconds=tmo, row_offset=a)
                # D2H in float32 (half the PCIe bytes of a device-side .double()) through a ring of pinned staging buffers
                # on a copy stream, widened into the reference's float64 container by a thread pool while the next piece
                # is on the bus (hostio.py); with log_times the chunk's time includes its hand-over
                # (one chunk, or timed chunks: the call itself; several untimed chunks: a helper thread, so that the next
                #  chunk's kernels are launched meanwhile)
                if log_times or n_chunks == 1:
                    to_host_f64(s, out=samples[a:b])
                else:
                    pending.append((a, b, to_host_f64(s, out=samples[a:b], wait=False)))
            except Exception as e:  # sbi_runner.py:6458-6460: failed objects are NaN rows
                logger.error(f"Error occurred while sampling objects {a}..{b}: {e}")
                samples[a:b] = np.nan
            times.extend([(time.time() - t0) / (b - a)] * (b - a))
        for a, b, p in pending:
            try:
                p.result()
            except Exception as e:
                logger.error(f"Error occurred while copying objects {a}..{b} to the host: {e}")
                samples[a:b] = np.nan
        if log_times and times:
            self.last_times_per_object = np.asarray(times)
            self.last_time_per_object = float(np.median(times))
            logger.info(f"Median time per sample: {np.median(times):.5f} seconds."
                        f"16th-84th: {np.percentile(times, 16):.5f}-{np.percentile(times, 84):.5f}s.")
        return np.squeeze(samples, 0) if single else samples

    def log_prob(self, X: np.ndarray, y: np.ndarray, posteriors: object = None, verbose=True,
                 norm_posterior: bool = True, num_rejection_samples: int = 10000) -> np.ndarray:
        """(N,) float64 posterior log-density of y[i] given X[i]  (ref: sbi_runner.py:7188-7198; upstream
        DirectPosterior default norm_posterior=True -> leakage-corrected)."""
        if posteriors is None:
            posteriors = self.posteriors
        Xt = torch.as_tensor(np.asarray(X, dtype=np.float32))
        yt = torch.as_tensor(np.asarray(y, dtype=np.float32))
        from .posterior import all_gather_rows, broadcast_seed, dist_world, shard_bounds
        rank, world = dist_world()
        if world > 1 and Xt.dim() == 2 and yt.dim() == 2 and len(Xt) == len(yt) and len(Xt) >= world:
            # rank-sharded (rows independent, ref: sbi_runner.py:7193-7196): every rank evaluates its row block; the
            # DISTINCT contexts behind the leakage correction are split over the ranks too and their acceptance rates
            # gathered, with streams keyed by the position in the sorted list of distinct rows -- same result as one process
            b = shard_bounds(len(Xt), world)
            members = getattr(posteriors, "posteriors", [posteriors])
            acc = None
            if norm_posterior and posteriors.prior is not None:
                dev = members[0].device
                acc = []
                for mem in members:
                    E = mem._embed(Xt)
                    ux, inv = torch.unique(E, dim=0, return_inverse=True)
                    ub = shard_bounds(ux.shape[0], world)
                    seed = broadcast_seed(mem._next_seed(None))
                    part = mem.acceptance_rows(ux[ub[rank]:ub[rank + 1]], num_rejection_samples, seed, row_offset=ub[rank])
                    acc_all = all_gather_rows(part, ub) if ux.shape[0] >= world else mem.acceptance_rows(ux, num_rejection_samples, seed)
                    acc.append(acc_all.to(dev)[inv][b[rank]:b[rank + 1]])
            sl = slice(b[rank], b[rank + 1])
            if hasattr(posteriors, "posteriors"):
                lp = posteriors.log_prob_catalogue(yt[sl], Xt[sl], norm_posterior, num_rejection_samples, acc=acc)
            else:
                lp = posteriors.log_prob_catalogue(yt[sl], Xt[sl], norm_posterior, num_rejection_samples,
                                                   **({} if acc is None else {"acc_rows": acc[0]}))
            return all_gather_rows(lp.contiguous(), b).double().cpu().numpy()
        lp = posteriors.log_prob_catalogue(yt, Xt, norm_posterior, num_rejection_samples)
        return lp.double().cpu().numpy()

    def save_state(self, out_dir, name_append: str = "", save_method: str = "joblib", has_grid: bool = True, **extras):
        """ref: sbi_runner.py:693-830 -- what a later session needs next to ``{name}_{append}_posterior.pkl``: feature /
        parameter names and units, the prior, the feature-array flags and (``has_grid``) the feature and parameter
        arrays, plus ``extras`` (train arguments, split indices, stats ...), as ``{name}{_append}_params.pkl``
        (``save_method`` 'joblib' -- the reference's default --, 'pickle' or 'torch'); ``stats`` also go to
        ``{name}{_append}_summary.json`` with the scalar entries of the dictionary appended."""
        import json
        import pickle
        os.makedirs(out_dir, exist_ok=True)
        param_dict = {"feature_names": self.feature_names, "feature_units": self.feature_units,
                      "fitted_parameter_units": self.fitted_parameter_units,
                      "fitted_parameter_names": self.fitted_parameter_names,
                      "timestamp": time.strftime("%Y%m%d_%H%M%S"), "prior": self._prior,
                      "library_path": self.library_path, "name": self.name, "has_simulator": self.has_simulator}
        if len(name_append) > 0 and name_append[0] != "_":
            name_append = f"_{name_append}"
        save_path = f"{out_dir}/{self.name}{name_append}_params.pkl"
        param_dict.update(extras)
        if has_grid:
            param_dict["feature_array_flags"] = getattr(self, "feature_array_flags", {})
            param_dict["feature_array"] = self.feature_array
            param_dict["parameter_array"] = self.fitted_parameter_array
        if "stats" in param_dict:
            summary = list(param_dict["stats"]) + [{k: v for k, v in param_dict.items()
                                                    if isinstance(v, (list, str, float, int, bool)) and k != "stats"}]
            try:
                with open(f"{out_dir}/{self.name}{name_append}_summary.json", "w") as fh:
                    json.dump(summary, fh, indent=4, default=lambda o: o.tolist() if hasattr(o, "tolist") else str(o))
            except Exception as e:  # (the reference logs and goes on)
                logger.error(f"Error saving stats: {e}")
        if save_method == "joblib":
            from joblib import dump
            dump(param_dict, save_path, compress=3)
        elif save_method == "pickle":
            with open(save_path, "wb") as fh:
                pickle.dump(param_dict, fh, protocol=pickle.HIGHEST_PROTOCOL)
        elif save_method == "torch":
            torch.save(param_dict, save_path)
        else:
            raise ValueError("save_method: 'joblib', 'pickle' or 'torch' on the HIP path")
        return save_path

    def load_model_from_pkl(self, model_file: str, set_self: bool = True, load_arrays: bool = True):
        """ref: sbi_runner.py:7401-7633 -- (posteriors, stats, params) of a model trained by THIS backend: ``model_file`` is
        the ``*_posterior.pkl`` or the directory that holds exactly one; ``*_summary.json`` and ``*_params.pkl`` next to
        it are read when present and, with ``set_self``, put back on the fitter (names, units, flags, prior, arrays,
        the train / test split).  sbi's own pickles need sbi and stay outside the build (see ``importer.py`` for their
        weights)."""
        import glob
        import json
        import pickle
        if not os.path.exists(model_file):
            raise ValueError(f"Model file {model_file} does not exist.")
        if os.path.isdir(model_file):
            files = glob.glob(os.path.join(model_file, "*_posterior.pkl"))
            if len(files) == 0:
                raise ValueError(f"No parameter files found in {model_file}.")
            if len(files) > 1:
                raise ValueError(f"Multiple parameter files found in {model_file}.\n                    Please specify a single file.")
            model_file = files[0]
        with open(model_file, "rb") as fh:
            posteriors = pickle.load(fh)
        stats = None
        sfile = model_file.replace("posterior.pkl", "summary.json")
        if os.path.exists(sfile):
            with open(sfile, "r", encoding="utf-8") as fh:
                stats = json.load(fh)
            if set_self:
                self.stats = stats
        else:
            logger.info(f"Warning: No summary file found for {model_file}.")
        if hasattr(posteriors, "to"):
            posteriors = posteriors.to(self.device) or posteriors
        if set_self:
            self.posteriors = posteriors
        pfile = model_file.replace("posterior.pkl", "params.pkl")
        params = None
        if os.path.exists(pfile):
            try:
                from joblib import load as jl_load
                params = jl_load(pfile)
            except Exception:
                try:
                    with open(pfile, "rb") as fh:
                        params = pickle.load(fh)
                except Exception:
                    params = torch.load(pfile, map_location="cpu", weights_only=False)
            if set_self:
                self.fitted_parameter_names = list(params["fitted_parameter_names"])
                self.simple_fitted_parameter_names = [str(i).split("/")[-1] for i in self.fitted_parameter_names]
                self.fitted_parameter_units = params.get("fitted_parameter_units", self.parameter_units)
                self.feature_names = params["feature_names"]
                self.feature_units = params.get("feature_units", None)
                if "feature_array_flags" in params:
                    self.feature_array_flags = params["feature_array_flags"]
                    self.has_features = True
                if load_arrays and params.get("feature_array") is not None:
                    self.fitted_parameter_array = params["parameter_array"]
                    self.feature_array = params["feature_array"]
                    self.has_features = True
                    self._train_indices = params.get("train_indices")
                    self._test_indices = params.get("test_indices")
                    self._train_fraction = params.get("train_fraction")
                    if self._train_indices is not None and self._test_indices is not None:
                        self._X_test = self.feature_array[self._test_indices]
                        self._y_test = self.fitted_parameter_array[self._test_indices]
                        self._X_train = self.feature_array[self._train_indices]
                        self._y_train = self.fitted_parameter_array[self._train_indices]
                self._train_args = params.get("train_args")
                self._prior = params.get("prior")
                self._ensemble_model_types = params.get("ensemble_model_types")
                self._ensemble_model_args = params.get("ensemble_model_args")
        else:
            logger.warning(f"No parameter file found for {model_file}.")
        return posteriors, stats, params

    def create_features_from_observations(self, observations, columns_to_feature_names: dict = None, flux_units=None,
                                          missing_data_flag=-99, override_transformations: dict = {},
                                          ignore_missing: bool = False):
        """ref: sbi_runner.py:2473-2937 -- the observed catalogue as the (N', F) float32 feature array the model was
        trained on, plus the mask of removed rows, from the transformations recorded by
        ``create_feature_array_from_raw_photometry`` (``self.feature_array_flags``).

        As in the reference the photometry columns must ALREADY be in the training units (``flux_units`` has to equal
        ``normed_flux_units``, an assertion there too); what happens here is column mapping and validation (every
        filter, every ``unc_`` column when errors were features, the ``norm_<filter>_<unit>`` column), the error-NaN
        check, the normalisation step exactly as the reference writes it (2844-2865), removal of rows that carry the
        missing-data flag, the ``norm_mag_limit`` clip and inf -> NaN.  Empirical noise models, flags and simulated
        missing fluxes are outside the HIP path."""
        import pandas as pd
        flags = dict(getattr(self, "feature_array_flags", None) or {})
        if len(flags) == 0:
            raise ValueError("No feature array flags found. Please create the feature array first.")
        flags.update(override_transformations)
        if not isinstance(observations, pd.DataFrame):
            raise TypeError("Observations must be a pandas DataFrame or an astropy Table.")
        if columns_to_feature_names is None:
            columns_to_feature_names = {col: col for col in observations.columns}
        feature_names_to_columns = {v: k for k, v in columns_to_feature_names.items()}
        for name in flags["raw_observation_names"]:
            if name not in feature_names_to_columns:
                raise ValueError(f"Column '{name}' not found in observations. Please provide a mapping for all photometry filters.")
        if flags.get("include_errors_in_feature_array"):
            for name in flags["error_names"]:
                if name not in feature_names_to_columns:
                    raise ValueError(f"Column '{name}' not found in observations. Please provide a mapping for all errors.")
        if flags.get("norm_name") is not None and flags["norm_name"] not in feature_names_to_columns:
            raise ValueError(f"Column '{flags['norm_name']}' not found in\n                observations. "
                             "Please provide a mapping for the normalization factor.")
        training_flux_units = flags["normed_flux_units"]
        assert flux_units == training_flux_units, (f"Flux units '{flux_units}' do not match\n                    "
                                                   f"training data units '{training_flux_units}'.")
        fnames = list(self.feature_names)
        nrows, ncols = observations.shape[0], np.shape(self.feature_array)[1]
        fa = np.zeros((ncols, nrows), dtype=np.float32)
        photometry_columns = [feature_names_to_columns[name] for name in flags["raw_observation_names"]]
        for col in photometry_columns:
            if col not in observations.columns:
                raise ValueError(f"Column '{col}' not found in observations.\n                    "
                                 "Please provide a mapping for all photometry filters.")
            fa[fnames.index(columns_to_feature_names[col]), :] = observations[col].values
        err_names = flags["error_names"] if flags.get("include_errors_in_feature_array") else []
        for col in err_names:
            ocol = feature_names_to_columns[col]
            if ocol not in observations.columns:
                raise ValueError(f"Column '{ocol}' not found in observations.\n                    "
                                 "Please provide a mapping for all errors.")
            fa[fnames.index(col), :] = observations[ocol].values
        for col in err_names:
            ev, fv = fa[fnames.index(col), :], fa[fnames.index(col.replace("unc_", "")), :]
            bad = np.isnan(ev) & ~np.isnan(fv)
            if np.sum(bad) > 0:
                raise ValueError(f"Error column '{col}' contains NaN values where the\n                    corresponding flux "
                                 f"column '{col.replace('unc_', '')}' does not.{np.sum(bad)} NaN values found.")
        if flags.get("norm_name") is not None:
            ncol = feature_names_to_columns[flags["norm_name"]]
            if ncol not in observations.columns:
                raise ValueError(f"Column '{flags['norm_name']}'\n                    not found in observations.\n"
                                 "                    Please provide a mapping for the normalization factor.")
            fa[fnames.index(flags["norm_name"]), :] = observations[ncol].values
        if flags.get("normalize_method") is not None:
            nf = fa[fnames.index(flags["norm_name"]), :]
            # exactly as the reference writes it (2847-2849): the AB column goes back to a uJy flux and is SUBTRACTED
            nf = 10 ** ((23.9 - nf) / 2.5)
            for col in photometry_columns:
                i = fnames.index(columns_to_feature_names[col])
                fa[i, :] = np.subtract(fa[i, :], nf)
        removed = np.zeros(nrows, dtype=bool)
        missing = np.isnan(fa) if (isinstance(missing_data_flag, float) and np.isnan(missing_data_flag)) else (fa == missing_data_flag)
        if not ignore_missing:
            if missing.sum() > 0:
                logger.info(f"Removing {int(missing.sum())} observations with missing data.")
            removed[missing.any(axis=0)] = True
        fa, missing = fa[:, ~removed], missing[:, ~removed]
        clip = (fa > flags["norm_mag_limit"]) & ~missing
        fa[clip] = flags["norm_mag_limit"]
        fa[~np.isfinite(fa)] = np.nan
        return fa.T, removed

    def fit_catalogue(self, observations, columns_to_feature_names: dict = None, num_samples: int = 1000,
                      quantiles=(0.16, 0.5, 0.84), sample_method: str = "direct", append_to_input: bool = True,
                      return_samples: bool = False, log_times: bool = False, seed: Optional[int] = None,
                      device_quantiles: bool = True, flux_units=None, missing_data_flag=-99,
                      override_transformations: dict = {}, timeout_seconds_per_row: float = 5,
                      return_feature_array: bool = False, return_full_samples: bool = False, **unknown):
        """Sampling + quantile section of the reference's fit_catalogue (sbi_runner.py:3230-3282).

        ``observations`` is a pandas DataFrame / dict of columns / (N, C) array.  With ``flux_units`` given and a feature
        array that was built by ``create_feature_array_from_raw_photometry`` the table goes through
        ``create_features_from_observations`` first (3061-3068: column mapping, normalisation, missing-data rows);
        otherwise its columns are taken as the model's feature columns.  Masked rows get NaN quantiles.
        ``timeout_seconds_per_row`` (reference default 5 s) x rows is the wall-clock ceiling of the catalogue call
        (sbi_runner.py:3246-3253); ``return_feature_array`` returns (feature_array, mask) like line 3092-3094;
        ``return_full_samples`` is the reference's name for ``return_samples``."""
        import pandas as pd
        _warn_unknown("fit_catalogue", unknown)
        return_samples = return_samples or return_full_samples
        if flux_units is not None and getattr(self, "feature_array_flags", None):
            df0 = pd.DataFrame(observations) if isinstance(observations, dict) else observations
            feats_ok, removed = self.create_features_from_observations(df0, columns_to_feature_names, flux_units,
                                                                       missing_data_flag, override_transformations)
            if return_feature_array:
                return feats_ok, removed
            full = np.full((len(df0), feats_ok.shape[1]), np.nan, dtype=np.float32)
            full[~removed] = feats_ok
            out = self.fit_catalogue(full, columns_to_feature_names=None, num_samples=num_samples, quantiles=quantiles,
                                     sample_method=sample_method, append_to_input=False, return_samples=return_samples,
                                     log_times=log_times, seed=seed, device_quantiles=device_quantiles,
                                     timeout_seconds_per_row=timeout_seconds_per_row)
            qt = out[0] if return_samples else out
            table = df0.copy() if append_to_input else pd.DataFrame({"ID": np.arange(len(df0)) + 1})
            for c in qt.columns:
                if c != "ID":
                    table[c] = qt[c].to_numpy()
            return (table, out[1]) if return_samples else table
        if isinstance(observations, np.ndarray):
            df = pd.DataFrame(observations, columns=list(self.feature_names)[: observations.shape[1]])
        elif isinstance(observations, dict):
            df = pd.DataFrame(observations)
        else:
            df = observations.copy()
        cols = list(self.feature_names)
        if columns_to_feature_names:
            df = df.rename(columns=columns_to_feature_names)
        missing = [c for c in cols if c not in df.columns]
        if missing:
            raise ValueError(f"observations lack the feature columns {missing}")
        feats = df[cols].to_numpy(dtype=np.float32)
        obs_mask = ~np.isfinite(feats).all(1)
        if return_feature_array:
            return feats[~obs_mask], obs_mask
        tmo = float(timeout_seconds_per_row) * max(1, int((~obs_mask).sum())) if timeout_seconds_per_row else None
        if device_quantiles and not return_samples and num_samples <= 8192:
            # f3: quantiles reduced on the GPU; only (N, D, Q) floats cross PCIe
            from .posterior import device_quantiles as _dq
            table = df.copy() if append_to_input else pd.DataFrame({"ID": np.arange(len(df)) + 1})
            qarr = np.full((len(df), len(self.fitted_parameter_names), len(quantiles)), np.nan)
            if (~obs_mask).any():
                if sample_method != "direct":
                    raise ValueError("Invalid sample method for the HIP backend. Use 'direct'.")
                from .posterior import all_gather_rows, broadcast_seed, dist_world, shard_bounds
                rank, world = dist_world()
                good = feats[~obs_mask]
                if world > 1 and len(good) >= world:   # rank r: its row block; only the (n, D, Q) quantiles are gathered
                    if seed is None:
                        seed = self.posteriors._next_seed(None)
                    seed = broadcast_seed(seed)
                    b = shard_bounds(len(good), world)
                    s_dev = self.posteriors.sample_catalogue(torch.as_tensor(good[b[rank]:b[rank + 1]]), num_samples, seed,
                                                             timeout_seconds=tmo, row_offset=b[rank])
                    qarr[~obs_mask] = all_gather_rows(_dq(s_dev, quantiles).contiguous(), b).double().cpu().numpy()
                else:
                    s_dev = self.posteriors.sample_catalogue(torch.as_tensor(good), num_samples, seed, timeout_seconds=tmo)
                    qarr[~obs_mask] = _dq(s_dev, quantiles).double().cpu().numpy()
            for i, param in enumerate(self.simple_fitted_parameter_names):
                for j, qv in enumerate(quantiles):
                    table[f"{param}_{int(qv * 100)}"] = qarr[:, i, j]
            return table
        samples = np.full((len(df), num_samples, len(self.fitted_parameter_names)), np.nan)
        if (~obs_mask).any():
            samples[~obs_mask] = self.sample_posterior(feats[~obs_mask], sample_method=sample_method,
                                                       num_samples=num_samples, log_times=log_times, seed=seed,
                                                       timeout_seconds_per_test=timeout_seconds_per_row,
                                                       gather="all")   # every rank fills the whole table
        samples_quant = samples.transpose(2, 0, 1)
        table = df.copy() if append_to_input else pd.DataFrame({"ID": np.arange(len(df)) + 1})
        for i, param in enumerate(self.simple_fitted_parameter_names):
            with np.errstate(invalid="ignore"):
                q = np.nanquantile(samples_quant[i], quantiles, axis=1) if (~obs_mask).any() else \
                    np.full((len(quantiles), len(df)), np.nan)
            for j, quant in enumerate(q):
                col = np.asarray(quant, dtype=np.float64)
                col[obs_mask] = np.nan
                table[f"{param}_{int(quantiles[j] * 100)}"] = col
        return (table, samples) if return_samples else table

    def evaluate_model(self, posteriors=None, X_test=None, y_test=None, num_samples: int = 1000,
                       independent_metrics: bool = True, seed: Optional[int] = None, samples=None,
                       verbose: bool = False, **unknown) -> dict:
        """The reference's evaluate_model for flow posteriors (sbi_runner.py:6484-6735), same keys and arithmetic:
        ``MSE``, ``RMSE``, ``mean_ae``, ``median_ae``, ``R_squared`` (total sum of squares about the GLOBAL mean of
        y_test, as there), ``RMSE_norm`` / ``mean_ae_norm`` (divided by the global std), ``log_dpit_max``
        (-0.5 log max |PIT - uniform|, 6613-6616) and ``mean_log_prob``; per parameter with ``independent_metrics``, else
        pooled.  ``tarp`` needs the third-party ``tarp`` package and is left out.  The draws stay on the device: means,
        medians and PIT ranks are reduced there and only (N, D) summaries cross PCIe."""
        _warn_unknown("evaluate_model", unknown)
        posteriors = posteriors if posteriors is not None else self.posteriors
        X_test = self._X_test if X_test is None else X_test
        y_test = self._y_test if y_test is None else y_test
        if isinstance(samples, str):
            samples = np.load(samples)
        if samples is not None:
            if samples.shape[1] != num_samples:
                raise ValueError(f"Samples must have {num_samples} samples per test sample, but got {samples.shape[1]}.")
            sd = torch.as_tensor(np.asarray(samples, dtype=np.float32)).to(self.device)
        else:
            sd = posteriors.sample_catalogue(torch.as_tensor(np.asarray(X_test, dtype=np.float32)), num_samples, seed)
        mean_pred = torch.nanmean(sd, dim=1).double().cpu().numpy()
        from .posterior import device_quantiles as _dq
        median_pred = _dq(sd, (0.5,)).double().cpu().numpy()[:, :, 0]       # numpy's rule (sf_quantiles), NaN-aware
        y = np.asarray(y_test, dtype=np.float64)
        axis = 0 if independent_metrics else None
        ss_res = np.sum((y - mean_pred) ** 2, axis=axis)
        ss_tot = np.sum((y - np.mean(y)) ** 2, axis=axis)
        pit = self.calculate_PIT(X_test, y, samples=sd, posteriors=posteriors)
        dpit_max = np.max(np.abs(pit - np.linspace(0, 1, len(pit))))
        metrics = {"MSE": np.mean((y - mean_pred) ** 2, axis=axis),
                   "RMSE": np.sqrt(np.mean((y - mean_pred) ** 2, axis=axis)),
                   "mean_ae": np.mean(np.abs(y - mean_pred), axis=axis),
                   "median_ae": np.median(np.abs(y - median_pred), axis=axis),
                   "R_squared": 1 - (ss_res / ss_tot),
                   "RMSE_norm": np.sqrt(np.mean((y - mean_pred) ** 2, axis=axis)) / np.std(y),
                   "mean_ae_norm": np.mean(np.abs(y - mean_pred), axis=axis) / np.std(y),
                   "log_dpit_max": float(-0.5 * np.log(dpit_max))}
        try:
            metrics["mean_log_prob"] = float(np.mean(self.log_prob(X_test, y_test, posteriors=posteriors)))
        except Exception:
            pass
        metrics = {k: (v.tolist() if isinstance(v, np.ndarray) else float(v)) for k, v in metrics.items()}
        if verbose:
            for k, v in metrics.items():
                logger.info(f"{k}: {v}")
        self.last_metrics = metrics
        return metrics

    def calculate_PIT(self, X: np.ndarray, y: np.ndarray, num_samples: int = 1000, posteriors=None,
                      samples=None, seed: Optional[int] = None) -> np.ndarray:
        """Sorted, max-normalised probability integral transform values, one per row
        (ref: sbi_runner.py:7128-7160: ``mean(samples[i] < y[i])`` over draws and parameters).  The draws are
        ranked on the device (``sf_pit_ranks``)."""
        from .features import pit_ranks
        posteriors = posteriors if posteriors is not None else self.posteriors
        if samples is None:
            sd = posteriors.sample_catalogue(torch.as_tensor(np.asarray(X, dtype=np.float32)), num_samples, seed)
        elif isinstance(samples, torch.Tensor):
            sd = samples.to(self.device).float()
        else:
            sd = torch.as_tensor(np.asarray(samples, dtype=np.float32)).to(self.device)
        y2 = np.asarray(y, dtype=np.float32).reshape(sd.shape[0], -1)
        n_valid = torch.isfinite(sd).sum(1).double().cpu().numpy()
        ranks = pit_ranks(sd, torch.as_tensor(y2)).double().cpu().numpy()
        # reference semantics: NaN draws compare False but stay in the denominator
        pit = np.nan_to_num(ranks * n_valid, nan=0.0).sum(1) / (sd.shape[1] * sd.shape[2])
        pit = np.sort(pit)
        return pit / pit[-1]
