"""CPU restatement of the feature transforms (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py; parity unpinned:
the reference holds no golden vectors for these functions).

Follows, as plain numpy float64:
  * flux_to_asinh / err: ref src/synference/utils.py:647-704 (``f_jy_to_asinh``, ``f_jy_err_to_asinh``)
  * scatter_depths:       ref src/synference/sbi_runner.py:580-691 (``_apply_depths``, 0-D / 1-D depths); the noise
                          comes from the build's Philox stream (seed, stream 2; counter = output row) instead of
                          numpy's global generator
  * pit_ranks:            ref src/synference/sbi_runner.py:7153-7158
  * feature_array_ab:     ref src/synference/sbi_runner.py:1566-1589, 1629-1655, 1698-1716, 1783-1834, 1917-1932,
                          1936-2027, 2084-2099 (the AB branch of create_feature_array_from_raw_photometry)
"""
import numpy as np

from . import philox

K = 2.5 * np.log10(np.e)


def flux_to_asinh(flux_njy, f_b_njy, err_njy=None):
    f = np.asarray(flux_njy, dtype=np.float64)
    fb = np.broadcast_to(np.asarray(f_b_njy, dtype=np.float64), f.shape[-1:])
    mag = -K * (np.arcsinh(f / (2 * fb)) + np.log(fb * 1e-9 / 3631.0))
    if err_njy is None:
        return mag
    return mag, K * np.asarray(err_njy, dtype=np.float64) / np.sqrt(f ** 2 + (2 * fb) ** 2)


def scatter_depths(flux, depths, n_scatters=5, depth_sigma=5.0, min_flux_pc_error=0.0, seed=0):
    f = np.asarray(flux, dtype=np.float64)
    N, C = f.shape
    sg = np.asarray(depths, dtype=np.float64) / depth_sigma
    # sigma rows: one for all scatter copies, or one per scatter copy ([n_scatters, C], the 2-D depths case after the
    # caller has drawn a depth set per band and scatter)
    sg = np.broadcast_to(sg, (C,))[None, :] if sg.ndim < 2 else sg
    rep = np.repeat(f, n_scatters, axis=0)
    sigma = np.maximum(np.tile(sg, (N, 1)) if sg.shape[0] == n_scatters and n_scatters > 1 else sg[:1], np.abs(rep) * min_flux_pc_error / 100.0)
    z = philox.normal(seed, np.arange(N * n_scatters, dtype=np.uint64), 0, C, stream=2).astype(np.float64)
    return rep + sigma * z, sigma


def pit_ranks(samples, truth):
    s = np.asarray(samples, dtype=np.float64)
    t = np.asarray(truth, dtype=np.float64)[:, None, :]
    valid = np.isfinite(s).sum(1)
    with np.errstate(invalid="ignore", divide="ignore"):
        return np.where(valid > 0, (s < t).sum(1) / valid, np.nan)


def feature_array_ab(grid_njy, names, normalize_method=None, normalization_unit="AB", scatter_fluxes=0, depths=None,
                     include_errors=False, min_flux_pc_error=0.0, norm_mag_limit=50.0, photometry_to_remove=(),
                     remove_nan_inf=True, drop_dropouts=False, drop_dropout_fraction=1.0, seed=0, asinh_f_b=None):
    """(feature_array [N', F] float64, feature_names, deleted row indices) from a (C, N) library grid in nJy.
    ``asinh_f_b`` (per-filter softening in nJy): the asinh branch (1718-1732) instead of AB magnitudes."""
    names = list(names)
    grid = np.asarray(grid_njy, dtype=np.float64)
    if photometry_to_remove:
        rm = [i for i, n in enumerate(names) if n in photometry_to_remove]
        grid = np.delete(grid, rm, axis=0)
        names = [n for i, n in enumerate(names) if i not in rm]
    phot, err = grid.T, None                                           # (N, C)
    if scatter_fluxes:
        phot, err = scatter_depths(phot, depths, scatter_fluxes, 5.0, min_flux_pc_error, seed)
        err = np.broadcast_to(err, phot.shape) if err.shape[0] == 1 else err
    if asinh_f_b is not None:
        mag = flux_to_asinh(phot, asinh_f_b) if err is None else None
        mag_err = None
        if err is not None:
            mag, mag_err = flux_to_asinh(phot, asinh_f_b, err)
        cols, fnames = [mag], list(names)
        if mag_err is not None and include_errors:
            cols.append(mag_err)
            fnames += [f"unc_{n}" for n in names]
        feat = np.concatenate(cols, axis=1)
        delete = ~np.isfinite(feat).all(axis=1) if remove_nan_inf else np.zeros(len(feat), bool)
        return feat[~delete], fnames, np.nonzero(delete)[0]
    with np.errstate(all="ignore"):
        mag = -2.5 * np.log10(phot / 1000.0) + 23.9                    # :1705 (nJy -> uJy)
        mag_err = None if err is None else 2.5 * err / (np.log(10) * phot)   # :1699-1702
    mag[phot < 0] = norm_mag_limit                                     # :1706, :1714
    norm_col = None
    zero = np.zeros(len(mag), bool)
    if normalize_method is not None:
        j = names.index(normalize_method)
        ref = mag[:, j]
        orig = grid[j] if not scatter_fluxes else np.repeat(grid[j], scatter_fluxes)
        mag = np.delete(mag, j, axis=1) - ref[:, None]                 # :1793-1797 (norm_func = np.subtract)
        if mag_err is not None:
            mag_err = np.delete(mag_err, j, axis=1)
        with np.errstate(all="ignore"):
            if normalization_unit == "AB":
                norm_col = -2.5 * np.log10(orig / 1000.0) + 23.9       # :1807-1818
            else:
                norm_col = np.log10(orig)                              # "log10 nJy": :1827-1830
                norm_col[np.isinf(norm_col)] = 0.0
        zero = ref == 0.0                                              # :1917-1925
        names = [n for i, n in enumerate(names) if i != j]
    mag[mag > norm_mag_limit] = norm_mag_limit                         # :1927-1932
    cols, fnames = [mag], list(names)
    if mag_err is not None and include_errors:
        cols.append(mag_err)
        fnames += [f"unc_{n}" for n in names]
    if norm_col is not None:
        cols.append(norm_col[:, None])
        fnames.append(f"norm_{normalize_method}_{normalization_unit}")
    feat = np.concatenate(cols, axis=1)
    delete = zero.copy()
    if remove_nan_inf:
        delete |= ~np.isfinite(feat).all(axis=1)                       # :2084-2094
    if drop_dropouts:
        nb = len(names)
        delete |= (np.abs(feat[:, :nb]) >= norm_mag_limit).sum(axis=1) >= nb * drop_dropout_fraction   # :2096-2113
    return feat[~delete], fnames, np.nonzero(delete)[0]
