"""MI355X-native amortised-posterior flow engine with Synference's API surface for that path."""
from .spec import FlowSpec  # noqa: F401
from .embedding import FCN  # noqa: F401
from .estimator import FlowEstimator, build_flow, load_nde_hip  # noqa: F401
from .posterior import EnsemblePosterior, FlowPosterior  # noqa: F401
from .priors import CustomIndependentUniform, Interval, prior_from_parameters  # noqa: F401
from .runner import HIPRunner, NumpyLoader, train_flow  # noqa: F401
from .fitter import SBI_Fitter  # noqa: F401

__all__ = ["FlowSpec", "FCN", "FlowEstimator", "build_flow", "load_nde_hip", "EnsemblePosterior", "FlowPosterior",
           "CustomIndependentUniform", "Interval", "prior_from_parameters", "HIPRunner", "NumpyLoader",
           "train_flow", "SBI_Fitter"]
