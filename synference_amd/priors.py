"""Box priors with the reference's names and predicates.

Mirrors ref: src/synference/custom_runner.py:971-1207 (``Interval``, ``CustomUniform``,
``CustomIndependentUniform``) for the part the hot path consumes: the accept/reject predicate
``low <= v <= high`` (custom_runner.py:982-987), ``log_prob`` = log(lb*ub) - log(high-low) with a
half-open upper bound (1102-1108), ``rsample`` (1096-1100), and the min/max-of-training-theta
construction of ``SBI_Fitter.create_priors`` (ref: src/synference/sbi_runner.py:3498-3557).
The named-parameter diagnostics (custom_runner.py:1131-1186) are reduced to ``acceptance_report``.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch
from torch.distributions import Distribution, constraints


class Interval(constraints.Constraint):
    """``[lower_bound, upper_bound]`` -- closed on both sides like the reference's check()."""

    def __init__(self, lower_bound, upper_bound):
        self.lower_bound = lower_bound
        self.upper_bound = upper_bound
        super().__init__()

    def check(self, value):
        return (self.lower_bound <= value) & (value <= self.upper_bound)

    def __repr__(self):
        return f"Interval(lower_bound={self.lower_bound}, upper_bound={self.upper_bound})"


class CustomIndependentUniform(Distribution):
    """Independent uniform box over D named parameters (event_shape = (D,))."""

    arg_constraints = {}
    has_rsample = True

    def __init__(self, low, high, name_list: Optional[Sequence[str]] = None, device="cpu",
                 verbose: bool = False, validate_args=None):
        self.low = torch.as_tensor(np.asarray(low, dtype=np.float32) if not torch.is_tensor(low) else low,
                                   dtype=torch.float32, device=device).reshape(-1)
        self.high = torch.as_tensor(np.asarray(high, dtype=np.float32) if not torch.is_tensor(high) else high,
                                    dtype=torch.float32, device=device).reshape(-1)
        if self.low.shape != self.high.shape:
            raise ValueError("low and high must have the same shape")
        if not bool((self.high > self.low).all()):
            raise ValueError("every prior range must be non-empty (high > low)")
        D = self.low.numel()
        self.name_list = list(name_list) if name_list is not None else [f"theta_{i}" for i in range(D)]
        if len(self.name_list) != D:
            raise ValueError(f"Length of name_list ({len(self.name_list)}) must match the number of "
                             f"parameters ({D}).")
        self.verbose = verbose
        super().__init__(batch_shape=torch.Size(), event_shape=torch.Size([D]), validate_args=False)

    @property
    def support(self):
        return constraints.independent(Interval(self.low, self.high), 1)

    @property
    def mean(self):
        return (self.high + self.low) / 2

    @property
    def stddev(self):
        return (self.high - self.low) / (12 ** 0.5)

    def rsample(self, sample_shape=torch.Size()):
        shape = torch.Size(sample_shape) + self.low.shape
        return self.low + torch.rand(shape, dtype=self.low.dtype, device=self.low.device) * (self.high - self.low)

    def sample(self, sample_shape=torch.Size()):
        with torch.no_grad():
            return self.rsample(sample_shape)

    def log_prob(self, value):
        value = torch.as_tensor(value, dtype=self.low.dtype, device=self.low.device)
        lb = self.low.le(value).type_as(self.low)
        ub = self.high.gt(value).type_as(self.low)
        return (torch.log(lb * ub) - torch.log(self.high - self.low)).sum(-1)

    def to(self, device):
        return CustomIndependentUniform(self.low.to(device), self.high.to(device), self.name_list, device=device,
                                        verbose=self.verbose)

    def acceptance_report(self, value: torch.Tensor) -> str:
        ok = (self.low <= value) & (value <= self.high)
        lines = []
        for i, n in enumerate(self.name_list):
            bad = int((~ok[..., i]).sum())
            if bad:
                lines.append(f"  - Parameter '{n}' (support [{self.low[i]:.2f}, {self.high[i]:.2f})): "
                             f"{bad}/{ok[..., i].numel()} samples are out of support.")
        tot = float(ok.all(-1).float().mean()) * 100
        lines.append(f"  - In total {tot:.2f}% samples are within support across all parameters.")
        return "\n".join(lines)


def prior_from_parameters(theta: np.ndarray, names: Optional[List[str]] = None, override: Optional[dict] = None,
                          extend_pc: float = 0.0, device="cpu") -> CustomIndependentUniform:
    """``SBI_Fitter.create_priors``: per-parameter min/max of the training theta
    (ref: sbi_runner.py:3498-3534), optional overrides and percentage extension."""
    theta = np.asarray(theta)
    D = theta.shape[1]
    names = list(names) if names is not None else [f"theta_{i}" for i in range(D)]
    override = override or {}
    low, high = [], []
    for i, n in enumerate(names):
        if n in override:
            lo, hi = override[n]
        else:
            pmin, pmax = float(np.min(theta[:, i])), float(np.max(theta[:, i]))
            lo, hi = pmin, pmax
            if extend_pc > 0.0:
                ext = (pmax - pmin) * extend_pc / 100.0
                lo, hi = pmin - ext, pmax + ext
                if lo < 0 and pmin >= 0:
                    lo = 0.0
                if np.isclose(pmax, 1.0, atol=0.05) and hi > 1.0:
                    hi = 1.0
        if np.isnan([lo, hi]).any():
            raise ValueError(f"NAN value found in prior range for parameter '{n}'.")
        if lo == hi:
            raise ValueError(f"Prior range for parameter '{n}' is zero ({lo} == {hi}). "
                             "Please provide a non-zero range.")
        low.append(lo)
        high.append(hi)
    return CustomIndependentUniform(np.array(low), np.array(high), names, device=device)
