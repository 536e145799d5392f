"""Synthetic SED catalogues of the BASELINE shapes: no Synthesizer, no datasets.

theta ~ U(lo, hi) per dimension with the README's 5-parameter box (ref: README.md:86-92; three
U(0,1) dimensions appended for D = 8), and an AB-magnitude-like smooth map
    x = 28 - 2.5 (A t + 0.3 sin(W t + phi)) + 0.1 eps,   t = standardised theta,
clipped at 50 like the reference's feature rule (ref: sbi_runner.py:1932), float32 (N, C) row-major
(ref: sbi_runner.py:2150).
"""
from __future__ import annotations

from typing import Tuple

import numpy as np

BOX5 = [(8.0, 12.0), (0.0, 10.0), (-4.0, -1.4), (0.0, 1000.0), (0.2, 2.0)]
NAMES5 = ["log_mass", "tau_v", "log_zmet", "peak_age", "tau"]


def parameter_box(D: int):
    box = list(BOX5[:D]) + [(0.0, 1.0)] * max(0, D - 5)
    names = list(NAMES5[:D]) + [f"extra_{i}" for i in range(max(0, D - 5))]
    return np.array(box, dtype=np.float64), names


def make_catalogue(N: int, C: int, D: int, seed: int = 1234, noise: float = 0.1,
                   model_seed: int = 1234) -> Tuple[np.ndarray, np.ndarray, list]:
    """Returns (x[N,C] float32, theta[N,D] float64 like the reference's parameter array, names).

    ``model_seed`` fixes the mock forward model (A, W, phi); ``seed`` draws theta and the noise, so
    catalogues made with different ``seed`` values come from the SAME simulator."""
    box, names = parameter_box(D)
    mrng = np.random.default_rng(model_seed)
    A = mrng.normal(size=(C, D))
    W = mrng.normal(size=(C, D))
    phi = mrng.uniform(0, 2 * np.pi, size=C)
    rng = np.random.default_rng([seed, 77])
    theta = rng.uniform(box[:, 0], box[:, 1], size=(N, D))
    mid, half = box.mean(1), (box[:, 1] - box[:, 0]) / np.sqrt(12.0)
    t = (theta - mid) / half
    x = 28.0 - 2.5 * (t @ A.T / np.sqrt(D) + 0.3 * np.sin(t @ W.T + phi)) + noise * rng.normal(size=(N, C))
    x = np.clip(x, None, 50.0).astype(np.float32)
    return np.ascontiguousarray(x), theta, names
