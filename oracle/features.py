"""CPU restatement of the feature transforms (TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py; parity unpinned:
the reference holds no golden vectors for these functions).

Follows, as plain numpy float64:
  * flux_to_asinh / err: ref src/synference/utils.py:647-704 (``f_jy_to_asinh``, ``f_jy_err_to_asinh``)
  * scatter_depths:       ref src/synference/sbi_runner.py:580-691 (``_apply_depths``, 0-D / 1-D depths); the noise
                          comes from the build's Philox stream (seed, stream 2; counter = output row) instead of
                          numpy's global generator
  * pit_ranks:            ref src/synference/sbi_runner.py:7153-7158
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
