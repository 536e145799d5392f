"""Timing perturbation: every multi-wave kernel of the path once more while an unrelated kernel keeps all CUs busy on another stream.

The cooperative kernels exchange tiles through LDS between barriers; a missing barrier is a race that one schedule may never lose
(round 5: the lampe training kernel lost one only when two workgroups shared a CU).  A foreign workload on a second stream changes which
waves share a SIMD and when they issue -- results must not move: seeded draws and per-row losses bit for bit, gradients bit for bit where
the kernel sums in a fixed order and to rounding where it adds with atomics."""
import numpy as np
import pytest
import torch

from cases import make_case

pytestmark = pytest.mark.gpu


class Noise:
    """Elementwise work on a side stream for as long as the context is open (enqueued ahead: ~150 ms of it)."""

    def __init__(self):
        self.stream = torch.cuda.Stream()
        self.buf = torch.randn(48 * 1024 * 1024, device="cuda")

    def __enter__(self):
        torch.cuda.synchronize()
        with torch.cuda.stream(self.stream):
            for _ in range(60):
                self.buf = torch.sin(self.buf) * 1.0001 + 0.1
        return self

    def __exit__(self, *exc):
        torch.cuda.synchronize()


@pytest.mark.parametrize("name", ["maf_cfg1", "maf_span6", "nsf_cfg3", "nsf_h69", "nsfar_cfg1", "nsfar_two", "mafar_cfg1"])
def test_results_do_not_move_under_a_concurrent_workload(name):
    from synference_amd.engine import HipFlow
    B = 20000
    ospec, spec, flat, theta, x = make_case(name, B=B, spread=0.2)
    f = HipFlow(spec, "cuda:0")
    fl = torch.as_tensor(flat).cuda()
    f.set_params(fl)
    T, X = torch.as_tensor(theta).cuda(), torch.as_tensor(x).cuda()
    free = f.sample(X[:64], 128, seed=3).reshape(-1, spec.D)
    lo = torch.quantile(free, 0.05, dim=0).cpu().numpy().astype(np.float32)
    hi = torch.quantile(free, 0.95, dim=0).cpu().numpy().astype(np.float32)

    def run():
        lp = f.log_prob(T, X).clone()
        loss, grad = f.loss_grad(fl, T, X, 1.0 / B)
        loss, grad = loss.clone(), grad.clone()
        draws = f.sample(X[:1500], 200, lo, hi, seed=77).clone()
        torch.cuda.synchronize()
        return lp, loss, grad, draws

    quiet = run()
    noise = Noise()
    for _ in range(2):
        with noise:
            busy = run()
        assert torch.equal(quiet[0], busy[0]), "log_prob moved"
        assert torch.equal(quiet[1], busy[1]), "per-row losses moved"
        assert torch.equal(quiet[3], busy[3]), "seeded draws moved"
        scale = quiet[2].abs().max().item()
        assert (quiet[2] - busy[2]).abs().max().item() <= 2e-5 * scale, "gradient moved"
        # and the losses are the density kernel's
        assert (busy[1] + busy[0]).abs().max().item() < 2e-4
