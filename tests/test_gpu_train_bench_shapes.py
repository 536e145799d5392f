"""The training-kernel instantiations that bench.py times, against fp64 autograd on the oracle.

bench.py's `train` leg runs batch 16 384 and its `throughput_regime` 131 072 rows per step: the cooperative kernels then
run with 8-wave workgroups (NG = 2) and, above one chunk per resident workgroup, the multi-chunk accumulation loop.  The
other gradient tests stop at 2 085 rows (NG = 1, one chunk per workgroup), so these shapes get their own rows here:
every kernel name `bench.py` prints in `train.kernel` is one this file has compared with the oracle.
ref loop: custom_runner.py:585-618."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from cases import make_case
from oracle import flows as OF
from test_gpu_train import oracle_loss_grad

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def check_against_oracle(ospec, flat, theta, x, loss, grad, tag):
    rloss, rgrad = oracle_loss_grad(ospec, flat, theta, x)
    assert np.abs(loss - rloss).max() < 1e-4, (tag, np.abs(loss - rloss).max())
    denom = np.abs(rgrad).max()
    assert np.abs(grad - rgrad).max() < 2e-4 * denom, (tag, np.abs(grad - rgrad).max() / denom)
    for n, s, o in OF.param_layout(ospec):
        k = int(np.prod(s))
        d = np.abs(grad[o:o + k] - rgrad[o:o + k]).max()
        assert d < 2e-4 * denom + 1e-7, (tag, n, d)


# (name, B, path the library must report: 2 = cooperative kernel with 8-wave workgroups, None = whatever it takes)
BENCH_SHAPES = [
    ("maf_cfg1", 16384, 2),            # bench `train`: k_maf_trainc<5,1,4,2>, one chunk per workgroup
    ("maf_cfg1", 131072 + 37, 2),      # bench `throughput_regime`: NG = 2, 8+ chunks per workgroup, ragged tail
    ("maf_span6", 16384, 2),           # span placement: the eight-slot instantiations (k_maf_trainc<5,1,4,2,8>)
    ("maf_span6", 131072 + 37, 2),
    ("maf_span_h64", 16384, 2),        # (8,12,64): seven degree groups over four tiles
    ("maf_cli", 16384, 2),             # (7,16,64,6): the reference's example CLI (span + T = 6 + two input tiles)
    ("maf_cli", 2048 + 37, 1),
    ("maf_d4", 16384, None),
    ("maf_d4", 131072 + 37, None),
    ("maf_t6", 16384, 2),              # T = 6 (the reference's example CLI): k_maf_trainc<6,1,4,2>
    ("maf_t6", 2048 + 37, 1),
    ("maf_t8", 16384 + 5, 2),          # T = 8: k_maf_trainc<8,1,3,2>
    ("nsf_cfg3", 16384, 3),            # bench --workload nsf_cfg3 `train`: k_nsf_trainc<4,6>, XCD replicas + f32 atomics
    ("nsf_cfg3", 65536 + 37, 3),       # several chunks per workgroup, ragged tail
    ("nsf_k10", 4096 + 5, 3),          # k_nsf_trainc<3,8>
    ("nsf_h69", 8192 + 3, 3),          # k_nsf_trainc<5,8>: five waves (the production width, bench --workload nsf_prod)
]


@pytest.mark.parametrize("name,B,path", BENCH_SHAPES)
def test_bench_batch_sizes_match_autograd(name, B, path):
    from synference_amd.engine import HipFlow
    ospec, spec, flat, theta, x = make_case(name, B=B)
    f = HipFlow(spec, "cuda:0")
    if path is not None:
        assert f.train_path(B) == path, (name, B, f.train_path(B))
    loss, grad = f.loss_grad(torch.as_tensor(flat), theta, x, 1.0 / B)
    check_against_oracle(ospec, flat, theta, x, loss.cpu().double().numpy(), grad.cpu().double().numpy(),
                         (name, B, f.train_path(B)))
    # same inputs, same bits (the cooperative MAF kernel sums per-workgroup partials in a fixed order at every batch size)
    if f.train_path(B) in (1, 2):
        _, grad2 = f.loss_grad(torch.as_tensor(flat), theta, x, 1.0 / B)
        assert torch.equal(grad, grad2)


@pytest.mark.parametrize("name", ["maf_cfg1", "maf_d4"])
def test_both_workgroup_shapes_on_the_same_rows(name, tmp_path):
    """SF_TRC_NG=1 / 2 forces the 4- / 8-wave instantiation (read once per process: fresh children); both meet the same
    2 085 rows and the same oracle gradient."""
    B = 2048 + 37
    got = {}
    for ng in (1, 2):
        out = tmp_path / f"ng{ng}.npz"
        env = dict(os.environ, SF_TRC_NG=str(ng))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "trc_child.py"), name, str(B), str(out)],
                           env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        got[ng] = np.load(out)
    ospec, spec, flat, theta, x = make_case(name, B=B)
    for ng in (1, 2):
        assert int(got[ng]["path"]) == ng, (name, ng, int(got[ng]["path"]))
        check_against_oracle(ospec, flat, theta, x, got[ng]["loss"].astype(np.float64), got[ng]["grad"].astype(np.float64),
                             (name, "NG", ng))


@pytest.mark.parametrize("name,B", [("nsf_cfg3", 16384), ("maf_cfg1", 16384)])
def test_fixed_point_accumulation_is_right_and_order_independent(name, B, tmp_path):
    """SF_GRAD_ACC=fix (what SF_DETERMINISTIC=1 selects for the NSF kernel at large batch): contributions as 2^-40 fixed point,
    added with int64 atomics into one replica per XCD (csrc/sf_fixacc.h).  Two fresh processes give the same bits, and the
    gradient is the oracle's."""
    got = []
    for k in range(2):
        out = tmp_path / f"fix{k}.npz"
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "helpers", "trc_child.py"), name, str(B), str(out)],
                           env=dict(os.environ, SF_GRAD_ACC="fix"), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        got.append(np.load(out))
    assert np.array_equal(got[0]["grad"], got[1]["grad"]) and np.array_equal(got[0]["loss"], got[1]["loss"])
    ospec, spec, flat, theta, x = make_case(name, B=B)
    check_against_oracle(ospec, flat, theta, x, got[0]["loss"].astype(np.float64), got[0]["grad"].astype(np.float64), (name, "fix"))
