"""Child of tests/test_gpu_train_bench_shapes.py: one loss_grad call in a fresh process, so that the environment
variables the library reads once (SF_TRC_NG: 4- or 8-wave cooperative workgroups; SF_NSFC) select the instantiation.

argv: case name, B, output .npz.  Writes loss, grad and the training path the library reports for that batch."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    name, B, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    from cases import make_case
    from synference_amd.engine import HipFlow
    ospec, spec, flat, theta, x = make_case(name, B=B)
    f = HipFlow(spec, "cuda:0")
    loss, grad = f.loss_grad(torch.as_tensor(flat), theta, x, 1.0 / B)
    torch.cuda.synchronize()
    np.savez(out, loss=loss.cpu().numpy(), grad=grad.cpu().numpy(), path=np.int64(f.train_path(B)))


if __name__ == "__main__":
    main()
