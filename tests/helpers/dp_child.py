"""One data-parallel rank of tests/test_gpu_configs.py::test_cfg3_data_parallel_training_on_the_hip_kernels.

Started as a fresh child process (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT / SF_DP_OUT in the environment); all
ranks share cuda:0, so the process group is gloo (RCCL refuses two ranks on one device) and device tensors are staged
through the host by synference_amd.runner's collectives.  Runs the REAL path: train_flow + HipTrainOps (fused-gather
loss_grad kernel, ONE all-reduce of the flat gradient per step, fused clip+Adam) on a BASELINE configs[3]-shaped mock."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    out = os.environ["SF_DP_OUT"]
    n_rows = int(os.environ.get("SF_DP_ROWS", "1000000"))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from synference_amd.estimator import build_flow
    from synference_amd.runner import dist_all_reduce, train_flow
    from synference_amd.synthetic import make_catalogue
    dev = torch.device("cuda:0")
    h5 = os.environ.get("SF_DP_H5")
    if h5:
        # BASELINE configs[3] as the reference stores it: a chunked + deflated library file (Grid/Photometry (C, N),
        # Grid/Parameters (D, N)) read without h5py -- chunks inflated on a thread pool straight into pinned memory
        import time
        from synference_amd.library import load_library_from_hdf5
        t0 = time.perf_counter()
        lib = load_library_from_hdf5(h5, pinned=True, workers=4)
        dt = time.perf_counter() - t0
        x = np.ascontiguousarray(lib["photometry"].T).astype(np.float32)
        theta = np.ascontiguousarray(lib["parameters"].T)
        assert x.shape == (n_rows, 20) and theta.shape == (n_rows, 8)
        if rank == 0:
            print(f"library file: {n_rows} rows read in {dt:.2f} s = {n_rows / dt / 1e6:.2f} M rows/s (hdf5_lite, 4 workers, pinned)",
                  flush=True)
    else:
        x, theta, _ = make_catalogue(n_rows, 20, 8, seed=11)
    X = torch.as_tensor(x).to(dev)
    T = torch.as_tensor(theta, dtype=torch.float32).to(dev)
    est = build_flow("nsf", theta[:20000], x[:20000], hidden_features=50, num_transforms=5, num_bins=8, device=dev,
                     generator=torch.Generator().manual_seed(3)).to(dev)
    flat0 = est.flat.detach().clone()
    # (1) the all-reduced sharded gradient of one GLOBAL batch
    gb = 8192
    rows = torch.arange(gb, device=dev, dtype=torch.int64) * 97 % n_rows
    mine = rows[rank * (gb // world):(rank + 1) * (gb // world)].contiguous()
    g = torch.empty_like(flat0)
    est.flow.loss_grad_rows(flat0, T, X, mine, 1.0 / gb, g)
    dist_all_reduce(g)
    # (2) a short data-parallel training run; rank 1 starts from different weights and a different seed (None -> time):
    # the runner must make both irrelevant
    if rank == 1:
        with torch.no_grad():
            est.flat.add_(0.01)
    summary = train_flow(est, T, X, batch_size=4096, learning_rate=1e-3, validation_fraction=0.1, stop_after_epochs=50,
                         max_num_epochs=1, seed=None if rank else 1234, log_every=0)
    torch.save({"flat": est.flat.detach().cpu(), "grad": g.cpu(), "flat0": flat0.cpu(), "summary": summary,
                "rows": rows.cpu()}, os.path.join(out, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
