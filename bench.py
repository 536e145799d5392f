#!/usr/bin/env python
"""bench.py -- headline benchmark of the amortised-posterior flow path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): NPE MAF (5 transforms, 50 hidden) on the 10k-galaxy 10-filter
NIRCam-like mock; one "step" = ``sample_posterior`` over the 2 000-galaxy test catalogue with
1 000 accepted draws per galaxy (the reference's published benchmark loop, ref:
src/synference/sbi_runner.py:6438-6442, S=1000 as in examples/paper/model_testing.ipynb:1543-1554),
prior-box rejection included, driven by the library's own sampler (sf_flow_sample: per-galaxy context table, dense
round 0, retry rounds; round 0 is bracketed by HIP events on the launch stream inside the library and read back
through sf_flow_sample_stats for the roofline).  value = accepted posterior samples / s over all ranks (each rank
owns its own 2 000-galaxy shard: weak scaling, no data-path collective).  The flow-train theta.x pairs/s leg
(forward+backward+RCCL all-reduce+clip+Adam) is timed right after with the same barrier protocol and
reported in the "train" object of the same JSON line.

Inputs are synthetic (synference_amd/synthetic.py) and resident in HBM before the
timed region; weights are random-init + a short seeded warm-up fit so the posterior is non-trivial.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic work per unit, SURVEY.md 8(d): mask-aware MACs x2
WORKLOADS = {
    # BASELINE configs[1] (the default, quoted metric): MAF 5x50 on the 10k-galaxy 10-filter mock
    "maf_cfg2": dict(kind="maf", D=5, C=10, K=10, n_lib=10_000, galaxies=2000, f_draw=175_150.0, f_gal=5_000.0,
                     f_lp=40_030.0, label="BASELINE configs[1]: NPE MAF T=5 H=50 on 10k-galaxy 10-filter mock"),
    # BASELINE configs[2]: NSF (8 bins) on the 100k-galaxy 20-filter mock
    "nsf_cfg3": dict(kind="nsf", D=8, C=20, K=8, n_lib=100_000, galaxies=20000, f_draw=148_640.0, f_gal=30_000.0,
                     f_lp=178_640.0, label="BASELINE configs[2]: NPE NSF T=5 H=50 K=8 on 100k-galaxy 20-filter mock"),
    # the reference's own production model (examples/sbi/configs/best_params.yaml: NSF, 15 transforms, 69 hidden;
    # the model behind the published 0.047 s/object H100 timing) on the cfg3-shaped mock
    "nsf_prod": dict(kind="nsf", D=8, C=20, K=10, n_lib=100_000, galaxies=1000, f_draw=0.0, f_gal=0.0, f_lp=0.0,
                     H=69, T=15, label="reference production NSF (T=15, H=69, K=10) on the 20-filter mock"),
}
PEAK_FP32_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 MFMA (= vector) dense peak


def executed_mfma_flops_per_draw(d):
    """FLOPs the MAF sampler kernel actually issues on the MFMA pipe per draw (dense padded tiles of
    the incremental inverse): group-steps x 4 MFMAs x 32x32x2 MACs x 2 / 32 samples."""
    D, T, NB, HT = d["D"], d["T"], d["NB"], d["HT"]
    if d["kind"] == 1:  # NSF: conditioner + spline head per transform (context products come from the galaxy table)
        steps = HT * d["nGu"] + NB * (2 * HT * d["nGh"])
        d_tr = [(D - (t & 1) + 1) // 2 for t in range(T)]
        heads = sum(((dt + 1) // 2) * d["PT"] * d["nGh"] for dt in d_tr)
        return (T * steps + heads) * 4 * (32 * 32 * 2) * 2 / 32.0
    if d.get("m16_ok") and not d["hidden_bf16"]:
        # 16-row incremental inverse (sf_maf16.hip): per pass p>=2 one 16-row tile of every layer:
        # 4 MFMAs for W0 u, 4 per input tile <= the pass's tile per hidden block; 16x16x4 MACs x 2 / 16 draws
        lo = d.get("g16_lo", d["g16_tile"])   # groups that straddle tiles recompute every tile they touch
        n = sum((d["g16_tile"][p - 1] - lo[p - 1] + 1) * (4 + NB * 4 * (d["g16_tile"][p - 1] + 1)) for p in range(2, D + 1))
        return T * n * (16 * 16 * 4) * 2 / 16.0
    if d["inc_ok"] and NB <= 2:
        steps = HT * d["nGc"]                                           # hoisted context product
        steps += sum(d["nGu"] + NB * d["g_kend"][p - 1] for p in range(2, D + 1))   # one hidden tile per pass
        # (the two head rows per pass are VALU dot products, not MFMA)
    else:
        steps = D * (HT * (d["nGu"] + d["nGc"]) + NB * sum(d["mt_kend"][:HT]) + d["nGh"])
    return T * steps * 4 * (32 * 32 * 2) * 2 / 32.0


def pmc_traffic():
    """HBM bytes per dense round-0 launch from the committed rocprofv3 PMC passes (profiles/), collected
    with the same command; None when no summary is committed."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_summary.json")))
    if not files:
        return None, None
    with open(files[-1]) as fh:
        d = json.load(fh)
    return d.get("hbm_bytes_per_launch"), os.path.basename(files[-1])


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="maf_cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--galaxies", type=int, default=0, help="test-catalogue rows per GPU (0 = workload default)")
    ap.add_argument("--draws", type=int, default=1000)
    ap.add_argument("--train-batch", type=int, default=16384, help="per-GPU training batch of the train leg")
    ap.add_argument("--train-steps", type=int, default=0, help="0 = same as --steps")
    ap.add_argument("--fit-steps", type=int, default=4000, help="untimed seeded warm-up fit")
    ap.add_argument("--fit-lr", type=float, default=2e-3)
    ap.add_argument("--hidden-bf16", action="store_true",
                    help="opt-in bf16 MFMA operands for the hidden HxH layers of the sampler (BASELINE configs[4])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    return ap.parse_args()


def barrier_sync(world):
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(t, world, dev):
    if world == 1:
        return t
    v = torch.tensor([t], dtype=torch.float64, device=dev)
    dist.all_reduce(v, op=dist.ReduceOp.MAX)
    return float(v.item())


def cpu_baseline(spec, flat, x_rows, lo, hi, S, budget_s):
    """Reference-style CPU path: the oracle restatement driven one galaxy at a time, S accepted draws
    each with prior-box rejection (SURVEY.md 8d / BASELINE.md B1), one thread, bounded wall time."""
    from oracle import flows as OF
    from oracle import posterior as OP
    torch.set_num_threads(1)
    ospec = OF.FlowSpec(kind=spec.kind, D=spec.D, C=spec.C, H=spec.H, T=spec.T, K=spec.K, NB=spec.NB,
                        perms=spec.perms, theta_mean=spec.theta_mean.astype(np.float64),
                        theta_std=spec.theta_std.astype(np.float64), x_mean=spec.x_mean.astype(np.float64),
                        x_std=spec.x_std.astype(np.float64))
    fl = torch.as_tensor(flat, dtype=torch.float32)
    times = []
    t_all = time.perf_counter()
    g = 0
    while g < len(x_rows) and (time.perf_counter() - t_all) < budget_s:
        t0 = time.perf_counter()
        OP.sample(ospec, fl, x_rows[g:g + 1], S, 2025 + g, lo, hi, dtype=torch.float32)
        times.append(time.perf_counter() - t0)
        g += 1
    med = float(np.median(times))
    # batched CPU number on all host cores so the GPU ratio is not credited for removing the loop
    ncores = min(16, os.cpu_count() or 1)  # the box's CPU share for one GPU
    torch.set_num_threads(ncores)
    nb = min(len(x_rows), 64)
    OP.sample(ospec, fl, x_rows[:4], S, 6, lo, hi, dtype=torch.float32)  # thread-pool warm-up
    t0 = time.perf_counter()
    OP.sample(ospec, fl, x_rows[:nb], S, 7, lo, hi, dtype=torch.float32)
    tb = time.perf_counter() - t0
    return {"value": S / med, "unit": "samples/s", "cores": 1, "kind": "port",
            "sample": f"{len(times)} galaxies x {S} accepted draws, one galaxy per call (oracle/posterior.py, "
                      f"torch fp32, 1 thread); median {med:.4f} s/object "
                      f"(16-84%: {np.percentile(times, 16):.4f}-{np.percentile(times, 84):.4f})",
            "batched_all_cores": {"value": nb * S / tb, "cores": ncores,
                                  "sample": f"{nb} galaxies x {S} draws in one call"}}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP flow engine has no CPU fallback)")
    # rehearsal knobs (never set by the driver): all ranks on one device / gloo instead of RCCL
    if os.environ.get("SF_BENCH_ONE_DEVICE") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("SF_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    assert world == a.gpus or world == 1, f"--gpus {a.gpus} but WORLD_SIZE={world}"

    from synference_amd.engine import retry_width
    from synference_amd.estimator import build_flow
    from synference_amd.posterior import FlowPosterior
    from synference_amd.priors import prior_from_parameters
    from synference_amd.runner import HipAdam
    from synference_amd.synthetic import make_catalogue

    # ---------------- data: 10k x 10-filter library (train) + per-rank test catalogue, all in HBM
    wl = WORKLOADS[a.workload]
    D, C = wl["D"], wl["C"]
    n_gal = a.galaxies or wl["galaxies"]
    x_lib, th_lib, names = make_catalogue(wl["n_lib"], C, D, seed=1234)
    x_test, th_test, _ = make_catalogue(n_gal, C, D, seed=4321 + rank)
    rs = np.random.RandomState(0)
    idx = rs.permutation(len(x_lib))
    tr = idx[: int(0.8 * len(idx))]
    prior = prior_from_parameters(th_lib[tr], names)
    gen = torch.Generator().manual_seed(42)
    est = build_flow(wl["kind"], th_lib[tr], x_lib[tr], hidden_features=wl.get("H", 50),
                     num_transforms=wl.get("T", 5), num_bins=wl["K"], device=dev, generator=gen).to(dev)
    if a.hidden_bf16:
        import dataclasses
        est.spec = dataclasses.replace(est.spec, hidden_bf16=True)
        est._flow = None
    flow = est.flow
    flat = est.flat.data
    Xtr = torch.as_tensor(x_lib[tr]).to(dev)
    Ttr = torch.as_tensor(th_lib[tr], dtype=torch.float32).to(dev)
    grad = torch.empty_like(flat)
    # ---------------- untimed seeded warm-up fit (identical on every rank)
    opt = HipAdam(flat, lr=1e-3)
    g2 = torch.Generator().manual_seed(7)
    fit_loss = float("nan")
    for it in range(a.fit_steps):
        bi = torch.randint(0, len(tr), (2048,), generator=g2).to(dev)
        opt.desc.lr = a.fit_lr * 0.5 * (1.0 + np.cos(np.pi * it / max(a.fit_steps, 1)))  # cosine decay
        lossv, _ = flow.loss_grad(flat, Ttr[bi], Xtr[bi], 1.0 / 2048, grad_out=grad)
        opt.step(grad, 5.0)
        if it == a.fit_steps - 1:
            fit_loss = float(lossv.mean().item())
    flow.set_params(flat)
    post = FlowPosterior(est, prior.to(dev), seed=2025)
    lo, hi = prior.low.to(dev), prior.high.to(dev)
    X = torch.as_tensor(x_test).to(dev)
    M, S = X.shape[0], a.draws
    out = torch.empty((M, S, D), dtype=torch.float32, device=dev)
    dense_ms = []
    drawn = [0]
    rounds = [0]
    rej0 = [0]

    def sample_step(k, timed):
        """sample_posterior over the catalogue through the library's own sampler (sf_flow_sample): per-galaxy context
        table, dense round 0, retry rounds until every slot is filled -- exactly what FlowPosterior.sample_catalogue
        runs.  The library brackets round 0 with HIP events on this stream (sf_flow_sample_stats)."""
        flow.sample(X, S, lo, hi, seed=1000 + k, max_attempts=64, out=out)
        st = flow.last_sample_stats
        if timed:
            dense_ms.append(st["dense_ms"])
        drawn[0] += st["evaluations"]
        rounds[0] += st["rounds"]
        rej0[0] += st["rejected_round0"]
        return flow.last_unfilled

    for k in range(a.warmup):
        sample_step(0, False)
    drawn[0] = 0
    rounds[0] = 0
    rej0[0] = 0
    barrier_sync(world)
    t0 = time.perf_counter()
    unfilled = 0
    for k in range(a.steps):
        unfilled += sample_step(k, True)
    barrier_sync(world)
    t_samp = max_over_ranks(time.perf_counter() - t0, world, dev)
    k0_ms = float(np.mean(dense_ms))
    accept = 1.0 - rej0[0] / float(a.steps * M * S)
    value = world * a.steps * (M * S - 0) / t_samp
    flops_launch = wl["f_draw"] * M * S   # the per-galaxy part (f_gal * M) runs once per step in the context-table kernel
    achieved = flops_launch / (k0_ms * 1e-3) / 1e12
    traffic, traffic_src = pmc_traffic() if a.workload == "maf_cfg2" and M == 2000 and S == 1000 else (None, None)
    exe_per_draw = executed_mfma_flops_per_draw(flow.describe())
    executed = exe_per_draw * M * S / (k0_ms * 1e-3) / 1e12
    minimal = (wl["f_lp"] if wl["kind"] == "maf" else wl["f_draw"]) * M * S / (k0_ms * 1e-3) / 1e12

    # ---------------- train leg: fwd+bwd (+ all-reduce) + clip + Adam at the per-GPU batch
    tsteps = a.train_steps or a.steps
    B = a.train_batch
    gscale = 1.0 / (B * world)
    opt2 = HipAdam(flat, lr=1e-4)
    bidx = [torch.randint(0, len(tr), (B,), generator=g2).to(dev) for _ in range(4)]

    def train_step(k):
        flow.loss_grad_rows(flat, Ttr, Xtr, bidx[k % 4], gscale, grad)   # row gather fused into the kernel
        if world > 1:
            dist.all_reduce(grad, op=dist.ReduceOp.SUM)
        opt2.step(grad, 5.0)

    for k in range(max(a.warmup, 1)):
        train_step(k)
    barrier_sync(world)
    t0 = time.perf_counter()
    for k in range(tsteps):
        train_step(k)
    barrier_sync(world)
    t_train = max_over_ranks(time.perf_counter() - t0, world, dev)
    pairs = world * tsteps * B / t_train
    # strong scaling: the GLOBAL batch stays at --train-batch, each rank takes 1/world of it
    Bs = max(32, B // world)
    sidx = [torch.randint(0, len(tr), (Bs,), generator=g2).to(dev) for _ in range(4)]
    gs = 1.0 / (Bs * world)

    def strong_step(k):
        flow.loss_grad_rows(flat, Ttr, Xtr, sidx[k % 4], gs, grad)
        if world > 1:
            dist.all_reduce(grad, op=dist.ReduceOp.SUM)
        opt2.step(grad, 5.0)

    for k in range(max(a.warmup, 1)):
        strong_step(k)
    barrier_sync(world)
    t0 = time.perf_counter()
    for k in range(tsteps):
        strong_step(k)
    barrier_sync(world)
    t_strong = max_over_ranks(time.perf_counter() - t0, world, dev)
    pairs_strong = world * tsteps * Bs / t_strong
    # reference-default batch (64): 200 steps in one library call (what train_flow does per epoch on one device)
    order64 = torch.randint(0, len(tr), (200 * 64,), generator=g2).to(dev)
    tl64 = torch.zeros((), dtype=torch.float64, device=dev)
    flow.train_epoch(flat, Ttr, Xtr, order64, 20, 64, 1.0 / 64, opt2.exp_avg, opt2.exp_avg_sq, opt2.desc, opt2.step_count,
                     5.0, opt2.scratch, grad, tl64)
    opt2.step_count += 20
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    flow.train_epoch(flat, Ttr, Xtr, order64, 200, 64, 1.0 / 64, opt2.exp_avg, opt2.exp_avg_sq, opt2.desc, opt2.step_count,
                     5.0, opt2.scratch, grad, tl64)
    opt2.step_count += 200
    torch.cuda.synchronize()
    pairs64 = 200 * 64 / (time.perf_counter() - t0)

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return
    rec = {
        "metric": "posterior samples/sec (accepted, prior-box rejection included)",
        "value": value, "unit": "samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * t_samp / a.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32" if not a.hidden_bf16 else "bf16 hidden-layer MFMA operands, f32 elsewhere",
        "data": "synthetic",
        "config": {"workload": f"{wl['label']}; sample_posterior over {M} test galaxies x {S} draws per GPU",
                   "name": a.workload,
                   "galaxies_per_gpu": M, "draws_per_galaxy": S, "theta_dim": D, "filters": C,
                   "parallelism": f"rows sharded over {world} GPU(s), no collective",
                   "acceptance": accept, "fit_steps": a.fit_steps, "fit_final_loss": fit_loss, "rounds_per_step": rounds[0] / a.steps,
                   "unfilled_slots": unfilled},
        "roofline": {"bound": "mfma", "kernel": (("k_maf_inv16<NB>" if flow.describe().get("m16_ok") and not a.hidden_bf16 else "k_inverse<MafOps<HT,1,LDS>>")
                                           if wl["kind"] == "maf" else "k_inverse<NsfOps<HT,PT,1,LDS>>")
                               + " (dense round 0)",
                     "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / PEAK_FP32_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                     "algorithmic_bytes_per_launch": 4.0 * D * M * S + 4.0 * C * M,
                     "launch_ms": k0_ms, "flops_per_launch": flops_launch,
                     "note": ("achieved = SURVEY 8d contract FLOPs of the REFERENCE algorithm (D full MADE passes per "
                              "transform: 175150 mask-aware FLOP/draw; the 5000 FLOP/galaxy context part runs once per "
                              "step in the context-table kernel) / measured launch time. The kernel produces the same "
                              "draws with an incremental inverse that needs ~1/3 of those FLOPs, so frac can exceed 1: "
                              "it is an algorithmic speed-up, not hardware utilisation. Hardware utilisation is "
                              "executed_mfma_frac (dense padded MFMA FLOPs actually issued / fp32 MFMA peak); fp32 MFMA "
                              "and VALU do not co-execute on gfx950 (SQ_VALU_MFMA_COEXEC_CYCLES = 0, profiles/), so the "
                              "ceiling for this kernel is MFMA-busy + VALU-busy <= 1 (profiles/*_pmc_summary.json). "
                              "minimal_* = mask-aware FLOPs of one MADE evaluation per transform (40030/draw).")
                     if wl["kind"] == "maf" else
                             "achieved uses the SURVEY 8d figure 148640 FLOP/draw (the 30000 FLOP/galaxy context part "
                             "runs once per step in the context-table kernel); executed_* = dense padded MFMA FLOPs "
                             "actually issued",
                     "executed_mfma_flop_per_draw": exe_per_draw, "executed_mfma_tflops": executed,
                     "executed_mfma_frac": executed / PEAK_FP32_TFLOPS,
                     "minimal_algorithm_tflops": minimal, "minimal_algorithm_frac": minimal / PEAK_FP32_TFLOPS},
        "train": {"metric": "flow-train theta.x pairs/sec (fwd+bwd+allreduce+clip+Adam)", "value": pairs,
                  "unit": "pairs/s", "per_gpu_batch": B, "steps": tsteps, "ms_per_step": 1e3 * t_train / tsteps,
                  "achieved_tflops": pairs * 3 * wl["f_lp"] / 1e12,
                  "batch64_pairs_per_s_1gpu": pairs64,
                  "strong_scaling": {"global_batch": Bs * world, "per_gpu_batch": Bs, "value": pairs_strong,
                                     "ms_per_step": 1e3 * t_strong / tsteps}},
    }
    if world == 1 and not a.no_cpu_baseline:
        rec["cpu_baseline"] = cpu_baseline(est.spec, flat.cpu().numpy(), x_test, prior.low.numpy(),
                                           prior.high.numpy(), S, a.cpu_seconds)
    print(json.dumps(rec))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
