#!/usr/bin/env python
"""bench.py -- headline benchmark of the amortised-posterior flow path on MI355X.

    python bench.py --gpus N --steps K --warmup W

With ``--gpus N > 1`` and no launcher environment (WORLD_SIZE unset) the script starts the N ranks itself
(``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`` as a child process,
before this process touches the GPU) and forwards rank 0's JSON line; started under ``torch.distributed.run`` it is one
rank.  Either way it asserts that the process group really has ``--gpus`` ranks and reports that number.

The default run prints ONE JSON line with: the configs[1] sampling leg (`value`, `roofline`, five timed blocks in `repeats`), the other
arithmetic mode of the sampler beside it (`roofline.split_bf16_sampler`), the API level (`api`: SBI_Fitter.sample_posterior -> host
float64), the per-object call the reference itself times (`per_object_call`), the training leg (`train`, `roofline_train`, `train.dp` =
the same epoch loop with the RCCL all-reduce inside, a one-rank communicator at N = 1), `log_prob`, the configs[4]-shaped catalogue
(`large_catalogue`), the configs[2] workload (`nsf_cfg3`: its own sampling / training / roofline objects) and the CPU baseline.

Workload (BASELINE.json configs[1]): NPE MAF (5 transforms, 50 hidden) on the 10k-galaxy 10-filter NIRCam-like mock;
one "step" = ``sample_posterior`` over the 2 000-galaxy test catalogue with 1 000 accepted draws per galaxy (the
reference's published benchmark loop, ref: src/synference/sbi_runner.py:6438-6442, S=1000 as in
examples/paper/model_testing.ipynb:1543-1554), prior-box rejection included, driven by the library's own sampler
(sf_flow_sample: per-galaxy context table + ONE persistent launch that works first attempts and retries to the end; the
launch is bracketed by HIP events on its stream inside the library and read back through sf_flow_sample_stats).
value = ACCEPTED posterior samples / s over all ranks (slots that end as NaN rows are not counted); each rank owns its
own 2 000-galaxy shard: weak scaling, no data-path collective.  The flow-train theta.x pairs/s leg
(forward+backward+RCCL all-reduce+clip+Adam) is timed right after with the same barrier protocol ("train" object).

roofline.frac is a hardware fraction (<= 1): USEFUL work -- the mask-aware FLOPs of one conditioner evaluation per
transform for every ACCEPTED draw, the least any algorithm needs (rejected draws are overhead) -- over
the measured kernel time, against the fp32 MFMA peak.  The MFMA FLOPs the kernel actually issued (padding and
rejected evaluations included) and the SURVEY 8d contract figure (the reference's D-pass algorithm) are named extras.

Inputs are synthetic (synference_amd/synthetic.py) and resident in HBM before the timed region; weights
are random-init + a short seeded warm-up fit so the posterior is non-trivial.
"""
import argparse
import json
import os
import socket
import subprocess
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
    "nsf_prod": dict(kind="nsf", D=8, C=20, K=10, n_lib=100_000, galaxies=1000, H=69, T=15,
                     label="reference production NSF (T=15, H=69, K=10) on the 20-filter mock"),
}


def nsf_flops(D, C, H, T, K, NB=2):
    """(per draw, per galaxy) FLOPs of the coupling NSF, counted as SURVEY.md 8(d) counts cfg3 (MACs x 2; the context
    products Win_c e(x), Wg e(x) are per galaxy): reproduces 148 640 / 30 000 for D=8, C=20, H=50, T=5, K=8."""
    draw = gal = 0
    for t in range(T):
        d_tr = (D - (t & 1) + 1) // 2
        d_id = D - d_tr
        draw += H * d_id + NB * 2 * H * H + d_tr * (3 * K - 1) * H + D * D     # Win_u, W1 + W2, Wout, LU
        gal += H * C + NB * H * C                                               # Win_c, gates
    return 2.0 * draw, 2.0 * gal


_d, _g = nsf_flops(8, 20, 69, 15, 10)
WORKLOADS["nsf_prod"].update(f_draw=_d, f_gal=_g, f_lp=_d + _g)


def nsfar_flops(D, C, H, T, K):
    """(f_lp, f_draw of the reference algorithm, per galaxy) of zuko's autoregressive NSF: mask-aware MACs x 2 of the hyper-network
    [theta; context] -> H -> H -> D (3K - 1), hidden unit h of type h mod D (csrc/sf_nsfar.hip); zuko inverts a transform with D
    passes over the whole network (f_draw), the context columns of the first layer are the same for every draw of a galaxy."""
    typ = np.arange(H) % D
    per_t = int(typ.sum()) + int((typ[:, None] >= typ[None, :]).sum()) + (3 * K - 1) * sum(int((typ <= r).sum()) for r in range(D))
    ctx = C * H
    return 2.0 * T * (per_t + ctx), 2.0 * T * D * per_t, 2.0 * T * ctx


_lp, _dr, _g = nsfar_flops(5, 10, 50, 5, 8)
# the flow ili.utils.load_nde_lampe(model="nsf") builds (backend="lampe": zuko.flows.NSF, 8 bins) on the cfg1 / cfg2 mock
WORKLOADS["nsfar_cfg2"] = dict(kind="nsf_ar", D=5, C=10, K=8, n_lib=10_000, galaxies=2000, f_lp=_lp, f_draw=_dr, f_gal=_g,
                               label="backend='lampe' NSF (zuko autoregressive, T=5 H=50 K=8) on the 10k-galaxy 10-filter mock")
PEAK_FP32_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 MFMA (= vector) dense peak


def pmc_traffic(tag):
    """HBM bytes per launch from the newest committed rocprofv3 PMC summary (profiles/rNN_pmc_summary.json) collected
    with this command; None when there is none for this kernel."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_summary.json")))
    if not files:
        return None, None
    with open(files[-1]) as fh:
        d = json.load(fh)
    if tag == "sample":
        return d.get("hbm_bytes_per_launch"), os.path.basename(files[-1])
    if tag == "sample_busy":   # issue-slot occupancy of the sampler kernel (SQ counters of the same summary)
        keys = ("mfma_busy_frac", "valu_busy_frac", "wait_frac_of_wave_cycles")
        return ({k: d[k] for k in keys if k in d} or None), os.path.basename(files[-1])
    return d.get("train", {}).get("hbm_bytes_per_launch"), os.path.basename(files[-1])


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
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="N > 1 launch logistics without the workload: ranks rendezvous, all-reduce one flat gradient of the workload's "
                         "size, rank 0 prints the observed world size / backend / all-reduce time (runs without a GPU under "
                         "SF_BENCH_BACKEND=gloo: the CPU test of the 8-rank launch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--skip-api", action="store_true", help="skip the API-level leg (SBI_Fitter.sample_posterior -> host float64)")
    ap.add_argument("--skip-large-catalogue", action="store_true",
                    help="leave out the configs[4]-sized sampling leg (1e5 galaxies x 1000 draws, N = 1 only)")
    ap.add_argument("--skip-throughput-regime", action="store_true",
                    help="leave out the 8 x batch training launches (profiling runs: they share the bench batch's grid size)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the sampling leg of the CPU baseline")
    ap.add_argument("--repeats", type=int, default=5, help="timed blocks of --steps sampling steps (the first one is `value`; all of them: `repeats`)")
    ap.add_argument("--skip-nsf-leg", action="store_true", help="leave out the configs[2] (NSF cfg3) leg of the default run")
    ap.add_argument("--skip-lampe-leg", action="store_true", help="leave out the lampe-backend (nsfar_cfg2) leg of the default run")
    ap.add_argument("--skip-per-object", action="store_true", help="leave out the per-object posterior.sample((S,), x=X[i]) loop")
    ap.add_argument("--skip-dp", action="store_true", help="leave out the RCCL leg of the train object (communicator over the ranks; one rank at N = 1)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------
# N ranks
# ---------------------------------------------------------------------------------------------------------------
def spawn_ranks(a):
    """Start ``--gpus`` ranks as ONE child process tree (torch.distributed.run) and forward its output.  Runs before
    this process has initialised the GPU (nothing here calls into torch.cuda)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    p = subprocess.run(cmd, env=env)
    return p.returncode


def rendezvous_only(a, world, rank, local):
    """What can go wrong before any kernel runs at N = 8 -- the launcher's environment, the rendezvous on 127.0.0.1, the
    process group, the flat-gradient all-reduce and the max-over-ranks timing -- exercised without the workload."""
    backend = os.environ.get("SF_BENCH_BACKEND", "nccl")
    use_gpu = torch.cuda.is_available() and backend == "nccl"
    dev = torch.device("cuda", 0 if os.environ.get("SF_BENCH_ONE_DEVICE") == "1" else local) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == a.gpus, (dist.get_world_size(), a.gpus)
    from synference_amd.spec import FlowSpec, num_params
    wl = WORKLOADS[a.workload]
    P = num_params(FlowSpec(kind=wl["kind"], D=wl["D"], C=wl["C"], H=wl.get("H", 50), T=wl.get("T", 5), K=wl["K"]))
    g = torch.full((P,), float(rank + 1), dtype=torch.float32, device=dev)
    us = None
    if world > 1:
        for _ in range(3):
            dist.all_reduce(g.clone())
        if use_gpu:
            torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(10):
            h = g.clone()
            dist.all_reduce(h)
        if use_gpu:
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        v = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(v, op=dist.ReduceOp.MAX)
        us = float(v.item()) * 1e6
        ok = bool(torch.allclose(h, torch.full_like(h, world * (world + 1) / 2.0)))
    else:
        ok = True
    if rank == 0:
        print(json.dumps({"rendezvous_only": True, "n_gpus": a.gpus, "rccl_ranks": dist.get_world_size() if world > 1 else 1,
                          "backend": dist.get_backend() if world > 1 else "none", "flat_gradient_floats": P,
                          "allreduce_us": us, "allreduce_sum_correct": ok, "device": str(dev)}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0 if ok else 1


def all_reduce_(t, op, gloo):
    if gloo and t.device.type == "cuda":   # rehearsal backend: stage device tensors through the host
        h = t.cpu()
        dist.all_reduce(h, op=op)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op)


def barrier_sync(world):
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(t, world, dev, gloo):
    if world == 1:
        return t
    v = torch.tensor([t], dtype=torch.float64, device=dev)
    all_reduce_(v, dist.ReduceOp.MAX, gloo)
    return float(v.item())


# ---------------------------------------------------------------------------------------------------------------
# CPU baseline (rank 0, N = 1 only): the oracle driven the way the reference drives sbi
# ---------------------------------------------------------------------------------------------------------------
def host_cpu():
    """(CPU model, logical cores of the host, cores this process may actually use): the last is the smaller of the
    affinity mask and the cgroup CPU quota -- on a shared GPU box the mask shows every core of the host while the quota
    grants a share, and a thread pool sized by the mask would mostly wait for its quota."""
    model, cores = "unknown", os.cpu_count() or 1
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else cores
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:            # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = fh.read().split()[:2]
            if q != "max":
                quota = int(q) / float(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
                q, per = int(fq.read()), int(fp.read())
                if q > 0:
                    quota = q / float(per)
        except (OSError, ValueError):
            pass
    if quota is not None:
        usable = max(1, min(usable, int(quota + 0.5)))
    return model, cores, usable, quota


def cpu_baseline(spec, flat, x_rows, th_rows, lo, hi, S, budget_s, train_theta, train_x):
    """BASELINE.md section 3, B1-B4, on a bounded sample (about 25-30 s of CPU work in total):
    B1 per-object sampling, one galaxy per call, 1 thread (mirrors sbi_runner.py:6438-6442; the statistic of log_times);
    B2 batched sampling on all usable cores; B3 log_prob per row (sbi_runner.py:7193-7196, raw density) and batched;
    B4 the epoch loop of custom_runner.py:580-618 (Adam, clip 5.0) at batch 64 and at the GPU leg's batch."""
    from oracle import flows as OF
    from oracle import posterior as OP
    model, cores, usable, quota = host_cpu()
    usable = min(usable, 32)   # torch CPU ops of this size stop scaling long before that
    note(f"CPU baseline on {model}: host {cores} logical cores, cgroup quota {quota}, using {usable} threads")
    ospec = OF.FlowSpec(kind=spec.kind, D=spec.D, C=spec.C, H=spec.H, T=spec.T, K=spec.K, NB=spec.NB,
                        tail_bound=spec.tail_bound, ar_slope=spec.ar_slope,
                        perms=spec.perms, theta_mean=spec.theta_mean.astype(np.float64),
                        theta_std=spec.theta_std.astype(np.float64), x_mean=spec.x_mean.astype(np.float64),
                        x_std=spec.x_std.astype(np.float64))
    fl = torch.as_tensor(flat, dtype=torch.float32)
    # ---- B1: the reference's own schedule -- sbi's batch accept/reject loop, one galaxy per call
    torch.set_num_threads(1)
    gen = torch.Generator().manual_seed(2025)
    times, g = [], 0
    t_all = time.perf_counter()
    while g < len(x_rows) and (time.perf_counter() - t_all) < budget_s:
        t0 = time.perf_counter()
        OP.accept_reject_sample(ospec, fl, x_rows[g], S, lo, hi, gen)
        times.append(time.perf_counter() - t0)
        g += 1
    med = float(np.median(times))
    note(f"B1 done: {len(times)} galaxies, median {med:.4f} s/object")
    # ---- B3 (1 thread): per row, then batched
    xt, tt = torch.as_tensor(x_rows), torch.as_tensor(th_rows, dtype=torch.float32)
    n_rows, t0 = 0, time.perf_counter()
    with torch.no_grad():
        while n_rows < len(x_rows) and time.perf_counter() - t0 < 3.0:
            OF.log_prob(ospec, fl, tt[n_rows:n_rows + 1], xt[n_rows:n_rows + 1])
            n_rows += 1
    lp_row = n_rows / (time.perf_counter() - t0)
    with torch.no_grad():
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 2.0:
            OF.log_prob(ospec, fl, tt, xt)
            reps += 1
    lp_batched_1 = reps * len(xt) / (time.perf_counter() - t0)

    # ---- B4 (1 thread, then all cores): Adam + clip_grad_norm_(5.0), loss = mean(-log_prob) via torch.autograd
    def train_rate(batch, seconds):
        p = torch.nn.Parameter(fl.clone())
        opt = torch.optim.Adam([p], lr=1e-4)
        rs = np.random.RandomState(0)
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            idx = rs.randint(0, len(train_x), size=batch)
            opt.zero_grad(set_to_none=True)
            loss = -OF.log_prob(ospec, p, torch.as_tensor(train_theta[idx], dtype=torch.float32),
                                torch.as_tensor(train_x[idx])).mean()
            loss.backward()
            torch.nn.utils.clip_grad_norm_([p], 5.0)
            opt.step()
            n += batch
        return n / (time.perf_counter() - t0)

    note("B3 (1 thread) done")
    tr64_1 = train_rate(64, 3.0)
    note("B4 (1 thread) done")
    # ---- all usable cores
    torch.set_num_threads(usable)
    # B2: the whole rejection loop batched over galaxies (one proposal per open slot per round, torch fp32 on all usable
    # cores) -- what removing the reference's per-galaxy Python loop alone buys on the CPU
    nb = min(len(x_rows), 256)
    gen_b = torch.Generator().manual_seed(77)
    OP.accept_reject_sample_batched(ospec, fl, x_rows[:8], S, lo, hi, gen_b)  # thread-pool warm-up
    t0 = time.perf_counter()
    sb, drawn_b = OP.accept_reject_sample_batched(ospec, fl, x_rows[:nb], S, lo, hi, gen_b)
    tb = time.perf_counter() - t0
    filled_b = int(np.isfinite(sb).all(-1).sum())
    with torch.no_grad():
        t0 = time.perf_counter()
        reps = 0
        while time.perf_counter() - t0 < 2.0:
            OF.log_prob(ospec, fl, tt, xt)
            reps += 1
    lp_batched_all = reps * len(xt) / (time.perf_counter() - t0)
    note("B2/B3 (all cores) done")
    tr64_all = train_rate(64, 2.0)
    trbig_all = train_rate(16384, 4.0)
    return {"value": S / med, "unit": "samples/s", "cores": 1, "kind": "port",
            "cpu_model": model, "host_cores": cores, "usable_cores": usable, "cgroup_cpu_quota": quota,
            "sample": f"{len(times)} galaxies x {S} accepted draws, one galaxy per call (oracle/posterior.py "
                      f"accept_reject_sample = sbi's batch loop, torch fp32, 1 thread); median {med:.4f} s/object "
                      f"(16-84%: {np.percentile(times, 16):.4f}-{np.percentile(times, 84):.4f})",
            "batched_all_cores": {"value": filled_b / tb, "unit": "samples/s", "cores": usable,
                                  "sample": f"{nb} galaxies x {S} accepted draws in one call (oracle/posterior.py "
                                            f"accept_reject_sample_batched: every round proposes one draw per open slot for "
                                            f"all galaxies at once; {drawn_b} proposals for {filled_b} accepted draws)"},
            "log_prob": {"per_row_1thread": {"value": lp_row, "unit": "rows/s", "cores": 1,
                                             "sample": f"{n_rows} rows, one row per call (sbi_runner.py:7193-7196 loop, raw density)"},
                         "batched_1thread": {"value": lp_batched_1, "unit": "rows/s", "cores": 1,
                                             "sample": f"{len(xt)} rows per call"},
                         "batched_all_cores": {"value": lp_batched_all, "unit": "rows/s", "cores": usable,
                                               "sample": f"{len(xt)} rows per call"}},
            "train": {"batch64_1thread": {"value": tr64_1, "unit": "pairs/s", "cores": 1,
                                          "sample": "3 s of Adam steps at batch 64 (custom_runner.py:580-618 loop, torch autograd on the oracle)"},
                      "batch64_all_cores": {"value": tr64_all, "unit": "pairs/s", "cores": usable, "sample": "2 s at batch 64"},
                      "batch16384_all_cores": {"value": trbig_all, "unit": "pairs/s", "cores": usable,
                                               "sample": "4 s at batch 16384"}}}


def note(msg):
    """progress line on stderr (rank 0): long runs must show signs of life"""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a))   # (children start before this process touches the GPU; their exit code is ours)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if a.rendezvous_only:
        return rendezvous_only(a, world, rank, local)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP flow engine has no CPU fallback)")
    # rehearsal knobs (never set by the driver): all ranks on one device / gloo instead of RCCL
    if os.environ.get("SF_BENCH_ONE_DEVICE") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    backend = os.environ.get("SF_BENCH_BACKEND", "nccl")
    gloo = backend != "nccl"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        assert dist.get_world_size() == a.gpus, (dist.get_world_size(), a.gpus)
    rec = run_workload(a, world, rank, dev, gloo)
    # ---------------- BASELINE configs[2] beside configs[1] in the default run: the coupling NSF (8 bins) on the 100k-galaxy
    # 20-filter mock -- sampling, training and both roofline objects, a few steps each (the full-length line is
    # `--workload nsf_cfg3`)
    if a.workload == "maf_cfg2" and not a.skip_nsf_leg and not a.hidden_bf16 and not a.galaxies:
        b = argparse.Namespace(**vars(a))
        b.workload, b.steps, b.warmup, b.train_steps, b.fit_steps, b.repeats = "nsf_cfg3", 4, 2, 20, 4000, 1
        b.skip_api = b.skip_large_catalogue = b.no_cpu_baseline = b.skip_per_object = b.skip_dp = True
        b.skip_throughput_regime = True
        note("configs[2] leg (NSF cfg3)")
        sub = run_workload(b, world, rank, dev, gloo)
        if rank == 0:
            keep = ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline", "roofline_train")
            rec["nsf_cfg3"] = {k: sub[k] for k in keep}
            rec["nsf_cfg3"]["train"] = {k: sub["train"][k] for k in ("value", "unit", "per_gpu_batch", "steps", "ms_per_step",
                                                                   "batch64_ms_per_step", "batch64_pairs_per_s_1gpu")}
            rec["nsf_cfg3"]["log_prob"] = sub["log_prob"]
    # ---------------- the reference's second backend beside it (row f4): the lampe flow (zuko NSF, 8 bins) on the configs[1] mock -- a
    # few steps of sampling and training with both roofline objects (the full-length line is `--workload nsfar_cfg2`)
    if a.workload == "maf_cfg2" and not a.skip_lampe_leg and not a.hidden_bf16 and not a.galaxies:
        b = argparse.Namespace(**vars(a))
        b.workload, b.steps, b.warmup, b.train_steps, b.repeats = "nsfar_cfg2", 5, 2, 20, 1
        b.skip_api = b.skip_large_catalogue = b.no_cpu_baseline = b.skip_per_object = b.skip_dp = True
        b.skip_throughput_regime = True
        note("lampe leg (autoregressive NSF, cfg2 mock)")
        sub = run_workload(b, world, rank, dev, gloo)
        if rank == 0:
            keep = ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "config", "roofline", "roofline_train")
            rec["lampe_nsfar_cfg2"] = {k: sub[k] for k in keep}
            rec["lampe_nsfar_cfg2"]["train"] = {k: sub["train"][k] for k in ("value", "unit", "per_gpu_batch", "steps", "ms_per_step",
                                                                           "batch64_ms_per_step", "batch64_pairs_per_s_1gpu")}
            rec["lampe_nsfar_cfg2"]["log_prob"] = sub["log_prob"]
    if rank == 0:
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.destroy_process_group()


def run_workload(a, world, rank, dev, gloo):
    """One workload's legs on this rank; returns the record on rank 0 (None elsewhere)."""
    from synference_amd.estimator import build_flow
    from synference_amd.priors import prior_from_parameters
    from synference_amd.runner import HipAdam
    from synference_amd.synthetic import make_catalogue

    # ---------------- data: 10k x 10-filter library (train) + per-rank test catalogue, all in HBM
    wl = WORKLOADS[a.workload]
    D, C = wl["D"], wl["C"]
    n_gal = a.galaxies or wl["galaxies"]
    x_lib, th_lib, names = make_catalogue(wl["n_lib"], C, D, seed=1234)
    # ONE test catalogue of world x n_gal rows, sharded by rank as contiguous row blocks; weak scaling: the
    # catalogue grows with the number of ranks, each rank's share stays at n_gal rows
    x_all, th_all, _ = make_catalogue(n_gal * world, C, D, seed=4321)
    x_test, th_test = x_all[rank * n_gal:(rank + 1) * n_gal], th_all[rank * n_gal:(rank + 1) * n_gal]
    rs = np.random.RandomState(0)
    idx = rs.permutation(len(x_lib))
    tr = idx[: int(0.8 * len(idx))]
    prior = prior_from_parameters(th_lib[tr], names)
    gen = torch.Generator().manual_seed(42)
    est = build_flow("nsf" if wl["kind"] == "nsf_ar" else wl["kind"], th_lib[tr], x_lib[tr], hidden_features=wl.get("H", 50),
                     num_transforms=wl.get("T", 5), num_bins=wl["K"], device=dev, generator=gen,
                     backend="lampe" if wl["kind"] == "nsf_ar" else "sbi").to(dev)
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
    note(f"warm-up fit done (loss {fit_loss:.3f}); sampling {a.warmup}+{a.steps} steps")
    lo, hi = prior.low.to(dev), prior.high.to(dev)
    X = torch.as_tensor(x_test).to(dev)
    M, S = X.shape[0], a.draws
    out = torch.empty((M, S, D), dtype=torch.float32, device=dev)
    kernel_ms, evals, launches, rej0 = [], [0.0], [0], [0]

    def sample_step(k, timed):
        """sample_posterior over the catalogue through the library's own sampler (sf_flow_sample): per-galaxy context
        table + the persistent launch -- exactly what FlowPosterior.sample_catalogue runs (no attempt ceiling)."""
        flow.sample(X, S, lo, hi, seed=1000 + k, out=out)
        st = flow.last_sample_stats
        if timed:
            kernel_ms.append(st["dense_ms"])
            evals[0] += st["evaluations"]
            launches[0] += st["rounds"]
            rej0[0] += st["rejected_round0"]
        return flow.last_unfilled

    for k in range(a.warmup):
        sample_step(0, False)
    barrier_sync(world)
    t0 = time.perf_counter()
    unfilled = 0
    for k in range(a.steps):
        unfilled += sample_step(k, True)
    barrier_sync(world)
    t_samp = max_over_ranks(time.perf_counter() - t0, world, dev, gloo)
    k_ms = float(np.mean(kernel_ms))
    accept = 1.0 - rej0[0] / float(a.steps * M * S)
    if world > 1:
        u = torch.tensor([float(unfilled)], dtype=torch.float64, device=dev)
        all_reduce_(u, dist.ReduceOp.SUM, gloo)
        unfilled_all = int(u.item())
    else:
        unfilled_all = unfilled
    value = (world * a.steps * M * S - unfilled_all) / t_samp
    desc = flow.describe()
    accepted_per_launch = M * S - unfilled / float(a.steps)
    evals_per_launch = evals[0] / float(a.steps)
    # one conditioner evaluation per transform (the coupling NSF's inverse IS one; the autoregressive flows' reference inverse is D)
    f_min = wl["f_lp"] if wl["kind"] in ("maf", "nsf_ar") else wl["f_draw"]
    useful = f_min * accepted_per_launch / (k_ms * 1e-3) / 1e12
    contract = wl["f_draw"] * accepted_per_launch / (k_ms * 1e-3) / 1e12
    default_wl = a.workload == "maf_cfg2" and M == 2000 and S == 1000
    traffic, traffic_src = pmc_traffic("sample") if default_wl else (None, None)
    busy, _ = pmc_traffic("sample_busy") if default_wl else (None, None)

    note(f"sampling done: {1e3 * t_samp / a.steps:.3f} ms/step, kernel {k_ms:.3f} ms, unfilled {unfilled_all}")
    # the same timed block again (same protocol: barrier + synchronize on both sides, max over ranks): run-to-run scatter of the
    # headline.  `value` stays the FIRST block (the contract's K steps); block 0 below is that block.
    block_ms = [1e3 * t_samp / a.steps]
    for rp in range(1, max(1, a.repeats)):
        barrier_sync(world)
        t0 = time.perf_counter()
        for k in range(a.steps):
            sample_step(k + rp * a.steps, False)
        barrier_sync(world)
        block_ms.append(1e3 * max_over_ranks(time.perf_counter() - t0, world, dev, gloo) / a.steps)
    repeats = {"blocks": len(block_ms), "steps_per_block": a.steps, "ms_per_step": block_ms, "min": float(np.min(block_ms)),
               "median": float(np.median(block_ms)), "max": float(np.max(block_ms)),
               "value_at_median": (world * M * S - unfilled_all / a.steps) / (float(np.median(block_ms)) * 1e-3)}
    # the same step in the OTHER arithmetic mode of the samplers' hidden blocks (sf_set_sampler_fp32), quoted beside the default.
    # Default (round 5): a MAF samples in fp32 throughout (k_maf_samp16<.., PREC = 1>: v_mfma_f32_16x16x4_f32), the opt-in fast
    # mode runs the hidden H x H blocks as split-bf16 x3 products; a coupling NSF's default is its split-bf16 sampler image and the
    # other mode is the all-fp32 image.  mode_cost: the same seed drawn in both modes, both sets re-scored by the fp32 density kernel.
    alt_leg = None
    default_split = (not a.hidden_bf16) and wl["kind"] == "nsf" and bool(desc.get("nsf_split_sampler"))
    has_alt = (not a.hidden_bf16) and (bool(desc.get("m16_ok")) if wl["kind"] == "maf" else bool(desc.get("nsf_split_sampler")))
    if has_alt:
        from synference_amd import _lib as _sflib
        alt_mode = 1 if default_split else 0
        _sflib.load().sf_set_sampler_fp32(alt_mode)
        try:
            sample_step(0, False)
            barrier_sync(world)
            t0 = time.perf_counter()
            n32 = max(2, min(a.steps, 5))
            unf32, k32 = 0, []
            for k in range(n32):
                unf32 += sample_step(k, False)
                k32.append(flow.last_sample_stats["dense_ms"])
            barrier_sync(world)
            t32 = max_over_ranks(time.perf_counter() - t0, world, dev, gloo)
            # Draws whose accepted attempt differs (a candidate within rounding of the box edge) are different draws and are
            # counted, not compared.
            ng = min(M, 256)
            oa = torch.empty((ng, S, D), dtype=torch.float32, device=dev)
            ob = torch.empty_like(oa)
            flow.sample(X[:ng], S, lo, hi, seed=4242, out=ob)             # the other mode (switch is on)
            _sflib.load().sf_set_sampler_fp32(-1)
            flow.sample(X[:ng], S, lo, hi, seed=4242, out=oa)             # default
            sig = torch.as_tensor(np.asarray(est.spec.theta_std), dtype=torch.float32, device=dev)
            dth = ((oa - ob).abs() / sig).amax(-1)                        # (ng, S) in units of the parameter std
            same = dth < 1e-3
            xr = X[:ng].repeat_interleave(S, 0)
            lpa = flow.log_prob(oa.reshape(-1, D), xr).reshape(ng, S)
            lpb = flow.log_prob(ob.reshape(-1, D), xr).reshape(ng, S)
            dlp = (lpa - lpb).abs()[same]
            split_cost = {"draws_compared": int(same.sum().item()), "draws_with_a_different_accepted_attempt": int((~same).sum().item()),
                          "max_abs_dlogp": float(dlp.max().item()), "median_abs_dlogp": float(dlp.median().item()),
                          "p999_abs_dlogp": float(torch.quantile(dlp.float(), 0.999).item()) if dlp.numel() < 16_000_000 else None,
                          "max_dtheta_over_sigma": float(dth[same].max().item()),
                          "note": "same seed through the split-bf16 x3 and the all-fp32 sampler kernels; |d log_prob| of the two "
                                  "draws under the fp32 density kernel; north_star tolerance 1e-4"}
            if wl["kind"] == "maf":
                alt_kernel = "k_maf_samp16<.., PREC = 0> (hidden H x H blocks as split-bf16 x3 products on v_mfma_f32_16x16x32_bf16, fp32 accumulate)"
            else:
                alt_kernel = "k_sample_persist<NsfOps<..., BF = 0>> (32-row tiles, v_mfma_f32_32x32x2_f32 everywhere)"
            k32_ms = float(np.mean(k32))
            alt_leg = {"mode": "fp32" if alt_mode == 1 else "split-bf16 x3 hidden blocks (opt-in: sf_set_sampler_fp32(0) / SF_SAMPLER_FP32=0)",
                       "kernel": alt_kernel, "ms_per_step": 1e3 * t32 / n32, "value": (world * n32 * M * S - unf32 * world) / t32,
                       "unit": "samples/s", "steps": n32, "launch_ms": k32_ms,
                       "roofline": {"bound": "mfma", "achieved": f_min * (M * S - unf32 / n32) / (k32_ms * 1e-3) / 1e12, "peak": PEAK_FP32_TFLOPS,
                                    "unit": "TFLOP/s", "frac": f_min * (M * S - unf32 / n32) / (k32_ms * 1e-3) / 1e12 / PEAK_FP32_TFLOPS,
                                    "traffic": None},
                       "split_bf16_cost": split_cost}
        finally:
            _sflib.load().sf_set_sampler_fp32(-1)
        note(f"{alt_leg['mode'].split()[0]} sampler leg: {alt_leg['ms_per_step']:.3f} ms/step")
    # ---------------- API level: the reference's own call, SBI_Fitter.sample_posterior(X, num_samples) -> HOST float64 (N, S, D)
    # (sbi_runner.py:6436-6442; `log_times` statistic of 6461-6469; examples/paper/model_testing.ipynb:1543-1553), and
    # fit_catalogue's quantile-only path (device quantiles, only (N, D, Q) leaves the GPU).  Rank-local (shard=False): every
    # rank fits its own block, like the engine-level leg above.
    api = None
    if not a.skip_api:
        from synference_amd.fitter import SBI_Fitter
        from synference_amd.hostio import usable_cores
        from synference_amd.posterior import EnsemblePosterior, FlowPosterior
        fitter = SBI_Fitter("bench", names, [f"F{i}" for i in range(C)], feature_array=x_lib, parameter_array=th_lib)
        fitter.posteriors = EnsemblePosterior([FlowPosterior(est, prior)], weights=[1.0])
        fitter._prior = prior
        n_api = max(2, min(a.steps, 10))
        for k in range(3):   # (held like the timed loop holds them: the result buffers the steady state alternates between exist)
            arr = fitter.sample_posterior(x_test, num_samples=S, seed=500 + k, shard=False)
        barrier_sync(world)
        t0 = time.perf_counter()
        nan_rows = 0
        for k in range(n_api):
            arr = fitter.sample_posterior(x_test, num_samples=S, seed=1000 + k, shard=False)
        barrier_sync(world)
        t_api = max_over_ranks(time.perf_counter() - t0, world, dev, gloo)
        assert arr.shape == (M, S, D) and arr.dtype == np.float64
        nan_rows = int(np.isnan(arr[:, :, 0]).sum())
        # the draws the API hands over are the engine's (same seed -> same bits, widened)
        flow.sample(X, S, lo, hi, seed=1000 + n_api - 1, out=out)
        same = bool(np.array_equal(arr, out.double().cpu().numpy(), equal_nan=True))
        fitter.sample_posterior(x_test, num_samples=S, seed=77, shard=False, log_times=True)
        tpo = np.asarray(fitter.last_times_per_object)
        import pandas as pd
        df_obs = pd.DataFrame(x_test, columns=list(fitter.feature_names))
        fitter.fit_catalogue(df_obs, num_samples=S, seed=5, append_to_input=False)
        t0 = time.perf_counter()
        n_q = max(2, min(a.steps, 5))
        for k in range(n_q):
            tab = fitter.fit_catalogue(df_obs, num_samples=S, seed=6 + k, append_to_input=False)
        t_q = (time.perf_counter() - t0) / n_q
        api = {"call": "SBI_Fitter.sample_posterior(X, num_samples=%d) -> host float64 (%d, %d, %d)" % (S, M, S, D),
               "value": world * n_api * M * S / t_api, "unit": "samples/s", "ms_per_call": 1e3 * t_api / n_api, "calls": n_api,
               "fraction_of_engine_value": (world * n_api * M * S / t_api) / value,
               "bit_identical_to_engine_draws": same, "nan_draws": nan_rows,
               "log_times": {"median_s_per_object": float(np.median(tpo)), "p16": float(np.percentile(tpo, 16)),
                             "p84": float(np.percentile(tpo, 84)),
                             "note": "the statistic of sbi_runner.py:6461-6469 (the catalogue call is timed in 16 chunks)"},
               "fit_catalogue_quantiles": {"ms_per_call": 1e3 * t_q, "samples_per_s": M * S / t_q, "columns": int(tab.shape[1]),
                                           "note": "16 / 50 / 84 % per parameter reduced on the device; the draws never leave the GPU"},
               "host": {"usable_cores": usable_cores(),
                        "pipeline": "one-member posterior: the sampling kernels write the float64 host array themselves (pinned memory mapped "
                                    "into the device's address space, sf_flow_set_sample_output_f64: the draws cross PCIe while the launch "
                                    "runs); otherwise sf_copy_to_host_f64 (fp32 D2H pieces into a pinned ring, widened by host threads)"}}
        note(f"API leg: {api['ms_per_call']:.2f} ms per sample_posterior call = {api['fraction_of_engine_value']:.2f} of the engine-level "
             f"value; fit_catalogue quantiles {1e3 * t_q:.2f} ms")
    # ---------------- the published statistic itself: one posterior.sample((S,), x=X[i]) call PER OBJECT, host array out --
    # the loop of sbi_runner.py:6438-6442 (sampler.sample = posterior.sample(...).detach().cpu().numpy(), ili DirectSampler), timed
    # per object as `log_times` times it (6461-6469: median and 16th-84th percentile).  The reference publishes 0.047 s per object
    # on an H100 and 0.069 s on a CPU for its production NSF (examples/paper/obs.ipynb:316): a different model, so no ratio is
    # formed here -- `--workload nsf_prod` times that shape.
    per_object = None
    if not a.skip_per_object:
        from synference_amd.posterior import FlowPosterior as _FP
        post1 = _FP(est, prior)
        n_obj = min(M, 200)
        samples_po = np.zeros((n_obj, S, D))
        for i in range(5):
            post1.sample((S,), x=x_test[i], seed=10 + i).detach().cpu().numpy()
        torch.cuda.synchronize()
        times_po = []
        for i in range(n_obj):
            st_t = time.time()
            samples_po[i] = post1.sample((S,), x=x_test[i], seed=100 + i).detach().cpu().numpy()
            times_po.append(time.time() - st_t)
        per_object = {"call": "posterior.sample((%d,), x=X[i]).detach().cpu().numpy() per object (sbi_runner.py:6438-6442)" % S,
                      "objects": n_obj, "median_s_per_object": float(np.median(times_po)),
                      "p16": float(np.percentile(times_po, 16)), "p84": float(np.percentile(times_po, 84)),
                      "samples_per_s": S / float(np.median(times_po)), "nan_draws": int(np.isnan(samples_po).any(-1).sum()),
                      "note": "one library call per object: context table of one row + persistent launch + find / resolve rounds + "
                              "20 kB D2H + host sync; the catalogue call above amortises all of that over 2 000 objects"}
        note(f"per-object call: median {1e3 * per_object['median_s_per_object']:.3f} ms")
    # ---------------- train leg: fwd+bwd (+ all-reduce) + clip + Adam at the per-GPU batch
    tsteps = a.train_steps or a.steps
    B = a.train_batch
    gscale = 1.0 / (B * world)
    opt2 = HipAdam(flat, lr=1e-4)
    bidx = [torch.randint(0, len(tr), (B,), generator=g2).to(dev) for _ in range(4)]

    def train_step(k):
        flow.loss_grad_rows(flat, Ttr, Xtr, bidx[k % 4], gscale, grad)   # row gather fused into the kernel
        if world > 1:
            all_reduce_(grad, dist.ReduceOp.SUM, gloo)
        opt2.step(grad, 5.0)

    for k in range(max(a.warmup, 1)):
        train_step(k)
    barrier_sync(world)
    t0 = time.perf_counter()
    for k in range(tsteps):
        train_step(k)
    barrier_sync(world)
    t_plain = max_over_ranks(time.perf_counter() - t0, world, dev, gloo)
    t_train, graph_used = t_plain, False
    if world == 1 and tsteps >= 4:
        # single GPU: the K steps as ONE sf_flow_train_epoch call (the product's single-GPU epoch loop, driven from C; with
        # SF_TRAIN_GRAPH=1 steps 2 .. K replay a captured HIP graph: step_begin -> prep -> flow -> gather -> clip + Adam ->
        # step_end, the batch's rows and Adam's step number on the device).  With N > 1 the all-reduce sits between the gather
        # and Adam and the step stays three calls (timed above).
        order = torch.cat([bidx[k % 4] for k in range(tsteps)]).contiguous()
        tl = torch.zeros((), dtype=torch.float64, device=dev)

        def epoch_call():
            flow.train_epoch(flat, Ttr, Xtr, order, tsteps, B, gscale, opt2.exp_avg, opt2.exp_avg_sq, opt2.desc, opt2.step_count, 5.0,
                             opt2.scratch, grad, tl)
            opt2.step_count += tsteps

        epoch_call()
        barrier_sync(world)
        t0 = time.perf_counter()
        epoch_call()
        barrier_sync(world)
        t_epoch = time.perf_counter() - t0
        graph_used = os.environ.get("SF_TRAIN_GRAPH", "0") == "1"   # (opt-in: measured slower than the plain launches)
        t_train = min(t_plain, t_epoch) if not graph_used else t_epoch
    # ---------------- data parallel as the product runs it: the SAME epoch call with the gradient all-reduce inside
    # (sf_flow_train_epoch_dp: prep -> flow -> gather -> ncclAllReduce on the library's stream -> clip + Adam), over an RCCL
    # communicator of the ranks of this job.  At N = 1 that is a one-rank communicator: RCCL is loaded, bootstrapped from a
    # unique id and executes one all-reduce per step -- what it adds to the step is its launch, not a transfer.
    dp = None
    if not a.skip_dp and not gloo and tsteps >= 4:
        try:
            from synference_amd.comm import RcclComm, library_info
            comm = RcclComm.from_process_group(dev) if world > 1 else RcclComm.create(dev, 1, 0)
            order_dp = torch.cat([bidx[k % 4] for k in range(tsteps)]).contiguous()
            tl_dp = torch.zeros((), dtype=torch.float64, device=dev)

            def epoch_dp():
                flow.train_epoch(flat, Ttr, Xtr, order_dp, tsteps, B, gscale, opt2.exp_avg, opt2.exp_avg_sq, opt2.desc, opt2.step_count,
                                 5.0, opt2.scratch, grad, tl_dp, comm=comm)
                opt2.step_count += tsteps

            epoch_dp()
            barrier_sync(world)
            t0 = time.perf_counter()
            epoch_dp()
            barrier_sync(world)
            t_dp = max_over_ranks(time.perf_counter() - t0, world, dev, gloo)
            for _ in range(10):
                comm.all_reduce_(grad)
            barrier_sync(world)
            t0 = time.perf_counter()
            for _ in range(200):
                comm.all_reduce_(grad)
            torch.cuda.synchronize()
            ar_us = max_over_ranks((time.perf_counter() - t0) / 200 * 1e6, world, dev, gloo)
            dp = {"call": "sf_flow_train_epoch_dp (gradient all-reduce inside the fused epoch loop)", "rccl_ranks": comm.nranks,
                  "rccl": library_info(), "ms_per_step": 1e3 * t_dp / tsteps, "value": world * tsteps * B / t_dp, "unit": "pairs/s",
                  "allreduce_us": ar_us, "allreduce_floats": int(grad.numel()), "steps": tsteps}
            if world > 1 and t_dp < t_train:
                t_train = t_dp
            comm.close()
            note(f"RCCL leg: {comm.nranks} rank(s), {dp['ms_per_step']:.4f} ms/step, bare all-reduce {ar_us:.1f} us")
        except Exception as e:   # (a box without a loadable RCCL: say so in the line instead of losing it)
            dp = {"error": f"{type(e).__name__}: {e}"}
    pairs = world * tsteps * B / t_train
    # kernel time of the forward+backward flow kernel alone (HIP events on its stream, inside the library)
    flow.set_profiling(True)
    tk = []
    for k in range(min(tsteps, 10)):
        train_step(k)
        tk.append(flow.train_kernel_ms())
    flow.set_profiling(False)
    train_kernel_ms = float(np.mean(tk))
    # throughput regime: the flow kernel alone at 8 x the batch (one library call per step, no optimiser)
    Bbig = 8 * B
    big_kernel_ms = float("nan")
    if not a.skip_throughput_regime:
        big_idx = torch.randint(0, len(tr), (Bbig,), generator=g2).to(dev)
        flow.set_profiling(True)
        tkb = []
        for k in range(6):
            flow.loss_grad_rows(flat, Ttr, Xtr, big_idx, 1.0 / Bbig, grad)
            if k >= 1:
                tkb.append(flow.train_kernel_ms())
        flow.set_profiling(False)
        big_kernel_ms = float(np.mean(tkb))
    # the bare gradient all-reduce (flat fp32 vector), so that the data-parallel step can be decomposed
    allreduce_us = None
    if world > 1:
        for _ in range(10):
            all_reduce_(grad, dist.ReduceOp.SUM, gloo)
        barrier_sync(world)
        t0 = time.perf_counter()
        for _ in range(100):
            all_reduce_(grad, dist.ReduceOp.SUM, gloo)
        torch.cuda.synchronize()
        allreduce_us = max_over_ranks((time.perf_counter() - t0) / 100 * 1e6, world, dev, gloo)
    # strong scaling: the GLOBAL batch stays at --train-batch, each rank takes 1/world of it
    Bs = max(32, B // world)
    sidx = [torch.randint(0, len(tr), (Bs,), generator=g2).to(dev) for _ in range(4)]
    gs = 1.0 / (Bs * world)

    def strong_step(k):
        flow.loss_grad_rows(flat, Ttr, Xtr, sidx[k % 4], gs, grad)
        if world > 1:
            all_reduce_(grad, dist.ReduceOp.SUM, gloo)
        opt2.step(grad, 5.0)

    for k in range(max(a.warmup, 1)):
        strong_step(k)
    barrier_sync(world)
    t0 = time.perf_counter()
    for k in range(tsteps):
        strong_step(k)
    barrier_sync(world)
    t_strong = max_over_ranks(time.perf_counter() - t0, world, dev, gloo)
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
    t64 = time.perf_counter() - t0
    pairs64 = 200 * 64 / t64
    # log_prob throughput (BASELINE.md B3 counterpart): rows/s over the test catalogue, raw density
    Tt = torch.as_tensor(th_test, dtype=torch.float32).to(dev)
    reps_lp = max(1, 200_000 // max(M, 1))
    Xl, Tl = X.repeat(reps_lp, 1), Tt.repeat(reps_lp, 1)
    flow.log_prob(Tl, Xl)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        flow.log_prob(Tl, Xl)
    torch.cuda.synchronize()
    lp_rows = 10 * Xl.shape[0] / (time.perf_counter() - t0)

    if rank != 0:
        return None
    f_train = 3.0 * wl["f_lp"]   # a training step costs 3x the log_prob figure per row
    train_tf = f_train * B / (train_kernel_ms * 1e-3) / 1e12
    ttraffic, ttraffic_src = pmc_traffic("train") if a.workload == "maf_cfg2" else (None, None)
    kname = ("k_ar_samp16<D,NI> (16-candidate register tiles, one hyper-network sweep per transform; find / resolve rounds on the same routine)"
             if wl["kind"] == "nsf_ar" and desc.get("sampler_tiles16") else
             "k_ar_sample (one wave per 64 draws, one hyper-network sweep per transform)") if wl["kind"] == "nsf_ar" else \
            (("k_maf_samp16<NB,SPAN,HM,TPW,DD,PREC=2> (16-row tiles, every product on v_mfma_f32_16x16x4_f32, first block layer folded into the input layer)" if desc.get("m16_ok") and not a.hidden_bf16 else "k_sample_persist<MafOps>")
             if wl["kind"] == "maf" else
             ("k_sample_persist<NsfOps<..., BF = 2>> (sampler image, split-bf16 hidden blocks)" if desc.get("nsf_split_sampler") and
              not a.hidden_bf16 else "k_sample_persist<NsfOps>"))
    split = default_split
    tpath = flow.train_path(B)
    tkname = ({1: "k_maf_trainc<TS,NI,NT,1> (cooperative 16-row tiles, 4 waves per 32 samples)",
               2: "k_maf_trainc<TS,NI,NT,2> (cooperative 16-row tiles, 8 waves per 64 samples)",
               3: "k_nsf_trainc<NT,OTQ> (cooperative 16-row tiles, 4 waves per 32 samples, conditioner recomputed in the backward sweep)",
               4: "sf_nsf1: context MLPs (k_mlp_*) + k_nsf1_train",
               5: "k_ar_train<NWV> (64 samples per workgroup, MFMA tiles on masked images)"}.get(tpath)
              or ("k_maf_train<HT>" if wl["kind"] == "maf" else "k_nsf_train<HT,PT>"))
    observed_world = dist.get_world_size() if (world > 1 and dist.is_initialized()) else 1
    observed_backend = dist.get_backend() if (world > 1 and dist.is_initialized()) else "none (single process, no process group)"
    rec = {
        "metric": "posterior samples/sec (accepted, prior-box rejection included)",
        "value": value, "unit": "samples/s", "n_gpus": world,
        "rccl_ranks": (dp.get("rccl_ranks", observed_world) if dp else observed_world), "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * t_samp / a.steps, "repeats": repeats, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None,
        "dtype": ("bf16 hidden-layer MFMA operands, f32 elsewhere" if a.hidden_bf16 else
                  ("f32 (sampler hidden HxH blocks: split-bf16 x3, fp32 accumulate; log_prob and training: f32 throughout)"
                   if split else "f32")),
        "data": "synthetic",
        "config": {"workload": f"{wl['label']}; sample_posterior over {M} test galaxies x {S} draws per GPU",
                   "name": a.workload,
                   "galaxies_per_gpu": M, "draws_per_galaxy": S, "theta_dim": D, "filters": C,
                   "parallelism": f"one catalogue of {world * M} rows sharded over {world} GPU(s) as contiguous row blocks, no collective",
                   "backend": observed_backend,
                   "first_attempt_acceptance": accept, "fit_steps": a.fit_steps, "fit_final_loss": fit_loss,
                   "launches_per_step": launches[0] / a.steps, "unfilled_slots": unfilled_all,
                   "flow_evaluations_per_step": evals_per_launch},
        "roofline": {"bound": "mfma", "kernel": kname + " (persistent: first attempts + retries in one launch)",
                     "achieved": useful, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                     "frac": useful / PEAK_FP32_TFLOPS, "traffic": traffic, "traffic_source": traffic_src,
                     "traffic_measured_in_this_run": False,
                     "algorithmic_bytes_per_launch": 4.0 * D * M * S + 4.0 * C * M,
                     "launch_ms": k_ms, "flops_per_launch": f_min * accepted_per_launch,
                     "accepted_draws_per_launch": accepted_per_launch,
                     "note": "achieved = USEFUL work / measured kernel time: mask-aware FLOPs of ONE conditioner evaluation per "
                             "transform (SURVEY 8d; MAF cfg1 40030 FLOP/draw -- the least any algorithm needs; the "
                             "reference's D-pass inverse spends 175150) x ACCEPTED draws; rejected evaluations, tile "
                             "padding and the per-galaxy context kernel are overhead.  issue_busy = SQ counters "
                             "of the committed PMC summary; contract_* = SURVEY "
                             "8d's figure for the reference algorithm x accepted draws (an algorithmic ratio, can exceed 1).",
                     "issue_busy": busy, ("fp32_sampler" if default_split else "split_bf16_sampler"): alt_leg,
                     "contract_tflops": contract, "contract_ratio": contract / PEAK_FP32_TFLOPS},
        "roofline_train": {"bound": "mfma", "kernel": tkname,
                           "achieved": train_tf, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                           "frac": train_tf / PEAK_FP32_TFLOPS, "traffic": ttraffic, "traffic_source": ttraffic_src,
                           "traffic_measured_in_this_run": False,
                           "launch_ms": train_kernel_ms, "rows_per_launch": B, "flops_per_launch": f_train * B,
                           "algorithmic_bytes_per_launch": 4.0 * (D + C) * B + 4.0 * 2 * flat.numel(),
                           "note": "3 x the log_prob figure per row (SURVEY 8d: forward + 2 x backward) / the flow kernel's "
                                   "duration (HIP events on its stream, sf_flow_train_stats); prep / gather / Adam launches "
                                   "are in train.ms_per_step, not here"},
        "api": api, "per_object_call": per_object,
        "train": {"metric": "flow-train theta.x pairs/sec (fwd+bwd+allreduce+clip+Adam)", "value": pairs,
                  "unit": "pairs/s", "per_gpu_batch": B, "steps": tsteps, "ms_per_step": 1e3 * t_train / tsteps,
                  "achieved_tflops": pairs * f_train / 1e12,
                  "batch64_pairs_per_s_1gpu": pairs64, "batch64_ms_per_step": 1e3 * t64 / 200,
                  "allreduce_us": allreduce_us if allreduce_us is not None else (dp or {}).get("allreduce_us"), "dp": dp,
                  "step_as_hip_graph": graph_used, "ms_per_step_python_loop": 1e3 * t_plain / tsteps,
                  "ms_per_step_epoch_call": (1e3 * t_epoch / tsteps) if (world == 1 and tsteps >= 4) else None,
                  "throughput_regime": {"per_gpu_batch": Bbig, "kernel_ms": big_kernel_ms,
                                        "achieved_tflops": f_train * Bbig / (big_kernel_ms * 1e-3) / 1e12,
                                        "frac": f_train * Bbig / (big_kernel_ms * 1e-3) / 1e12 / PEAK_FP32_TFLOPS},
                  "strong_scaling": {"global_batch": Bs * world, "per_gpu_batch": Bs, "value": pairs_strong,
                                     "ms_per_step": 1e3 * t_strong / tsteps}},
        "log_prob": {"value": lp_rows, "unit": "rows/s", "rows_per_call": int(Xl.shape[0]),
                     "achieved_tflops": lp_rows * wl["f_lp"] / 1e12},
    }
    # ---------------- BASELINE configs[4] on one GPU: 1e5 galaxies x 1000 draws through the same call (its own catalogue; the
    # few galaxies of acceptance ~1e-4 that a catalogue of this size holds put two thirds of the evaluations into the
    # find / resolve launches of the deep tail -- DESIGN.md section 3)
    if (world == 1 and a.workload == "maf_cfg2" and not a.skip_large_catalogue and not a.galaxies and not a.hidden_bf16
            and torch.cuda.get_device_properties(dev).total_memory > 16 << 30):
        nbig = 100000
        xb, _, _ = make_catalogue(nbig, C, D, seed=97531)
        Xb = torch.as_tensor(xb).to(dev)
        outb = torch.empty((nbig, S, D), dtype=torch.float32, device=dev)
        flow.sample(Xb, S, lo, hi, seed=5, out=outb)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        nrep, unfb, evb, kb = 2, 0, 0.0, []
        for k in range(nrep):
            flow.sample(Xb, S, lo, hi, seed=2000 + k, out=outb)
            unfb += flow.last_unfilled
            evb += flow.last_sample_stats["evaluations"]
            kb.append(flow.last_sample_stats["dense_ms"])
        torch.cuda.synchronize(dev)
        tb = (time.perf_counter() - t0) / nrep
        rec["large_catalogue"] = {"workload": "BASELINE configs[4] shape on one GPU: 100000 galaxies x %d draws, same flow" % S,
                                  "value": (nbig * S - unfb / nrep) / tb, "unit": "samples/s", "ms_per_step": 1e3 * tb,
                                  "persistent_launch_ms": float(np.mean(kb)), "flow_evaluations_per_step": evb / nrep,
                                  "unfilled_slots": unfb, "steps": nrep}
        del outb, Xb
        note(f"large catalogue: {1e3 * tb:.1f} ms per 1e8 draws")
    note("GPU legs done; CPU baseline (about 30 s)" if world == 1 and not a.no_cpu_baseline else "GPU legs done")
    if world == 1 and not a.no_cpu_baseline:
        rec["cpu_baseline"] = cpu_baseline(est.spec, flat.cpu().numpy(), x_test, th_test, prior.low.numpy(),
                                           prior.high.numpy(), S, a.cpu_seconds, th_lib[tr], x_lib[tr])
    return rec


if __name__ == "__main__":
    sys.exit(main())
