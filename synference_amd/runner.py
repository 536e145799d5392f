"""Runner protocol of the reference on the HIP engine, with data-parallel training.

Protocol: ``Runner.load(backend, engine, prior, nets, train_args, out_dir, name,
device)`` then ``runner(loader[, validation_loader=]) -> (posterior, stats: List[dict])``
(ref: sbi_runner.py:4892-4936); the loader offers ``get_all_data()`` / ``get_all_parameters()``
(ref: custom_runner.py:164-165).

The epoch loop restates ``SBICustomRunner._build_and_train_model`` / ``_train_model``
(ref: src/synference/custom_runner.py:291-372, 532-742): random split, per-epoch random batches
with ``drop_last`` when the subset is larger than the batch, ``loss = mean(-log_prob)``,
``clip_grad_norm_``, Adam/AdamW, epoch loss = sum of per-sample losses / (num_batches*batch_size),
early stop on ``epochs_since_improvement >= stop_after_epochs``, best-state restore, checkpoint
every 10 epochs with auto-resume, and the ``stats`` schema of custom_runner.py:259-275, 724-730.

Data parallel (new; the reference has none): one process per GPU, every rank holds the whole
(theta, x) set in HBM, the training index set is sharded by rank, each step ends in ONE RCCL
all-reduce(SUM) of the flat fp32 gradient, followed by the identical fused clip+Adam on every rank.
"""
from __future__ import annotations

import ctypes as C
import json
import logging
import os
import time
from copy import deepcopy
from pathlib import Path
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from .estimator import FlowEstimator, load_nde_hip
from .posterior import EnsemblePosterior, FlowPosterior

logger = logging.getLogger("synference_amd")


class NumpyLoader:
    """ili ``NumpyLoader``: holds x (features) and theta (parameters)."""

    def __init__(self, x: np.ndarray, theta: np.ndarray):
        self.x = np.asarray(x)
        self.theta = np.asarray(theta)
        if len(self.x) != len(self.theta):
            raise ValueError("x and theta must have the same number of rows")

    def __len__(self):
        return len(self.x)

    def get_all_data(self):
        return self.x

    def get_all_parameters(self):
        return self.theta


class HipAdam:
    """Fused global-norm clip + Adam/AdamW on one flat tensor (sf_adam_apply)."""

    def __init__(self, param: torch.Tensor, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 decoupled=False):
        self.param = param
        self.exp_avg = torch.zeros_like(param)
        self.exp_avg_sq = torch.zeros_like(param)
        self.step_count = 0
        self.desc = _lib.sf_adam_desc(lr, betas[0], betas[1], eps, weight_decay, 1 if decoupled else 0)
        self.scratch = torch.zeros(2, dtype=torch.float32, device=param.device)
        self.lib = _lib.load()

    def step(self, grad: torch.Tensor, max_norm: Optional[float]):
        self.step_count += 1
        st = C.c_void_p(torch.cuda.current_stream(self.param.device).cuda_stream)
        _lib.check(self.lib.sf_adam_apply(
            C.c_void_p(self.param.data_ptr()), C.c_void_p(grad.data_ptr()), C.c_void_p(self.exp_avg.data_ptr()),
            C.c_void_p(self.exp_avg_sq.data_ptr()), self.param.numel(), C.byref(self.desc), self.step_count,
            C.c_float(max_norm if max_norm is not None else 0.0), C.c_void_p(self.scratch.data_ptr()), st))

    def last_grad_norm(self) -> float:
        return float(self.scratch[1].item())

    def state_dict(self):
        return {"exp_avg": self.exp_avg.detach().cpu(), "exp_avg_sq": self.exp_avg_sq.detach().cpu(),
                "step": self.step_count}

    def load_state_dict(self, sd):
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.step_count = int(sd["step"])


class TorchEmbeddingTrainOps:
    """Compute backend of ``train_flow`` when an embedding net sits in front of the flow: the flow's
    forward / backward (incl. the context gradient) still run in the HIP library through
    ``FlowEstimator``'s autograd function; the embedding module and the optimiser step are torch's."""

    def __init__(self, estimator: FlowEstimator):
        self.est = estimator
        self.params = [p for p in estimator.parameters() if p.requires_grad]

    def loss_grad(self, flat, theta, x, scale, grad_out):
        for p in self.params:
            p.grad = None
        losses = self.est.loss(theta, x)
        (losses.sum() * scale).backward()
        return losses.detach()

    def refresh(self, flat):
        self.est._packed_version = None

    def log_prob(self, theta, x):
        with torch.no_grad():
            return self.est.log_prob(theta, context=x)

    def make_optimizer(self, flat, lr, weight_decay, decoupled):
        ops = self

        class _Opt:
            def __init__(s):
                s.o = (torch.optim.AdamW if decoupled else torch.optim.Adam)(ops.params, lr=lr,
                                                                             weight_decay=weight_decay)

            def step(s, grad, max_norm):  # grads live on the parameters (autograd); `grad` is unused
                _, world = _dist_info()
                if world > 1:  # ONE collective over the concatenated gradients (latency-bound message)
                    flatg = torch.cat([p.grad.reshape(-1) for p in ops.params])
                    dist_all_reduce(flatg)
                    off = 0
                    for p in ops.params:
                        p.grad.copy_(flatg[off:off + p.numel()].view_as(p.grad))
                        off += p.numel()
                if max_norm:
                    torch.nn.utils.clip_grad_norm_(ops.params, max_norm)
                s.o.step()
                ops.est._packed_version = None

            def state_dict(s):
                return s.o.state_dict()

            def load_state_dict(s, sd):
                s.o.load_state_dict(sd)
        return _Opt()


class HipTrainOps:
    """Compute backend of ``train_flow``: the HIP library.  (The CPU test-suite injects a test double
    with the same four methods to exercise the epoch loop and the data-parallel logic under gloo.)"""

    def __init__(self, estimator: FlowEstimator):
        self.flow = estimator.flow

    def loss_grad(self, flat, theta, x, scale, grad_out):
        return self.flow.loss_grad(flat, theta, x, scale, grad_out=grad_out)[0]

    def loss_grad_rows(self, flat, theta, x, rows, scale, grad_out, loss_sum):
        """Batch = rows of the training arrays, gathered inside the kernel; the loss is summed into ``loss_sum``."""
        self.flow.loss_grad_rows(flat, theta, x, rows, scale, grad_out, loss_sum=loss_sum)

    def train_epoch(self, flat, theta, x, order, n_batches, batch, scale, opt, max_norm, grad, loss_sum, comm=None):
        """All steps of an epoch in one library call (no host round trip per step); with ``comm`` (RcclComm) every step's
        gradient is all-reduced over the ranks inside the call."""
        self.flow.train_epoch(flat, theta, x, order, n_batches, batch, scale, opt.exp_avg, opt.exp_avg_sq, opt.desc,
                              opt.step_count, max_norm if max_norm is not None else 0.0, opt.scratch, grad, loss_sum, comm=comm)
        opt.step_count += n_batches

    def refresh(self, flat):
        self.flow.set_params(flat)

    def log_prob(self, theta, x):
        return self.flow.log_prob(theta, x)

    def make_optimizer(self, flat, lr, weight_decay, decoupled):
        return HipAdam(flat, lr=lr, weight_decay=weight_decay, decoupled=decoupled)


def _get_state(estimator, embedded):
    if embedded:
        return {k: v.detach().clone() for k, v in estimator.state_dict().items()}
    return estimator.flat.data.clone()


def _set_state(estimator, state, embedded):
    if embedded:
        estimator.load_state_dict(state)
    else:
        estimator.flat.data.copy_(state)


def _dist_info():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def _host_staged(t: torch.Tensor) -> bool:
    """gloo (CPU rehearsals, several ranks on one GPU) is only relied on for host tensors: device tensors are staged
    through the host; RCCL (backend "nccl") takes them as they are."""
    return t.device.type == "cuda" and dist.get_backend() == "gloo"


def dist_all_reduce(t: torch.Tensor, op=None) -> None:
    op = dist.ReduceOp.SUM if op is None else op
    if _host_staged(t):
        h = t.detach().cpu()
        dist.all_reduce(h, op=op)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op)


def dist_broadcast(t: torch.Tensor, src: int = 0) -> None:
    if _host_staged(t):
        h = t.detach().cpu()
        dist.broadcast(h, src=src)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src)


def split_indices(n: int, validation_fraction: float, generator: torch.Generator):
    """custom_runner.py:337-343: ``num_val = int(f*n)``; random_split of range(n)."""
    num_val = int(validation_fraction * n)
    perm = torch.randperm(n, generator=generator)
    return perm[: n - num_val], perm[n - num_val:]


def train_flow(estimator: FlowEstimator, theta: torch.Tensor, x: torch.Tensor, *, batch_size: int = 32,
               learning_rate: float = 1e-4, validation_fraction: float = 0.1, stop_after_epochs: int = 20,
               clip_max_norm: Optional[float] = 5.0, optimizer_choice: str = "Adam", max_num_epochs: int = 2 ** 31 - 1,
               save_dir: Optional[str] = None, seed: Optional[int] = None, val_theta=None, val_x=None,
               log_every: int = 1, on_epoch: Optional[Callable] = None, ops=None) -> Dict:
    """Trains in place; returns the summary dict (custom_runner.py:724-730 + timing)."""
    rank, world = _dist_info()
    dev = theta.device
    flat = estimator.flat
    embedded = bool(getattr(estimator, "has_embedding", False))
    if ops is None:
        ops = TorchEmbeddingTrainOps(estimator) if embedded else HipTrainOps(estimator)
    if seed is None:
        seed = int(time.time())
    if world > 1:
        # every rank must draw the SAME train/validation split and the same epoch orders: rank 0's seed wins
        # (a per-rank time() seed would make shards overlap and leak validation rows into training)
        sd = torch.tensor([int(seed)], dtype=torch.int64, device=dev)
        dist_broadcast(sd, src=0)
        seed = int(sd.item())
    gen = torch.Generator().manual_seed(int(seed))
    N = theta.shape[0]
    if val_theta is None:
        tr_idx, va_idx = split_indices(N, validation_fraction, gen)
        tr_idx, va_idx = tr_idx.to(dev), va_idx.to(dev)
        val_theta_t, val_x_t = theta[va_idx], x[va_idx]
    else:
        tr_idx = torch.arange(N, device=dev)
        val_theta_t, val_x_t = val_theta, val_x
    if world > 1:  # identical start (flow, embedding net, buffers) + disjoint shards of equal size
        seen = set()
        for t in list(estimator.parameters()) + list(estimator.buffers()):
            if t.data_ptr() in seen:
                continue
            seen.add(t.data_ptr())
            dist_broadcast(t.data, src=0)
        per = tr_idx.numel() // world
        tr_idx = tr_idx[rank * per:(rank + 1) * per]
        vper = val_theta_t.shape[0] // world
        if vper > 0:
            val_theta_t, val_x_t = val_theta_t[rank * vper:(rank + 1) * vper], val_x_t[rank * vper:(rank + 1) * vper]
    n_tr, n_va = tr_idx.numel(), val_theta_t.shape[0]
    if n_tr == 0:
        raise ValueError("no training rows on this rank")
    bs_tr = min(batch_size, n_tr)
    drop_tr = n_tr > batch_size
    nb_tr = n_tr // bs_tr if drop_tr else 1
    bs_va = min(batch_size, max(n_va, 1))
    nb_va = (n_va // bs_va) if n_va > batch_size else (1 if n_va > 0 else 0)
    opt = ops.make_optimizer(flat.data, learning_rate, (0.01 if optimizer_choice == "AdamW" else 0.0),
                             optimizer_choice == "AdamW")
    grad = torch.empty_like(flat.data)
    gscale = 1.0 / (bs_tr * world)
    # flow-only parameters, library optimiser: run each epoch's batch loop inside the library -- on one device as it is, under
    # an RCCL process group with the per-step gradient all-reduce inside the call (sf_flow_train_epoch_dp: prep -> flow ->
    # gather -> ncclAllReduce -> clip + Adam on one stream, no host round trip per step).  Other groups (gloo rehearsals, several
    # ranks on one device) keep the per-step path: loss_grad_rows -> c10d all_reduce -> optimiser step.
    plain_f32 = (theta.dtype == torch.float32 and x.dtype == torch.float32 and theta.is_contiguous() and x.is_contiguous())
    can_fuse = not embedded and hasattr(ops, "train_epoch") and isinstance(opt, HipAdam) and plain_f32
    comm = None
    if world > 1 and can_fuse and isinstance(ops, HipTrainOps):
        from .comm import default_comm
        comm = default_comm(dev)
    fused_epoch = can_fuse and (world == 1 or comm is not None)
    fused_rows = not fused_epoch and not embedded and hasattr(ops, "loss_grad_rows") and plain_f32

    if isinstance(ops, HipTrainOps) and rank == 0:
        try:    # shape cliffs are not silent: the generic kernels run at well under half the cooperative kernels' rate
            if ops.flow.train_path(bs_tr) == 0:
                sp = estimator.spec
                logger.warning(f"{sp.kind} D={sp.D} C={sp.C} H={sp.H} T={sp.T} K={sp.K}: this shape trains on the generic kernels "
                               "(k_maf_train / k_nsf_train, sf_flow_train_path = 0), not on the cooperative 16-row kernels "
                               "(MAF: two blocks, D <= 8, H <= 64, T <= 8; NSF: two blocks, D 2..8, H <= 80, 3K - 1 <= 32, C <= 40)")
        except Exception:
            pass
    best_val, since, best_state = float("inf"), 0, None
    train_log, val_log, epoch = [], [], 0
    ckpt = f"{save_dir}checkpoint_posterior.pt" if save_dir else None
    resume = bool(ckpt and rank == 0 and os.path.exists(ckpt))
    ck = torch.load(ckpt, map_location="cpu") if resume else None
    if world > 1:
        # rank 0 decides (storage need not be shared) and hands the checkpoint to the others, so that every rank
        # resumes at the same epoch and the collectives stay in step
        box = [ck]
        dist.broadcast_object_list(box, src=0)
        ck = box[0]
    if ck is not None:  # custom_runner.py:559-573
        if embedded:
            estimator.load_state_dict(ck["model_state_dict"])
        else:
            flat.data.copy_(ck["model_state_dict"]["flat"])
        opt.load_state_dict(ck["optimizer_state_dict"])
        epoch, train_log, val_log = ck.get("epoch", 0), ck.get("train_loss", []), ck.get("val_loss", [])
        since, best_val = ck.get("epochs_since_improvement", 0), ck.get("best_val_loss", float("inf"))
        best_state = ck.get("best_model_state_dict", None)
        if best_state is not None:
            best_state = ({k: v.to(dev) for k, v in best_state.items()} if embedded else best_state["flat"].to(dev))
        logger.info(f"Resumed from epoch {epoch} with best validation loss {best_val:.4f}")

    t0 = time.time()
    rows_seen = 0
    acc = torch.zeros(2, dtype=torch.float64, device=dev)
    while epoch <= max_num_epochs and since < stop_after_epochs:
        # ---- train: fresh random order of this rank's shard (SubsetRandomSampler)
        order = tr_idx[torch.randperm(n_tr, generator=gen).to(dev)]
        tl = torch.zeros((), dtype=torch.float64, device=dev)
        if fused_epoch:
            # the whole batch loop in one library call: row gather fused into the kernel, loss summed on the device
            if comm is not None:
                ops.train_epoch(flat.data, theta, x, order.contiguous(), nb_tr, bs_tr, gscale, opt, clip_max_norm, grad, tl, comm=comm)
            else:
                ops.train_epoch(flat.data, theta, x, order.contiguous(), nb_tr, bs_tr, gscale, opt, clip_max_norm, grad, tl)
        elif fused_rows:
            # data parallel: fused-gather loss_grad, ONE all-reduce of the flat gradient, identical step on every rank
            order = order.contiguous()
            for b in range(nb_tr):
                ops.loss_grad_rows(flat.data, theta, x, order[b * bs_tr:(b + 1) * bs_tr], gscale, grad, tl)
                if world > 1:
                    dist_all_reduce(grad)
                opt.step(grad, clip_max_norm)
        else:
            for b in range(nb_tr):
                idx = order[b * bs_tr:(b + 1) * bs_tr]
                loss = ops.loss_grad(flat.data, theta[idx], x[idx], gscale, grad)
                if world > 1 and not embedded:
                    dist_all_reduce(grad)
                opt.step(grad, clip_max_norm)
                tl += loss.double().sum()
        rows_seen += nb_tr * bs_tr * world
        epoch += 1
        # ---- validate (no_grad): raw per-sample losses under the updated parameters
        ops.refresh(flat.data)
        vl = torch.zeros((), dtype=torch.float64, device=dev)
        if nb_va > 0:
            vorder = torch.randperm(n_va, generator=gen).to(dev)[: nb_va * bs_va]
            vl = -ops.log_prob(val_theta_t[vorder], val_x_t[vorder]).double().sum()
        acc[0], acc[1] = tl, vl
        if world > 1:
            dist_all_reduce(acc)
        tsum, vsum = acc.tolist()
        train_avg = tsum / (nb_tr * bs_tr * world)
        val_avg = vsum / (nb_va * bs_va * world) if nb_va > 0 else float("nan")
        train_log.append(train_avg)
        val_log.append(val_avg)
        if val_avg < best_val:  # custom_runner.py:655-660
            best_val, since = val_avg, 0
            best_state = _get_state(estimator, embedded)
        else:
            since += 1
        elapsed = time.time() - t0
        if rank == 0 and log_every and epoch % log_every == 0:
            logger.info(f"Epoch {epoch}: TL: {train_avg:.3f}, VL: {val_avg:.3f}, Best VL: {best_val:.3f}, "
                        f"ESI: {since}/{stop_after_epochs}, Avg. Time/epoch: {elapsed / epoch:.2f}s")
        if on_epoch is not None:
            on_epoch(epoch, train_avg, val_avg)
        if rank == 0 and ckpt and epoch % 10 == 0:  # custom_runner.py:690-706
            os.makedirs(os.path.dirname(ckpt) or ".", exist_ok=True)
            torch.save({"epoch": epoch,
                        "model_state_dict": ({k: v.cpu() for k, v in estimator.state_dict().items()} if embedded
                                             else {"flat": flat.data.cpu()}),
                        "optimizer_state_dict": opt.state_dict(), "train_loss": train_log, "val_loss": val_log,
                        "epochs_since_improvement": since, "best_val_loss": best_val,
                        "best_model_state_dict": None if best_state is None else (
                            {k: v.cpu() for k, v in best_state.items()} if embedded else {"flat": best_state.cpu()}),
                        "time_elapsed": elapsed}, ckpt)
    if best_state is not None:
        _set_state(estimator, best_state, embedded)
    ops.refresh(flat.data)
    estimator._packed_version = (flat.data_ptr(), flat._version)
    estimator.zero_grad(set_to_none=True)
    if rank == 0 and ckpt and os.path.exists(ckpt):
        os.remove(ckpt)
    elapsed = time.time() - t0
    return {"training_loss": train_log, "validation_loss": val_log, "best_validation_loss": [best_val],
            "epochs_trained": [epoch], "converged": since >= stop_after_epochs,
            "training_time_sec": elapsed, "pairs_per_sec": rows_seen / max(elapsed, 1e-9)}


def finish_summary(s: Dict) -> Dict:
    """Sign-flipped mirrors exactly as custom_runner.py:259-275."""
    s["training_log_probs"] = [-1.0 * v for v in s["training_loss"]]
    s["validation_log_probs"] = [-1.0 * v for v in s["validation_loss"]]
    s["best_validation_log_prob"] = [-1.0 * v for v in s["best_validation_loss"]]
    return s


class HIPRunner:
    """Drop-in for ili ``InferenceRunner`` / ``SBICustomRunner`` on the NPE + MAF/NSF path."""

    def __init__(self, prior, engine: str = "NPE", nets: Optional[List[Callable]] = None,
                 net_configs: Optional[List[Dict]] = None, embedding_net=None, train_args: Optional[Dict] = None,
                 out_dir=None, device: str = "cuda", proposal=None, name: str = "", signatures=None):
        if "NPE" not in engine.upper():
            raise ValueError(f"engine '{engine}' is not on the HIP path: only (S)NPE is built")
        self.prior, self.engine = prior, engine
        self.nets = list(nets) if nets else []
        self.net_configs = net_configs or []
        self.embedding_net = embedding_net
        self.train_args = dict(train_args or {})
        self.out_dir = Path(out_dir) if out_dir is not None else None
        self.device = "cuda" if str(device).startswith("cuda") else str(device)
        self.name = name or ""
        self.signatures = signatures

    @classmethod
    def load(cls, backend=None, engine="NPE", prior=None, nets=None, train_args=None, out_dir=None, device="cuda",
             name="", signatures=None, **kw):
        """Signature of ili ``InferenceRunner.load`` as called at ref: sbi_runner.py:4892-4901."""
        if backend not in (None, "hip"):
            raise ValueError(f"backend '{backend}' is not served by HIPRunner")
        return cls(prior=prior, engine=engine, nets=nets, train_args=train_args, out_dir=out_dir, device=device,
                   name=name, signatures=signatures, **kw)

    def _device(self):
        if not torch.cuda.is_available():
            raise RuntimeError("HIPRunner needs a GPU: the HIP flow engine has no CPU fallback")
        return torch.device("cuda", torch.cuda.current_device())

    def __call__(self, loader, validation_loader=None, seed: Optional[int] = None):
        dev = self._device()
        x = torch.from_numpy(np.ascontiguousarray(loader.get_all_data())).float().to(dev)
        theta = torch.from_numpy(np.ascontiguousarray(loader.get_all_parameters())).float().to(dev)
        ta = self.train_args
        nets = list(self.nets)
        for cfg in self.net_configs:  # custom-runner style configs: {"model": "nsf", "hidden_features": ..}
            cfg = {k: v for k, v in cfg.items() if k not in ("signature", "repeats")}
            nets.append(load_nde_hip(self.engine, embedding_net=self.embedding_net, **cfg))
        if not nets:
            raise ValueError("no density-estimator factories were given")
        gen = torch.Generator().manual_seed(seed if seed is not None else 0)
        posteriors, stats = [], []
        val_theta = val_x = None
        if validation_loader is not None and ta.get("use_validation_loader", False):
            val_x = torch.from_numpy(np.ascontiguousarray(validation_loader.get_all_data())).float().to(dev)
            val_theta = torch.from_numpy(np.ascontiguousarray(validation_loader.get_all_parameters())).float().to(dev)
        for i, build_fn in enumerate(nets):
            est = build_fn(batch_theta=theta, batch_x=x, device=dev, generator=gen).to(dev)
            summary = train_flow(
                est, theta, x,
                batch_size=ta.get("training_batch_size", 32), learning_rate=ta.get("learning_rate", 1e-4),
                validation_fraction=ta.get("validation_fraction", 0.1),
                stop_after_epochs=ta.get("stop_after_epochs", 20), clip_max_norm=ta.get("clip_max_norm", 5.0),
                optimizer_choice=ta.get("optimizer_choice", "Adam"),
                max_num_epochs=ta.get("max_num_epochs", 2 ** 31 - 1),
                save_dir=(f"{self.out_dir}/{self.name}" if self.out_dir else None),
                seed=None if seed is None else seed + i, val_theta=val_theta, val_x=val_x,
                log_every=ta.get("log_every", 1))
            stats.append(finish_summary(summary))
            posteriors.append(FlowPosterior(est, self.prior, seed=(seed or 0) + i))
        v = torch.tensor([s["best_validation_log_prob"][0] for s in stats], dtype=torch.float64)
        weights = torch.exp(v - v.max())  # ili: w_i proportional to exp(v_i - max v)
        posterior = EnsemblePosterior(posteriors, weights=(weights / weights.sum()).float(), seed=seed or 0)
        posterior.name = self.name
        posterior.signatures = self.signatures if self.signatures is not None else [""] * len(posteriors)
        rank, _ = _dist_info()
        if self.out_dir and rank == 0:
            self._save_models(posterior, stats)
        return posterior, stats

    def _save_models(self, posterior, stats):
        """ili file names: {out_dir}/{name}posterior.pkl and {name}summary.json."""
        import pickle
        self.out_dir.mkdir(parents=True, exist_ok=True)
        # FlowEstimator.__getstate__ drops the ctypes handle; it is rebuilt lazily after unpickling
        with open(self.out_dir / f"{self.name}posterior.pkl", "wb") as fh:
            pickle.dump(posterior, fh)
        with open(self.out_dir / f"{self.name}summary.json", "w") as fh:
            json.dump([{k: v for k, v in s.items()} for s in stats], fh)
