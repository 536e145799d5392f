"""Net-factory protocol of the reference on the HIP engine.

The reference builds its density estimator with
    ``net = ili.utils.load_nde_sbi(engine, model=..., embedding_net=..., hidden_features=...,
                                  num_transforms=..., **extra)``      (ref: sbi_runner.py:5123-5146)
    ``estimator = net(batch_x=x_train, batch_theta=theta_train)``     (ref: custom_runner.py:320-326)
and then only uses ``estimator.log_prob(theta, context=x)`` / ``estimator.loss(theta, x)``,
``.parameters()``, ``.state_dict()/.load_state_dict()``, ``.train()/.eval()/.to()/.zero_grad()``
(custom_runner.py:563, 596-604, 658, 709-712).  ``load_nde_hip`` / ``FlowEstimator`` offer exactly
that surface; all arithmetic happens in libsynference_hip.so.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Union

import numpy as np
import torch
from torch import nn

from .engine import HipFlow
from .spec import FlowSpec, init_params, num_params, random_perms, state_dict_views, zscore_stats

SUPPORTED_MODELS = ("maf", "nsf")


class _NegLogProb(torch.autograd.Function):
    """loss_b = -log p(theta_b | x_b); backward is the weighted HIP backward sweep
    (vector-Jacobian product with the incoming per-sample gradient)."""

    @staticmethod
    def forward(ctx, flat, theta, x, est):
        est._sync_params()
        ctx.est = est
        ctx.save_for_backward(flat, theta, x)
        return -est.flow.log_prob(theta, x)

    @staticmethod
    def backward(ctx, gout):
        flat, theta, x = ctx.saved_tensors
        dctx = torch.empty_like(x) if ctx.needs_input_grad[2] else None
        _, grad = ctx.est.flow.loss_grad(flat.detach(), theta, x, 1.0, weights=gout.contiguous(), dctx_out=dctx)
        ctx.est._packed_version = None  # loss_grad re-tiled the image from `flat`; re-check next call
        return (grad if ctx.needs_input_grad[0] else None), None, dctx, None


class FlowEstimator(nn.Module):
    """Conditional flow q(theta | x) with ONE flat fp32 parameter (logical layout of the C ABI)."""

    def __init__(self, spec: FlowSpec, flat: Optional[torch.Tensor] = None, device="cuda:0",
                 generator: Optional[torch.Generator] = None, embedding_net: Optional[nn.Module] = None,
                 x_mean=None, x_std=None):
        """``embedding_net`` (optional, any nn.Module): the flow then sees
        ``embedding_net((x - x_mean) / x_std)`` as its context (sbi: Sequential(Standardize, embedding),
        SURVEY.md B.2) and ``spec.C`` is the EMBEDDED width with a no-op in-flow standardisation."""
        super().__init__()
        self.spec = spec
        if flat is None:
            flat = init_params(spec, generator)
        if flat.numel() != num_params(spec):
            raise ValueError("flat parameter vector has the wrong length")
        self.flat = nn.Parameter(flat.detach().clone().float())
        self._flow: Optional[HipFlow] = None
        self._device = torch.device(device)
        self._packed_version = None
        self.embedding_net = embedding_net if embedding_net is not None else nn.Identity()
        self.has_embedding = not isinstance(self.embedding_net, nn.Identity)
        if self.has_embedding:
            self.register_buffer("x_mean_raw", torch.as_tensor(np.asarray(x_mean), dtype=torch.float32))
            self.register_buffer("x_std_raw", torch.as_tensor(np.asarray(x_std), dtype=torch.float32))

    def embed(self, x):
        """Context handed to the flow for raw features x (identity embedding: x itself)."""
        x = torch.as_tensor(x, dtype=torch.float32, device=self.flat.device)
        if not self.has_embedding:
            return x
        return self.embedding_net((x - self.x_mean_raw) / self.x_std_raw)

    # ---- plumbing ---------------------------------------------------------------------------
    @property
    def flow(self) -> HipFlow:
        if self._flow is None:
            self._flow = HipFlow(self.spec, self._device)
            self._packed_version = None  # a fresh handle has no operand image yet
        return self._flow

    def to(self, device=None, *a, **k):  # keep the handle's device in step with the parameter's
        out = super().to(device, *a, **k)
        if device is not None and torch.device(device).type == "cuda":
            dev = torch.device(device)
            dev = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device()
                               if torch.cuda.is_available() else 0)
            if dev != self._device:
                self._device, self._flow, self._packed_version = dev, None, None
        return out

    def _sync_params(self):
        """Re-tile the operand image when the parameter tensor changed (torch version counter)."""
        key = (self.flat.data_ptr(), self.flat._version)
        if self._flow is None or self._packed_version != key:
            if self.flat.device.type != "cuda":
                raise RuntimeError("FlowEstimator parameters live on the CPU: call .to('cuda') -- the HIP "
                                   "flow engine has no CPU fallback")
            self.flow.set_params(self.flat.detach())
            self._packed_version = key

    def __getstate__(self):
        d = self.__dict__.copy()
        d["_flow"] = None
        d["_packed_version"] = None
        return d

    def __setstate__(self, d):
        super().__setstate__(d)
        self._flow = None
        self._packed_version = None

    # ---- estimator surface ------------------------------------------------------------------
    def log_prob(self, inputs, context=None, condition=None):
        x = context if context is not None else condition
        theta = torch.as_tensor(inputs, dtype=torch.float32, device=self.flat.device)
        e = self.embed(x)
        if torch.is_grad_enabled() and (self.flat.requires_grad or e.requires_grad):
            return -_NegLogProb.apply(self.flat, theta, e.contiguous(), self)
        self._sync_params()
        return self.flow.log_prob(theta, e)

    def loss(self, theta, x):
        """sbi >= 0.23 estimator API: per-sample negative log-likelihood (custom_runner.py:596-601)."""
        return -self.log_prob(theta, context=x)

    def sample(self, num_samples: int, context, seed: int = 0):
        """nflows ``Flow.sample(n, context)`` -> (M, n, D), unconstrained draws."""
        self._sync_params()
        with torch.no_grad():
            e = self.embed(context)
        return self.flow.sample(e, int(num_samples), seed=seed)

    def sample_and_log_prob(self, num_samples, context, seed: int = 0):
        s = self.sample(num_samples, context, seed)
        M = s.shape[0]
        ctx = torch.as_tensor(context, dtype=torch.float32, device=s.device).repeat_interleave(num_samples, 0)
        return s, self.log_prob(s.reshape(M * num_samples, -1), ctx).reshape(M, num_samples)

    def named_tensors(self):
        return state_dict_views(self.spec, self.flat.detach())


def build_flow(model: str, batch_theta, batch_x, hidden_features: int = 50, num_transforms: int = 5,
               num_bins: Optional[int] = None, num_blocks: int = 2, z_score_theta="independent", z_score_x="independent",
               embedding_net: Optional[nn.Module] = None, device="cuda:0",
               generator: Optional[torch.Generator] = None, backend: str = "sbi", **extra) -> FlowEstimator:
    """sbi ``build_maf`` / ``build_nsf`` ([UPSTREAM]) on the HIP engine; with ``backend="lampe"``
    the flow ``ili.utils.load_nde_lampe(model="nsf")`` builds: ``zuko.flows.NSF(D, C, transforms=num_transforms,
    hidden_features=[hidden_features] * 2)`` -- 8 bins, bound 5, autoregressive -- behind standardising affines; ``model="maf"`` there
    is ``zuko.flows.MAF``: the same hyper-network with zuko's MonotonicAffineTransform as the univariate map (kind ``maf_ar``)."""
    if model not in SUPPORTED_MODELS:
        raise ValueError(f"model '{model}' is not on the HIP path; supported: {SUPPORTED_MODELS}")
    fixed = {}
    if backend == "lampe":
        if model not in ("nsf", "maf"):
            raise ValueError(f"backend 'lampe': models 'nsf' (zuko.flows.NSF) and 'maf' (zuko.flows.MAF) are on the HIP path, not '{model}'")
        model = "nsf_ar" if model == "nsf" else "maf_ar"
        num_bins = 8 if num_bins is None else num_bins
        num_blocks = 2
        fixed = dict(tail_bound=5.0)
    elif backend != "sbi":
        raise ValueError(f"backend '{backend}' is not on the HIP path: 'sbi' or 'lampe'")
    num_bins = 10 if num_bins is None else num_bins
    theta = torch.as_tensor(np.asarray(batch_theta.detach().cpu() if torch.is_tensor(batch_theta) else batch_theta),
                            dtype=torch.float32)
    x = torch.as_tensor(np.asarray(batch_x.detach().cpu() if torch.is_tensor(batch_x) else batch_x),
                        dtype=torch.float32)
    D, C = theta.shape[1], x.shape[1]
    st = zscore_stats(theta, x)
    if z_score_theta in (None, "none", False):
        st["theta_mean"], st["theta_std"] = np.zeros(D, np.float32), np.ones(D, np.float32)
    if z_score_x in (None, "none", False):
        st["x_mean"], st["x_std"] = np.zeros(C, np.float32), np.ones(C, np.float32)
    perms = random_perms(D, num_transforms, generator) if model == "maf" else None
    if embedding_net is not None and not isinstance(embedding_net, nn.Identity):
        # sbi: C_e = embedding_net(standardised x[:1]).numel(); the flow does not standardise again
        from .embedding import FCN
        if isinstance(embedding_net, FCN):
            if embedding_net.n_input is None:
                embedding_net.initialize(C)
            Ce = embedding_net.n_hidden[-1]
        else:
            with torch.no_grad():
                pdev = next(iter(embedding_net.parameters()), torch.zeros(1)).device
                probe = embedding_net(((x[:2] - torch.as_tensor(st["x_mean"])) / torch.as_tensor(st["x_std"])).to(pdev))
            Ce = int(probe[0].numel())
        spec = FlowSpec(kind=model, D=D, C=Ce, H=int(hidden_features), T=int(num_transforms), K=int(num_bins),
                        NB=int(num_blocks), perms=perms, theta_mean=st["theta_mean"], theta_std=st["theta_std"], **fixed)
        return FlowEstimator(spec, device=device, generator=generator, embedding_net=embedding_net,
                             x_mean=st["x_mean"], x_std=st["x_std"])
    spec = FlowSpec(kind=model, D=D, C=C, H=int(hidden_features), T=int(num_transforms), K=int(num_bins),
                    NB=int(num_blocks), perms=perms, **st, **fixed)
    est = FlowEstimator(spec, device=device, generator=generator)
    return est


def load_nde_hip(engine: str = "NPE", model: str = "maf", embedding_net: Optional[nn.Module] = None,
                 repeats: int = 1, backend: str = "sbi", **model_args) -> Union[Callable, List[Callable]]:
    """Counterpart of ``ili.utils.load_nde_sbi`` -- and, with ``backend="lampe"``, of ``ili.utils.load_nde_lampe`` --
    (called at ref: sbi_runner.py:5121-5146 and custom_runner.py:320-324): returns ``build_fn(batch_theta=, batch_x=)``
    or a list of them.  ``device=`` (which the reference passes to the lampe loader) is accepted and ignored: the flow
    lives on the GPU the builder is called for."""
    if "NPE" not in engine.upper():
        raise ValueError(f"engine '{engine}' is not on the HIP path: only (S)NPE is built")
    if model not in SUPPORTED_MODELS:
        raise ValueError(f"model '{model}' is not on the HIP path; supported: {SUPPORTED_MODELS}")
    if backend not in ("sbi", "lampe"):
        raise ValueError(f"backend '{backend}' is not on the HIP path: 'sbi' or 'lampe'")
    if backend == "lampe" and model not in ("nsf", "maf"):
        raise ValueError(f"backend 'lampe': models 'nsf' (zuko.flows.NSF) and 'maf' (zuko.flows.MAF) are on the HIP path, not '{model}'")
    model_args.pop("device", None)

    def build_fn(batch_theta=None, batch_x=None, device="cuda:0", generator=None):
        return build_flow(model, batch_theta, batch_x, embedding_net=embedding_net, device=device,
                          generator=generator, backend=backend, **model_args)

    build_fn.model = model
    build_fn.backend = backend
    build_fn.model_args = dict(model_args)
    return build_fn if repeats == 1 else [build_fn for _ in range(repeats)]
