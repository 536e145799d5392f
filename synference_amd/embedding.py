"""Fully connected embedding net on the HIP engine.

``FCN(n_hidden, act_fn="SiLU")`` has the constructor of ltu-ili's ``FCN`` ([UPSTREAM]; the reference
mentions it at examples/sbi/scripts/train_spectral_model.py:316-317 and passes embedding nets through the
``embedding_net`` kwarg, ref: sbi_runner.py:4432, custom_runner.py:321): ``Linear(C, n_hidden[0]) -> act
-> ... -> Linear(n_hidden[-2], n_hidden[-1])``, activation after every layer except the last, input width
inferred at first use.  Forward and backward run in libsynference_hip.so (``sf_mlp_forward`` /
``sf_mlp_backward``); the module owns ONE flat fp32 parameter so torch optimisers and DDP-style
all-reduce see a single tensor.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional

import numpy as np
import torch
from torch import nn

from . import _lib

_ACT = {"silu": 0, "relu": 1, "tanh": 2}


class _FCNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flat, x, mod):
        ctx.mod = mod
        ctx.save_for_backward(flat, x)
        return mod._forward(flat, x)

    @staticmethod
    def backward(ctx, gout):
        flat, x = ctx.saved_tensors
        return ctx.mod._backward(flat, x, gout.contiguous()), None, None


class FCN(nn.Module):
    def __init__(self, n_hidden: List[int], act_fn: str = "SiLU", n_input: Optional[int] = None,
                 generator: Optional[torch.Generator] = None):
        super().__init__()
        if act_fn.lower() not in _ACT:
            raise ValueError(f"act_fn '{act_fn}' is not built on the HIP path; supported: SiLU, ReLU, Tanh")
        if not 1 <= len(n_hidden) <= 4 or max(n_hidden) > 128:
            raise ValueError("the HIP FCN supports 1..4 layers of width <= 128")
        self.n_hidden = [int(v) for v in n_hidden]
        self.act_fn = act_fn
        self.n_input = None
        self._handle = None
        self._generator = generator
        self.flat = nn.Parameter(torch.zeros(0))
        if n_input is not None:
            self.initialize(int(n_input))

    # ---- parameters ---------------------------------------------------------------------------
    def layout(self):
        out, off, n_in = [], 0, self.n_input
        for l, w in enumerate(self.n_hidden):
            out.append((f"layers.{l}.weight", (w, n_in), off)); off += w * n_in
            out.append((f"layers.{l}.bias", (w,), off)); off += w
            n_in = w
        return out, off

    def initialize(self, n_input: int):
        """torch nn.Linear default init (U(+-1/sqrt(fan_in)) for weight and bias)."""
        self.n_input = int(n_input)
        lay, n = self.layout()
        flat = torch.empty(n)
        for name, shape, off in lay:
            fan_in = shape[1] if name.endswith("weight") else dict((a, s) for a, s, _ in lay)[name[:-4] + "weight"][1]
            k = int(np.prod(shape))
            flat[off:off + k] = (torch.rand(k, generator=self._generator) * 2 - 1) / math.sqrt(fan_in)
        self.flat = nn.Parameter(flat.to(self.flat.device))
        self._handle = None

    def named_tensors(self):
        lay, _ = self.layout()
        return {n: self.flat.detach()[o:o + int(np.prod(s))].view(*s) for n, s, o in lay}

    # ---- HIP calls ----------------------------------------------------------------------------
    def _h(self):
        if self._handle is None:
            lib = _lib.load()
            d = _lib.sf_mlp_desc(n_in=self.n_input, n_layers=len(self.n_hidden),
                                 widths=(C.c_int32 * 4)(*(self.n_hidden + [0] * (4 - len(self.n_hidden)))),
                                 act=_ACT[self.act_fn.lower()], x_mean=None, x_std=None)
            h = C.c_void_p()
            _lib.check(lib.sf_mlp_create(C.byref(d), C.byref(h)))
            assert int(lib.sf_mlp_num_params(h)) == self.flat.numel()
            self._handle = h
        return self._handle

    def __getstate__(self):
        d = self.__dict__.copy()
        d["_handle"] = None
        return d

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h is not None and h.value:
            _lib.load().sf_mlp_destroy(h)

    def _check(self, x):
        if self.flat.device.type != "cuda":
            raise RuntimeError("FCN parameters live on the CPU: call .to('cuda') -- the HIP embedding has no CPU fallback")
        if x.shape[-1] != self.n_input:
            raise ValueError(f"expected {self.n_input} input features, got {x.shape[-1]}")

    def _forward(self, flat, x):
        lib, st = _lib.load(), C.c_void_p(torch.cuda.current_stream(flat.device).cuda_stream)
        x = x.contiguous()
        out = torch.empty((x.shape[0], self.n_hidden[-1]), dtype=torch.float32, device=flat.device)
        _lib.check(lib.sf_mlp_forward(self._h(), C.c_void_p(flat.data_ptr()), C.c_void_p(x.data_ptr()), x.shape[0],
                                      C.c_void_p(out.data_ptr()), st))
        return out

    def _backward(self, flat, x, gout):
        lib, st = _lib.load(), C.c_void_p(torch.cuda.current_stream(flat.device).cuda_stream)
        grad = torch.empty_like(flat)
        _lib.check(lib.sf_mlp_backward(self._h(), C.c_void_p(flat.data_ptr()), C.c_void_p(x.contiguous().data_ptr()),
                                       C.c_void_p(gout.data_ptr()), x.shape[0], C.c_void_p(grad.data_ptr()), st))
        return grad

    def forward(self, x):
        x = torch.as_tensor(x, dtype=torch.float32)
        if self.n_input is None:  # ili's FCN is lazily sized by its first input
            dev = self.flat.device
            self.initialize(x.shape[-1])
            self.to(dev)
        x = x.to(self.flat.device)
        lead = x.shape[:-1]
        x2 = x.reshape(-1, x.shape[-1])
        self._check(x2)
        if torch.is_grad_enabled() and self.flat.requires_grad:
            out = _FCNFn.apply(self.flat, x2.detach(), self)
        else:
            out = self._forward(self.flat.detach(), x2)
        return out.reshape(*lead, self.n_hidden[-1])
