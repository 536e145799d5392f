"""Import an upstream (sbi / pyknos-nflows) MAF or NSF into the HIP engine's flat parameter vector.

SURVEY.md 8f row f4.  The reference stores trained posteriors as sbi pickles
(ref: src/synference/sbi_runner.py:7436-7485, ``load_model_from_pkl``); opening one needs sbi, which this
build never imports.  What crosses the boundary instead is the estimator's ``state_dict()`` -- a plain
``{name: array}`` mapping that ``scripts/export_upstream_golden.py`` (run off-box, where sbi is installed)
writes into an ``.npz`` -- and this module maps it, by module path, onto ``FlowSpec`` + the flat vector
documented in include/synference_hip.h.  No dependency beyond numpy / torch.

Module paths follow pyknos-nflows 0.14 / sbi 0.22-0.23:

    [net.]_transform._transforms.0._shift / ._scale                      z-score of theta (AffineTransform)
    [net.]_embedding_net.0._mean / ._std                                 z-score of x (Standardize)
  MAF, block i = 2t (transform) and 2t+1 (permutation) of  _transform._transforms.1._transforms:
    .{2t}.autoregressive_net.initial_layer.{weight,bias}                 W0, b0          (+ .mask buffers, see below)
    .{2t}.autoregressive_net.context_layer.{weight,bias}                 Wc, bc
    .{2t}.autoregressive_net.blocks.{k}.linear.{weight,bias}             W{k+1}, b{k+1}
    .{2t}.autoregressive_net.final_layer.{weight,bias}                   Wf, bf
    .{2t+1}._permutation                                                 perms[t]
  NSF, block i = 2t (coupling) and 2t+1 (LULinear; absent when D == 1):
    .{2t}.transform_net.initial_layer.{weight,bias}                      Win [H, d_id + C], bin
    .{2t}.transform_net.blocks.{k}.context_layer.{weight,bias}           Wg, bg
    .{2t}.transform_net.blocks.{k}.linear_layers.{0,1}.{weight,bias}     W1, b1, W2, b2
    .{2t}.transform_net.final_layer.{weight,bias}                        Wout, bout
    .{2t+1}.{lower_entries,upper_entries,unconstrained_upper_diag,bias}  lu.*

nflows keeps its MADE masks (``*.mask``) and the coupling split (``transform_features`` / ``identity_features``) as
BUFFERS, so a real checkpoint carries upstream's own connectivity.  The importer compares them with the connectivity
this engine builds from (D, H) alone -- degrees ``j % max(1, D-1) + min(1, D-1)``, ``>=`` in hidden layers, strict at
the output, even dimensions transformed by even coupling blocks -- and raises ``ValueError`` on the first difference:
a checkpoint cannot be imported into a flow that would wire it differently.

Names are matched by SUFFIX after the transform index, so wrapper prefixes of other sbi versions
(``net.``, ``_neural_net.``, ``posterior_estimator.``) do not matter.  Anything that cannot be mapped raises
``KeyError`` naming what is missing -- nothing is guessed.
"""
from __future__ import annotations

import re
from typing import Dict, Mapping, Optional, Tuple

import numpy as np

from .spec import FlowSpec, num_params, param_layout


def _np(v) -> np.ndarray:
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.asarray(v)


def _strip(sd: Mapping[str, object]) -> Dict[str, np.ndarray]:
    """name -> array with any wrapper prefix in front of ``_transform`` / ``_embedding_net`` removed."""
    out = {}
    for k, v in sd.items():
        m = re.search(r"(_transform\.|_embedding_net\.|_distribution\.)", k)
        out[k[m.start():] if m else k] = _np(v)
    return out


def _blocks(sd: Dict[str, np.ndarray]) -> Dict[int, Dict[str, np.ndarray]]:
    """{i: {suffix: array}} for the entries of the inner CompositeTransform (``_transform._transforms.1._transforms.i.*``;
    without a theta z-score the chain sits directly under ``_transform._transforms.i.*``)."""
    pat_inner = re.compile(r"^_transform\._transforms\.1\._transforms\.(\d+)\.(.+)$")
    pat_flat = re.compile(r"^_transform\._transforms\.(\d+)\.(.+)$")
    inner = any(pat_inner.match(k) for k in sd)
    out: Dict[int, Dict[str, np.ndarray]] = {}
    for k, v in sd.items():
        m = (pat_inner if inner else pat_flat).match(k)
        if m and not (not inner and m.group(2) in ("_shift", "_scale")):
            out.setdefault(int(m.group(1)), {})[m.group(2)] = v
    return out


def made_masks(D: int, H: int):
    """(M0 [H, D], Mh [H, H], Mf [2D, H]) boolean MADE masks of this engine (sf_layout.cpp; [UPSTREAM] nflows
    MaskedLinear._get_mask_and_degrees with sequential degrees and output multiplier 2)."""
    mx, mn = max(1, D - 1), min(1, D - 1)
    deg_in = np.arange(1, D + 1)
    deg_h = np.arange(H) % mx + mn
    M0 = deg_h[:, None] >= deg_in[None, :]
    Mh = deg_h[:, None] >= deg_h[None, :]
    deg_out = np.repeat(deg_in, 2)                 # rows (a_0, m_0, a_1, m_1, ...)
    Mf = deg_out[:, None] > deg_h[None, :]
    return M0, Mh, Mf


def _check_connectivity(kind: str, D: int, H: int, NB: int, blocks, tidx) -> int:
    """Compare upstream's connectivity buffers, where the state dict has them, with this engine's; returns how many
    buffers were checked."""
    n = 0
    if kind == "maf":
        M0, Mh, Mf = made_masks(D, H)
        want = {"autoregressive_net.initial_layer.mask": M0, "autoregressive_net.final_layer.mask": Mf}
        for k in range(NB):
            want[f"autoregressive_net.blocks.{k}.linear.mask"] = Mh
        for t, i in enumerate(tidx):
            for key, ours in want.items():
                got = blocks[i].get(key)
                if got is None:
                    continue
                n += 1
                if got.shape != ours.shape or not np.array_equal(got != 0, ours):
                    raise ValueError(f"transform {t}: upstream '{key}' differs from this engine's MADE mask "
                                     f"({int(((got != 0) != ours).sum()) if got.shape == ours.shape else 'shape'} entries): "
                                     "the checkpoint was built with a different degree assignment")
    else:
        for t, i in enumerate(tidx):
            ours_tr = np.arange(D)[np.arange(D) % 2 == t % 2]     # even dims in even blocks (sbi: mask = alternating +-1)
            got = blocks[i].get("transform_features")
            if got is not None:
                n += 1
                if not np.array_equal(np.sort(got.reshape(-1)), ours_tr):
                    raise ValueError(f"coupling block {t}: upstream transforms dimensions {got.reshape(-1).tolist()}, this engine "
                                     f"{ours_tr.tolist()}")
    return n


def spec_and_flat_from_state_dict(state_dict: Mapping[str, object], num_bins: Optional[int] = None,
                                  **spec_overrides) -> Tuple[FlowSpec, np.ndarray]:
    """(FlowSpec, flat float32 vector) of an upstream MAF / NSF ``state_dict``.

    ``num_bins`` is needed for NSF only when it cannot be inferred (it is: final_layer rows = d_tr * (3K - 1)).
    ``spec_overrides``: constants that the state dict does not carry (``tail_bound``, ``scale_fn`` ...)."""
    sd = _strip(state_dict)
    blocks = _blocks(sd)
    if not blocks:
        raise KeyError("no '_transform._transforms.*' entries: this is not an nflows Flow state_dict")
    csm = any("transform_net.spline_predictor.0.weight" in b for b in blocks.values())   # sbi ContextSplineMap: scalar theta
    kind = "maf" if any("autoregressive_net.initial_layer.weight" in b for b in blocks.values()) else \
           "nsf" if (csm or any("transform_net.initial_layer.weight" in b for b in blocks.values())) else None
    if kind is None:
        raise KeyError("neither MaskedAffineAutoregressiveTransform nor PiecewiseRationalQuadraticCouplingTransform "
                       "parameters found")
    # ---- z-score buffers (sbi standardizing_transform: scale = 1/std, shift = -mean/std)
    if "_transform._transforms.0._scale" in sd:
        scale = sd["_transform._transforms.0._scale"].reshape(-1).astype(np.float64)
        shift = sd["_transform._transforms.0._shift"].reshape(-1).astype(np.float64)
        theta_std, theta_mean = 1.0 / scale, -shift / scale
    else:
        theta_std = theta_mean = None
    x_mean = sd.get("_embedding_net.0._mean")
    x_std = sd.get("_embedding_net.0._std")
    extra = [k for k in sd if k.startswith("_embedding_net.") and not k.startswith("_embedding_net.0.")]
    if extra:
        raise KeyError(f"the estimator has a trainable embedding net ({extra[0]} ...): import it separately and pass its "
                       "output width as the context")
    if kind == "maf":
        tidx = sorted(i for i, b in blocks.items() if "autoregressive_net.initial_layer.weight" in b)
        T = len(tidx)
        b0 = blocks[tidx[0]]
        H, D = b0["autoregressive_net.initial_layer.weight"].shape
        C = b0["autoregressive_net.context_layer.weight"].shape[1]
        NB = len({int(m.group(1)) for k in b0 for m in [re.match(r"autoregressive_net\.blocks\.(\d+)\.linear\.weight", k)] if m})
        perms = []
        for i in tidx:
            p = blocks.get(i + 1, {}).get("_permutation")
            perms.append(np.arange(D) if p is None else p.astype(np.int64))
        spec = FlowSpec(kind="maf", D=D, C=C, H=H, T=T, NB=NB, perms=np.stack(perms).astype(np.int32),
                        theta_mean=theta_mean, theta_std=theta_std, x_mean=x_mean, x_std=x_std, **spec_overrides)
        names = {"W0": "autoregressive_net.initial_layer.weight", "b0": "autoregressive_net.initial_layer.bias",
                 "Wc": "autoregressive_net.context_layer.weight", "bc": "autoregressive_net.context_layer.bias",
                 "Wf": "autoregressive_net.final_layer.weight", "bf": "autoregressive_net.final_layer.bias"}
        for k in range(NB):
            names[f"W{k + 1}"] = f"autoregressive_net.blocks.{k}.linear.weight"
            names[f"b{k + 1}"] = f"autoregressive_net.blocks.{k}.linear.bias"
        src = {t: (blocks[i], names) for t, i in enumerate(tidx)}
        _check_connectivity("maf", D, H, NB, blocks, tidx)
    elif csm:
        # one-parameter NSF (sbi build_nsf, x_numel == 1): every block is a coupling transform with mask [1] whose
        # transform_net is ContextSplineMap.spline_predictor = Sequential(Linear, ReLU, Linear, ReLU, Linear); no LULinear
        tidx = sorted(i for i, b in blocks.items() if "transform_net.spline_predictor.0.weight" in b)
        T = len(tidx)
        b0 = blocks[tidx[0]]
        H, C = b0["transform_net.spline_predictor.0.weight"].shape
        if "transform_net.spline_predictor.6.weight" in b0:
            raise KeyError("ContextSplineMap with hidden_layers > 1 (it repeats ONE Linear module) is not built")
        nout = b0["transform_net.spline_predictor.4.weight"].shape[0]
        K = num_bins if num_bins is not None else (nout + 1) // 3
        if 3 * K - 1 != nout:
            raise KeyError(f"spline_predictor has {nout} outputs: not 3K - 1 for K = {K}")
        spec = FlowSpec(kind="nsf", D=1, C=C, H=H, T=T, K=K, theta_mean=theta_mean, theta_std=theta_std,
                        x_mean=x_mean, x_std=x_std, **spec_overrides)
        names = {"csm.W0": "transform_net.spline_predictor.0.weight", "csm.b0": "transform_net.spline_predictor.0.bias",
                 "csm.W1": "transform_net.spline_predictor.2.weight", "csm.b1": "transform_net.spline_predictor.2.bias",
                 "csm.W2": "transform_net.spline_predictor.4.weight", "csm.b2": "transform_net.spline_predictor.4.bias"}
        src = {t: (blocks[i], names) for t, i in enumerate(tidx)}
    else:
        tidx = sorted(i for i, b in blocks.items() if "transform_net.initial_layer.weight" in b)
        T = len(tidx)
        b0 = blocks[tidx[0]]
        H, win = b0["transform_net.initial_layer.weight"].shape
        lu0 = blocks.get(tidx[0] + 1, {})
        if "unconstrained_upper_diag" in lu0:
            D = lu0["unconstrained_upper_diag"].shape[0]
        elif theta_std is not None:
            D = len(theta_std)
        else:
            raise KeyError("cannot infer the theta dimension (no LULinear block and no z-score buffers)")
        d_id0 = D // 2                      # block 0 transforms dims 0,2,4,...: identity dims = floor(D/2)
        C = win - d_id0
        d_tr0 = D - d_id0
        nout = b0["transform_net.final_layer.weight"].shape[0]
        K = num_bins if num_bins is not None else (nout // d_tr0 + 1) // 3
        if d_tr0 * (3 * K - 1) != nout:
            raise KeyError(f"final_layer has {nout} rows: not d_tr * (3K - 1) for d_tr = {d_tr0}, K = {K}")
        NB = len({int(m.group(1)) for k in b0 for m in [re.match(r"transform_net\.blocks\.(\d+)\.context_layer\.weight", k)] if m})
        spec = FlowSpec(kind="nsf", D=D, C=C, H=H, T=T, K=K, NB=NB, theta_mean=theta_mean, theta_std=theta_std,
                        x_mean=x_mean, x_std=x_std, **spec_overrides)
        names = {"Win": "transform_net.initial_layer.weight", "bin": "transform_net.initial_layer.bias",
                 "Wout": "transform_net.final_layer.weight", "bout": "transform_net.final_layer.bias"}
        for k in range(NB):
            names[f"blk{k}.Wg"] = f"transform_net.blocks.{k}.context_layer.weight"
            names[f"blk{k}.bg"] = f"transform_net.blocks.{k}.context_layer.bias"
            names[f"blk{k}.W1"] = f"transform_net.blocks.{k}.linear_layers.0.weight"
            names[f"blk{k}.b1"] = f"transform_net.blocks.{k}.linear_layers.0.bias"
            names[f"blk{k}.W2"] = f"transform_net.blocks.{k}.linear_layers.1.weight"
            names[f"blk{k}.b2"] = f"transform_net.blocks.{k}.linear_layers.1.bias"
        _check_connectivity("nsf", D, H, NB, blocks, tidx)
        lun = {"lu.lower": "lower_entries", "lu.upper": "upper_entries", "lu.udiag": "unconstrained_upper_diag",
               "lu.bias": "bias"}
        src = {}
        for t, i in enumerate(tidx):
            merged = dict(blocks[i])
            merged.update({"__lu__." + k: v for k, v in blocks.get(i + 1, {}).items()})
            nm = dict(names)
            nm.update({k: "__lu__." + v for k, v in lun.items()})
            src[t] = (merged, nm)
    flat = np.zeros(num_params(spec), dtype=np.float32)
    for name, shape, off in param_layout(spec):
        t = int(name[1:name.index(".")])
        leaf = name[name.index(".") + 1:]
        blk, nm = src[t]
        key = nm.get(leaf)
        if key is None or key not in blk:
            raise KeyError(f"upstream tensor for '{name}' ({key}) not found in transform {t}")
        arr = blk[key]
        n = int(np.prod(shape))
        if arr.size != n:
            raise KeyError(f"'{key}' of transform {t} has {arr.size} elements, the layout expects {shape}")
        flat[off:off + n] = arr.reshape(-1).astype(np.float32)
    return spec, flat


def state_dict_from_flat(spec: FlowSpec, flat, prefix: str = "") -> Dict[str, np.ndarray]:
    """The inverse mapping (upstream module paths; used to hand weights back to an sbi estimator off-box, and by the
    tests to round-trip the importer)."""
    flat = _np(flat).astype(np.float32)
    out: Dict[str, np.ndarray] = {}
    p = prefix + "_transform._transforms."
    out[p + "0._scale"] = (1.0 / spec.theta_std.astype(np.float64)).astype(np.float32)
    out[p + "0._shift"] = (-spec.theta_mean.astype(np.float64) / spec.theta_std.astype(np.float64)).astype(np.float32)
    out[prefix + "_embedding_net.0._mean"] = spec.x_mean.copy()
    out[prefix + "_embedding_net.0._std"] = spec.x_std.copy()
    inv = {}
    if spec.kind == "maf":
        inv = {"W0": "autoregressive_net.initial_layer.weight", "b0": "autoregressive_net.initial_layer.bias",
               "Wc": "autoregressive_net.context_layer.weight", "bc": "autoregressive_net.context_layer.bias",
               "Wf": "autoregressive_net.final_layer.weight", "bf": "autoregressive_net.final_layer.bias"}
        for k in range(spec.NB):
            inv[f"W{k + 1}"] = f"autoregressive_net.blocks.{k}.linear.weight"
            inv[f"b{k + 1}"] = f"autoregressive_net.blocks.{k}.linear.bias"
    elif spec.nsf_1d:
        inv = {"csm.W0": "transform_net.spline_predictor.0.weight", "csm.b0": "transform_net.spline_predictor.0.bias",
               "csm.W1": "transform_net.spline_predictor.2.weight", "csm.b1": "transform_net.spline_predictor.2.bias",
               "csm.W2": "transform_net.spline_predictor.4.weight", "csm.b2": "transform_net.spline_predictor.4.bias"}
    else:
        inv = {"Win": "transform_net.initial_layer.weight", "bin": "transform_net.initial_layer.bias",
               "Wout": "transform_net.final_layer.weight", "bout": "transform_net.final_layer.bias",
               "lu.lower": "+lower_entries", "lu.upper": "+upper_entries", "lu.udiag": "+unconstrained_upper_diag",
               "lu.bias": "+bias"}
        for k in range(spec.NB):
            inv[f"blk{k}.Wg"] = f"transform_net.blocks.{k}.context_layer.weight"
            inv[f"blk{k}.bg"] = f"transform_net.blocks.{k}.context_layer.bias"
            inv[f"blk{k}.W1"] = f"transform_net.blocks.{k}.linear_layers.0.weight"
            inv[f"blk{k}.b1"] = f"transform_net.blocks.{k}.linear_layers.0.bias"
            inv[f"blk{k}.W2"] = f"transform_net.blocks.{k}.linear_layers.1.weight"
            inv[f"blk{k}.b2"] = f"transform_net.blocks.{k}.linear_layers.1.bias"
    for name, shape, off in param_layout(spec):
        t = int(name[1:name.index(".")])
        leaf = name[name.index(".") + 1:]
        key = inv[leaf]
        i = t if spec.nsf_1d else 2 * t + (1 if key.startswith("+") else 0)   # (no LULinear blocks between the 1-D transforms)
        out[f"{p}1._transforms.{i}.{key.lstrip('+')}"] = flat[off:off + int(np.prod(shape))].reshape(shape).copy()
    if spec.kind == "maf":
        M0, Mh, Mf = made_masks(spec.D, spec.H)
        for t in range(spec.T):
            out[f"{p}1._transforms.{2 * t + 1}._permutation"] = spec.perms[t].astype(np.int64)
            q = f"{p}1._transforms.{2 * t}.autoregressive_net."
            out[q + "initial_layer.mask"] = M0.astype(np.float32)
            out[q + "final_layer.mask"] = Mf.astype(np.float32)
            for k in range(spec.NB):
                out[q + f"blocks.{k}.linear.mask"] = Mh.astype(np.float32)
    elif spec.nsf_1d:
        for t in range(spec.T):
            out[f"{p}1._transforms.{t}.transform_features"] = np.array([0], dtype=np.int64)
            out[f"{p}1._transforms.{t}.identity_features"] = np.array([], dtype=np.int64)
    else:
        for t in range(spec.T):
            d = np.arange(spec.D)
            out[f"{p}1._transforms.{2 * t}.transform_features"] = d[d % 2 == t % 2].astype(np.int64)
            out[f"{p}1._transforms.{2 * t}.identity_features"] = d[d % 2 != t % 2].astype(np.int64)
    return out


# ------------------------------------------------------------------------------------------------------------------------
# the lampe backend's flow: zuko.flows.NSF  (ref: src/synference/sbi_runner.py:5123-5125)
# ------------------------------------------------------------------------------------------------------------------------
# [UPSTREAM, module paths restated from the published zuko sources -- zuko is not installed here, the names are unpinned like
# everything else upstream]: ``zuko.flows.NSF(features=D, context=C, transforms=T, hidden_features=[H, H], bins=K)`` holds
#     transform.transforms.{t}.hyper.{0,2,4}.{weight,bias}     MaskedLinear layers of the MaskedMLP (ReLU at 1, 3)
#     transform.transforms.{t}.hyper.{0,2,4}.mask              their masks (buffers)
#     transform.transforms.{t}.order                           the ordering of transform t (buffer)
# (lampe's NPE prefixes ``flow.``, ltu-ili's wrapper its own: names are matched by suffix).  The checkpoint's masks and orders
# are COMPARED with what this engine builds from (D, C, H, t) -- unit h of a hidden layer has type h mod D, an output of type r
# sees inputs / units of type < r (first layer) / <= r (later layers), orders alternate 0..D-1 / D-1..0 -- and a difference
# raises: a flow that would be wired differently is not imported.  Standardising constants are not part of a zuko flow: they
# are taken from the keyword arguments (the z-scores ltu-ili's lampe loader applies around the flow), identity by default.
_ZUKO_RE = re.compile(r"transforms\.(\d+)\.hyper\.(\d+)\.(weight|bias|mask)$")
_ZUKO_ORDER_RE = re.compile(r"transforms\.(\d+)\.order$")


def zuko_masks(D: int, C: int, H: int, NP: int, order: np.ndarray):
    """[first hidden (H, D + C), second hidden (H, H), head (D * NP, H)] boolean masks of one transform (oracle/flows.py ar_masks)."""
    in_order = np.concatenate([np.asarray(order), np.full(C, -1)])
    typ = np.arange(H) % D
    out_type = np.repeat(np.asarray(order), NP)
    return [typ[:, None] > in_order[None, :], typ[:, None] >= typ[None, :], out_type[:, None] >= typ[None, :]]


def spec_and_flat_from_zuko_state_dict(state_dict: Mapping[str, object], tail_bound: float = 5.0, ar_slope: float = 1e-3,
                                       theta_mean=None, theta_std=None, x_mean=None, x_std=None) -> Tuple[FlowSpec, np.ndarray]:
    """(FlowSpec(kind="nsf_ar"), flat float32 vector) of a ``zuko.flows.NSF`` ``state_dict``; a head with TWO rows per dimension is a
    ``zuko.flows.MAF`` (MonotonicAffineTransform: [shift, scale]) and gives kind ``maf_ar``."""
    layers: Dict[int, Dict[int, Dict[str, np.ndarray]]] = {}
    orders: Dict[int, np.ndarray] = {}
    for k, v in state_dict.items():
        m = _ZUKO_RE.search(k)
        if m:
            layers.setdefault(int(m.group(1)), {}).setdefault(int(m.group(2)), {})[m.group(3)] = _np(v)
            continue
        m = _ZUKO_ORDER_RE.search(k)
        if m:
            orders[int(m.group(1))] = _np(v).astype(np.int64).reshape(-1)
    if not layers:
        raise KeyError("no 'transforms.N.hyper.M.weight' entries: this is not a zuko autoregressive flow state_dict")
    tidx = sorted(layers)
    if tidx != list(range(len(tidx))):
        raise KeyError(f"transform indices {tidx} are not 0..T-1 (an element-wise transform between them is not supported)")
    T = len(tidx)
    lidx = sorted(layers[0])
    if len(lidx) != 3:
        raise ValueError(f"{len(lidx) - 1} hidden layers: the HIP engine builds the two-hidden-layer hyper-network of the lampe loader")
    W0, W1, W2 = (layers[0][i]["weight"] for i in lidx)
    H = W0.shape[0]
    if W1.shape != (H, H) or W2.shape[1] != H:
        raise ValueError("hidden layers of different widths are not supported")
    if 0 not in orders:
        raise KeyError("transforms.0.order is missing")
    D = int(orders[0].size)
    C = W0.shape[1] - D
    if C < 1 or W2.shape[0] % D:
        raise ValueError(f"shapes do not fit an autoregressive NSF with context: W0 {W0.shape}, head {W2.shape}, D = {D}")
    NP = W2.shape[0] // D
    if NP != 2 and (NP + 1) % 3:
        raise ValueError(f"{NP} parameters per dimension are neither 2 (zuko MAF) nor 3 K - 1 (zuko NSF)")
    K = 8 if NP == 2 else (NP + 1) // 3
    f = lambda a, n, fill: np.full(n, fill, np.float32) if a is None else np.asarray(a, np.float32).reshape(n)
    spec = FlowSpec(kind="maf_ar" if NP == 2 else "nsf_ar", D=D, C=C, H=H, T=T, K=K, NB=2, tail_bound=float(tail_bound), ar_slope=float(ar_slope),
                    theta_mean=f(theta_mean, D, 0.0), theta_std=f(theta_std, D, 1.0), x_mean=f(x_mean, C, 0.0), x_std=f(x_std, C, 1.0))
    flat = np.zeros(num_params(spec), np.float32)
    lay = {n: (s, o) for n, s, o in param_layout(spec)}
    for t in tidx:
        want = np.arange(D) if t % 2 == 0 else np.arange(D)[::-1]
        if t not in orders or not np.array_equal(orders[t], want):
            raise ValueError(f"transform {t}: order {orders.get(t)} is not the alternating {want.tolist()} this engine builds")
        if sorted(layers[t]) != lidx:
            raise KeyError(f"transform {t}: layer indices {sorted(layers[t])} differ from transform 0's {lidx}")
        masks = zuko_masks(D, C, H, NP, want)
        for j, li in enumerate(lidx):
            ent = layers[t][li]
            if "weight" not in ent or "bias" not in ent:
                raise KeyError(f"transforms.{t}.hyper.{li}: weight / bias missing")
            shape, off = lay[f"t{t}.ar.W{j}"]
            if ent["weight"].shape != shape:
                raise ValueError(f"transforms.{t}.hyper.{li}.weight has shape {ent['weight'].shape}, expected {shape}")
            if "mask" in ent and not np.array_equal(ent["mask"].astype(bool), masks[j]):
                raise ValueError(f"transforms.{t}.hyper.{li}.mask differs from the connectivity this engine builds from (D, C, H): "
                                 "the checkpoint would be wired differently")
            flat[off:off + int(np.prod(shape))] = (ent["weight"].astype(np.float32) * masks[j]).reshape(-1)
            bshape, boff = lay[f"t{t}.ar.b{j}"]
            flat[boff:boff + bshape[0]] = ent["bias"].astype(np.float32).reshape(-1)
    return spec, flat


def zuko_state_dict_from_flat(spec: FlowSpec, flat, prefix: str = "") -> Dict[str, np.ndarray]:
    """The inverse mapping: zuko's module paths with the mask and order buffers (for handing weights back, and for the tests)."""
    if spec.kind not in ("nsf_ar", "maf_ar") or spec.NB != 2:
        raise ValueError("zuko_state_dict_from_flat takes an nsf_ar / maf_ar spec with two hidden layers")
    flat = _np(flat).astype(np.float32)
    out: Dict[str, np.ndarray] = {}
    NP = spec.ar_np
    lay = {n: (s, o) for n, s, o in param_layout(spec)}
    for t in range(spec.T):
        order = np.arange(spec.D) if t % 2 == 0 else np.arange(spec.D)[::-1]
        out[f"{prefix}transform.transforms.{t}.order"] = order.astype(np.int64).copy()
        masks = zuko_masks(spec.D, spec.C, spec.H, NP, order)
        for j, li in enumerate((0, 2, 4)):
            shape, off = lay[f"t{t}.ar.W{j}"]
            out[f"{prefix}transform.transforms.{t}.hyper.{li}.weight"] = flat[off:off + int(np.prod(shape))].reshape(shape).copy()
            bshape, boff = lay[f"t{t}.ar.b{j}"]
            out[f"{prefix}transform.transforms.{t}.hyper.{li}.bias"] = flat[boff:boff + bshape[0]].copy()
            out[f"{prefix}transform.transforms.{t}.hyper.{li}.mask"] = masks[j].copy()
    return out
