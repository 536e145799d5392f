#!/usr/bin/env python
"""OFF-BOX tool: emit tests/golden/upstream_<model>.npz from a real sbi / nflows flow.

Run this where ``sbi`` is installed (it is NOT in the build image, and the GPU boxes have no network):

    pip install "sbi>=0.22"            # pulls pyknos / nflows
    python scripts/export_upstream_golden.py --out tests/golden
    pip install zuko && python scripts/export_upstream_golden.py --out tests/golden --zuko     # the lampe backend's NSF

For each of {maf, nsf} it builds the estimator exactly the way the reference does
(ref: src/synference/sbi_runner.py:5123-5146 -> ili.utils.load_nde_sbi -> sbi ``posterior_nn(model, hidden_features,
num_transforms, ...)``; ref: src/synference/custom_runner.py:320-326 ``estimator_builder(batch_x=, batch_theta=)``),
perturbs the weights away from their initial values (so that splines / LU are not the identity), and stores

    sd/<name>            every tensor of ``estimator.state_dict()``
    theta, x             evaluation batch (float32)
    log_prob             estimator log-density of (theta | x)                       [UPSTREAM arithmetic]
    z, theta_from_z      base noise and  transform.inverse(z, context=embedding(x)) [UPSTREAM arithmetic]
    logabsdet_inv        its log |det|
    meta                 json: model, sbi / nflows versions, builder kwargs

``tests/test_golden.py::test_upstream_golden_*`` picks these files up when they exist: the state dict is mapped by
``synference_amd.importer`` onto the flat vector, and both the CPU oracle (always) and the HIP kernels (on a GPU box)
must reproduce ``log_prob`` to 1e-4 and ``theta_from_z`` to fp32 tolerance.  That is the only route from
"parity unpinned" to pinned parity: nothing in this repository can stand in for running upstream code.
"""
import argparse
import json
import os

import numpy as np
import torch


def build(model, theta, x, hidden_features, num_transforms, num_bins):
    try:
        from sbi.neural_nets import posterior_nn          # sbi >= 0.23
    except ImportError:
        from sbi.utils import posterior_nn                # sbi 0.22
    kw = dict(model=model, hidden_features=hidden_features, num_transforms=num_transforms,
              z_score_theta="independent", z_score_x="independent")
    if model == "nsf":
        kw["num_bins"] = num_bins
    est = posterior_nn(**kw)(theta, x)
    return est, kw


def pip_freeze():
    """exact versions of the packages the arithmetic lives in (they are un-pinned upstream)"""
    import subprocess
    import sys
    try:
        txt = subprocess.run([sys.executable, "-m", "pip", "freeze"], capture_output=True, text=True, timeout=120).stdout
    except Exception:
        return []
    keep = ("sbi", "nflows", "pyknos", "torch", "numpy", "ltu-ili", "ili", "zuko", "lampe", "pyro")
    return [ln for ln in txt.splitlines() if ln.split("==")[0].split(" @")[0].lower() in keep]


def export_zuko(out_dir, seed):
    """The lampe backend's flow (ref: src/synference/sbi_runner.py:5123-5125 -> ili.utils.load_nde_lampe -> zuko.flows.NSF): needs
    only ``pip install zuko``.  No standardisation is applied here (identity z-scores): the vectors pin the flow arithmetic --
    masks, hidden-unit types, the monotonic rational-quadratic spline -- of oracle/flows.py kind "nsf_ar" and of csrc/sf_nsfar.hip.
    Files: tests/golden/upstream_zuko_<case>.npz, picked up by tests/test_golden.py through
    synference_amd.importer.spec_and_flat_from_zuko_state_dict."""
    import zuko
    for tag, D, C, H, T, K in (("nsf_cfg1", 5, 10, 50, 5, 8), ("nsf_small", 3, 4, 17, 2, 5), ("nsf_d1", 1, 6, 16, 3, 8),
                               ("maf_cfg1", 5, 10, 50, 5, 0), ("maf_small", 3, 4, 17, 2, 0)):   # K = 0: zuko.flows.MAF (kind "maf_ar")
        torch.manual_seed(seed)
        flow = (zuko.flows.NSF(features=D, context=C, transforms=T, hidden_features=[H, H], bins=K) if K else
                zuko.flows.MAF(features=D, context=C, transforms=T, hidden_features=[H, H]))
        with torch.no_grad():
            for p in flow.parameters():
                p.add_(0.3 * p.abs().mean().clamp_min(0.05) * torch.randn_like(p))
        flow.eval()
        rng = np.random.default_rng(seed)
        te = torch.as_tensor(rng.normal(size=(256, D)) * 1.5, dtype=torch.float32)
        xe = torch.as_tensor(rng.normal(size=(256, C)), dtype=torch.float32)
        z = torch.randn(256, D)
        with torch.no_grad():
            dist = flow(xe)                                  # NormalizingFlow(transform, base) conditioned on xe
            lp = dist.log_prob(te)
            th, lad = dist.transform.inv.call_and_ladj(z)    # base noise -> theta, log |d theta / d z|
        out = {"sd/" + k: v.detach().cpu().numpy() for k, v in flow.state_dict().items()}
        out.update(theta=te.numpy(), x=xe.numpy(), log_prob=lp.numpy().astype(np.float64), z=z.numpy(),
                   theta_from_z=th.numpy().astype(np.float64), logabsdet_inv=lad.numpy().astype(np.float64),
                   meta=np.array(json.dumps(dict(model="zuko_nsf" if K else "zuko_maf", case=tag, zuko=getattr(zuko, "__version__", "?"), torch=torch.__version__,
                                                 numpy=np.__version__, pip_freeze=pip_freeze(),
                                                 builder_kwargs=dict(features=D, context=C, transforms=T, hidden_features=[H, H], **({"bins": K} if K else {}))))))
        os.makedirs(out_dir, exist_ok=True)
        path = os.path.join(out_dir, f"upstream_zuko_{tag}.npz")
        np.savez_compressed(path, **out)
        print("wrote", path, "log_prob[:3] =", lp[:3].tolist())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="tests/golden")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--zuko", action="store_true", help="export the lampe backend's zuko.flows.NSF instead of the sbi flows")
    a = ap.parse_args()
    if a.zuko:
        export_zuko(a.out, a.seed)
        return
    import sbi
    try:
        import pyknos.nflows as nf
        nfv = getattr(nf, "__version__", "pyknos")
    except ImportError:
        import nflows as nf
        nfv = getattr(nf, "__version__", "nflows")
    cases = (("maf", "maf", 5, 10, 50, 5, 10),          # BASELINE configs[1]
             ("maf_span", "maf", 6, 10, 64, 3, 10),    # degree groups straddle the 16-row tiles of the sampler
             ("nsf", "nsf", 8, 20, 50, 5, 8),           # BASELINE configs[2]
             ("nsf_k10", "nsf", 5, 10, 30, 3, 10))      # sbi's default num_bins
    for tag, model, D, C, H, T, K in cases:
        torch.manual_seed(a.seed)
        rng = np.random.default_rng(a.seed)
        theta = torch.as_tensor(rng.normal(size=(2000, D)) * rng.uniform(0.5, 2.0, size=D) + rng.normal(size=D), dtype=torch.float32)
        x = torch.as_tensor(rng.normal(size=(2000, C)) * rng.uniform(0.5, 2.0, size=C) + rng.normal(size=C), dtype=torch.float32)
        est, kw = build(model, theta, x, H, T, K)
        flow = getattr(est, "net", est)                    # sbi >= 0.23 wraps the nflows Flow in NFlowsFlow(net=...)
        with torch.no_grad():
            for p in flow.parameters():                    # away from the init (identity splines / LU, tiny last layers)
                p.add_(0.3 * p.abs().mean().clamp_min(0.05) * torch.randn_like(p))
        flow.eval()
        te, xe = theta[:256], x[:256]
        z = torch.randn(256, D)
        with torch.no_grad():
            lp = flow.log_prob(te, context=xe)
            emb = flow._embedding_net(xe)
            th, lad = flow._transform.inverse(z, context=emb)
        extra = {}
        if model == "maf":
            # which scale parametrisation this nflows build uses (softplus(a) + 1e-3 since 0.14, sigmoid(a + 2) + 1e-3
            # before): with theta = 0 the first transform's log|det| is sum log(scale(final-layer bias))
            t0 = flow._transform._transforms[-1]._transforms[0]
            with torch.no_grad():
                a_ = t0.autoregressive_net.final_layer.bias.detach().view(-1, 2)[:, 0].double()
                _, ld0 = t0(torch.zeros(1, D), context=emb[:1])
            sp = float(torch.log(torch.nn.functional.softplus(a_) + 1e-3).sum())
            sg = float(torch.log(torch.sigmoid(a_ + 2.0) + 1e-3).sum())
            # (the first output's degree is 1: it sees no hidden unit, so its scale is a function of the bias alone;
            #  the other outputs depend on the context -- compare the bias-only output)
            extra["scale_fn_probe"] = np.array([sp, sg, float(ld0)])
        # [UPSTREAM] DirectPosterior.sample against a fixed prior box: acceptance rate per observation
        try:
            from sbi.inference.posteriors import DirectPosterior
            from sbi.utils import BoxUniform
            lo = theta.mean(0) - 1.0 * theta.std(0)
            hi = theta.mean(0) + 1.0 * theta.std(0)
            post = DirectPosterior(posterior_estimator=est, prior=BoxUniform(lo, hi))
            acc = []
            for i in range(8):
                with torch.no_grad():
                    smp = flow.sample(20000, context=xe[i:i + 1])[0]
                acc.append(float(((smp >= lo) & (smp <= hi)).all(-1).float().mean()))
            s8 = post.sample((64,), x=xe[0], show_progress_bars=False)
            extra.update(box_lo=lo.numpy(), box_hi=hi.numpy(), box_acceptance=np.array(acc),
                         direct_posterior_samples_x0=s8.numpy())
        except Exception as e:  # the surface moved between sbi versions: the density / inverse vectors above are what matters
            extra["direct_posterior_error"] = np.array(repr(e))
        out = {"sd/" + k: v.detach().cpu().numpy() for k, v in flow.state_dict().items()}
        out.update(extra)
        out.update(theta=te.numpy(), x=xe.numpy(), log_prob=lp.numpy().astype(np.float64), z=z.numpy(),
                   theta_from_z=th.numpy().astype(np.float64), logabsdet_inv=lad.numpy().astype(np.float64),
                   meta=np.array(json.dumps(dict(model=model, case=tag, sbi=sbi.__version__, nflows=str(nfv), torch=torch.__version__,
                                                 numpy=np.__version__, pip_freeze=pip_freeze(), builder_kwargs=kw,
                                                 wrapper=type(est).__name__))))
        os.makedirs(a.out, exist_ok=True)
        path = os.path.join(a.out, f"upstream_{tag}.npz")
        np.savez_compressed(path, **out)
        print("wrote", path, "log_prob[:3] =", lp[:3].tolist())


if __name__ == "__main__":
    main()
