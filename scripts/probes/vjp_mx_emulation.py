#!/usr/bin/env python3
"""CPU emulation: what would MX-fp6 cross terms in the TRAINING sweep (udf_mlp_vjp_kernel: forward with a tangent column, backward with two
adjoint columns) do to dL/dW?  The candidate for the next round (docs/DESIGN_LOG_r1-r4.md par. 7): the sweep is MFMA-bound below the power cap, its
GEMMs are split-fp16 with f16 cross terms (3 passes), and its consumer - the weight-gradient GEMM on f16 hi parts - is good to 4e-4
of each tensor's maximum, an order of magnitude looser than what the value+gradient pass had to meet.

Every GEMM of the sweep goes through scripts/probes/mx6_emulation.py:gemm (exact products of the split operands, fp32 sums): "hh+hl+lh" =
as shipped, "k6:rne" = hi x hi in f16 + both cross terms as MX e2m3 blocks the way udf_mlp_rev32_kernel builds them.  The weight
gradients are then formed (a) in fp64 from the sweep's operands - isolating the sweep's own error - and (b) as wgrad.hip forms them, from the
f16 hi parts of both operands.  Errors: max |dW - dW_ref| / max |dW_ref| per layer, worst layer; dW_ref = the fp64 mirror.

    python scripts/probes/vjp_mx_emulation.py [--points 4096]          (CPU, ~2 min)"""
import argparse
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts", "probes"))
import mx6_emulation as MX  # noqa: E402   (patches precision_emulation.gemm)
import wgrad_precision_emulation as WG  # noqa: E402
from emap_amd import synthetic  # noqa: E402
from oracle import emap_oracle as O  # noqa: E402
from oracle import vjp_mirror as M  # noqa: E402


def sweep(state, cfg, x, du, dg, passes, passes_bwd=None):
    """The recurrences of WG.operands with every GEMM through MX.gemm(W, x, passes) (fp32 emulation); passes = None: fp64 exact."""
    dt = torch.float64 if passes is None else torch.float32
    Ws, bs = O._weights(state, cfg, dt)
    mm = (lambda W, v: v @ W.t()) if passes is None else (lambda W, v: MX.gemm(W.contiguous(), v.contiguous(), passes))
    mmb = mm if passes_bwd is None else (lambda W, v: MX.gemm(W.contiguous(), v.contiguous(), passes_bwd))      # the backward GEMMs (W^T)
    xs = (x * cfg.scale).to(dt)
    pe, dpe = M.pe_and_tangent(xs, dg.to(dt), cfg.multires)
    # the kernel evaluates the tangent / adjoint columns for K (du, dg), K a power of two that brings them into fp16's range; emulated by
    # scaling the tangent seed here and the adjoint seeds below, undone at the end (exact)
    k_t = 2.0 ** float(torch.floor(torch.log2(1.0 / dpe.abs().max().clamp_min(1e-30))))
    a, ap = pe, dpe * k_t
    ins, acts = [], []
    for l in range(cfg.n_lin):
        if l in cfg.skip_in:
            a = torch.cat([a, pe], 1) / np.sqrt(2)
            ap = torch.cat([ap, dpe * k_t], 1) / np.sqrt(2)
        ins.append((a, ap))
        z = mm(Ws[l], a) + bs[l]
        zp = mm(Ws[l], ap)
        if l < cfg.n_lin - 1:
            s = torch.sigmoid(100.0 * z)
            a, ap = F.softplus(z, beta=100), s * zp
            acts.append((s, ap))
        else:
            h, hp = z[:, :1], zp[:, :1]
    U1 = torch.sign(h)
    zb = torch.zeros(x.shape[0], Ws[-1].shape[0], dtype=dt)
    zb[:, :1] = du.reshape(-1, 1).to(dt) * U1 / cfg.scale * k_t
    zbp = torch.zeros_like(zb)
    zbp[:, :1] = U1
    out = {}
    for l in range(cfg.n_lin - 1, -1, -1):
        a, ap = ins[l]
        # dW = zb^T a + zb'^T a' ; undo the range scale: zb carries k_t, a' carries k_t
        out[l] = (torch.cat([zb / k_t, zbp], 0).t().contiguous().double(), torch.cat([a, ap / k_t], 0).t().contiguous().double())
        if l == 0:
            break
        ab, abp = mmb(Ws[l].t(), zb), mmb(Ws[l].t(), zbp)
        if l in cfg.skip_in:
            n_prev = Ws[l].shape[1] - pe.shape[1]
            ab, abp = ab[:, :n_prev] / np.sqrt(2), abp[:, :n_prev] / np.sqrt(2)
        s, apl = acts[l - 1]
        zb = s * ab + 100.0 * (1.0 - s) * apl * abp
        zbp = s * abp
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=4096)
    a = ap.parse_args()
    torch.manual_seed(0)
    kw = dict(d_in=3, d_out=1, d_hidden=256, n_layers=8, skip_in=(4,), multires=10, bias=0.5)
    st32 = synthetic.make_udf_state(seed=42, pert=0.02, **kw)
    st64 = {k: v.double() for k, v in st32.items()}
    cfg = O.UDFConfig(d_hidden=256, n_layers=8, multires=10)
    x = (torch.rand(a.points, 3) * 2.4 - 1.2)
    du, dg = torch.randn(a.points) * 1e-3, torch.randn(a.points, 3) * 1e-4
    ref = {l: Z @ A.t() for l, (Z, A) in sweep(st64, cfg, x.double(), du.double(), dg.double(), None).items()}
    f3 = "hh+hl+lh"
    for name, passes, pb in (("f16 cross terms (shipped)", f3, None), ("MX e2m3 cross terms, forward and backward GEMMs", "k6:rne", None),
                             ("forward f16 x 3, backward GEMMs MX e2m3 cross terms", f3, "k6:rne"), ("forward f16 x 3, backward GEMMs hi x hi only", f3, "hh"),
                             ("forward f16 x 3, backward GEMMs without the weights' lo parts", f3, "hh+hl"),
                             ("forward f16 x 3, backward GEMMs without the deltas' lo parts", f3, "hh+lh"),
                             ("only W_hi x_lo everywhere (the weights' lo parts dropped)", "hh+hl", None), ("only W_lo x_hi everywhere (the activations' lo parts dropped)", "hh+lh", None),
                             ("no cross terms at all (hi x hi)", "hh", None)):
        ops = sweep(st32, cfg, x, du, dg, passes, pb)
        e_exact = max(float(((Z @ A.t()) - ref[l]).abs().max() / ref[l].abs().max()) for l, (Z, A) in ops.items())
        e_wgrad = max(float(((WG.hi16(Z) @ WG.hi16(A).t()) - ref[l]).abs().max() / ref[l].abs().max()) for l, (Z, A) in ops.items())
        print(json.dumps({"sweep GEMMs": name, "points": a.points, "dW from the sweep's operands in fp64: worst layer": float(f"{e_exact:.3e}"),
                          "dW as wgrad forms it (f16 hi parts of both operands): worst layer": float(f"{e_wgrad:.3e}")}), flush=True)


if __name__ == "__main__":
    main()
