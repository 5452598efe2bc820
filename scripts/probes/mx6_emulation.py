#!/usr/bin/env python3
"""CPU emulation of the MX (block-scaled fp8 / fp6) cross terms of udf_mlp_rev32_kernel<256, f16x3> - the numbers behind docs/DESIGN_LOG_r1-r4.md par. 6c.

Extends precision_emulation.py (exact products of f16 operands, fp32 sums) with GEMM variants whose cross terms W_hi x_lo + W_lo x_hi use
MX operands: 32-value blocks in the kernel's own block shape (the 32 values one lane holds of a K64-step: features 32 (2 Sigma + t) + R(r, hh)),
one E8M0 scale per block, elements rounded to nearest even onto e4m3 / e2m3 / e3m2, saturating.

  hh+mx:<fmt>   first estimate: lo parts as f16 (x 2^11) re-quantised per block, scale from the block's largest magnitude
  k6:<mode>     the kernel as built: hi parts RTZ ("rtz*") or RNE ("rne*") to f16; the lo block's scale is 2 x the hi block's scale x 2^-11
                ("*2x", what one v_cvt_scalef32 pass over both would give), 1 x ("*1x"), or taken from the lo block's own maximum (no suffix = shipped)

The script prints udf / grad_x errors (max abs / max |reference|) on the g2 golden points and on 4096 random points for: the f16x3 baseline,
MX cross terms in the reverse sweep only (shipped: k6:rne), and MX cross terms in the forward sweep as well (not shipped: the margin to 1e-4 is gone).
usage: python scripts/probes/mx6_emulation.py            (CPU, ~1 min)"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts", "probes"))
import precision_emulation as PE  # noqa: E402
from emap_amd import synthetic  # noqa: E402
from oracle import emap_oracle as O  # noqa: E402


def kperm(K):
    """feature order such that consecutive groups of 32 are the kernel's MX blocks (Sigma, hh)"""
    idx = []
    for Sg in range(K // 64):
        for hh in range(2):
            for t in range(2):
                for r in range(16):
                    idx.append(32 * (2 * Sg + t) + (r & 3) + 8 * (r >> 2) + 4 * hh)
    return torch.tensor(idx)


def q_e2m3(v):
    a = v.abs().clamp(max=7.5)
    step = torch.where(a < 2, torch.full_like(a, 0.125), torch.where(a < 4, torch.full_like(a, 0.25), torch.full_like(a, 0.5)))
    return torch.sign(v) * (torch.round(a / step) * step).clamp(max=7.5)


def q_e3m2(v):
    a = v.abs().clamp(max=28.0)
    e = torch.floor(torch.log2(a.clamp_min(1e-30))).clamp(min=-2, max=4)
    step = 2.0 ** (e - 2)
    return torch.sign(v) * (torch.round(a / step) * step).clamp(max=28.0)


def blocks(x):
    sh = x.shape
    K = sh[-1]
    xp = F.pad(x, (0, (-K) % 64))
    perm = kperm(xp.shape[-1])
    return xp[..., perm].reshape(*sh[:-1], -1, 32), perm, K


def unblocks(y, perm, K, sh):
    y = y.reshape(*sh[:-1], -1)
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(perm.numel())
    return y[..., inv][..., :K]


def q_mx(x, fmt):
    xb, perm, K = blocks(x)
    emax, vmax = {"e4m3": (8, 448.0), "e2m3": (2, 7.5), "e3m2": (4, 28.0)}[fmt]
    m = xb.abs().amax(-1, keepdim=True).clamp_min(1e-30) * ((2.0 ** (emax + 1)) / vmax)
    s = 2.0 ** (torch.floor(torch.log2(m)) - emax)
    v = xb / s
    q = v.clamp(-448, 448).to(torch.float8_e4m3fn).float() if fmt == "e4m3" else (q_e2m3(v) if fmt == "e2m3" else q_e3m2(v))
    return unblocks(q * s, perm, K, x.shape)


def rtz16(x):
    h = x.half()
    hb = h.view(torch.int16)
    hb = torch.where(h.float().abs() > x.abs(), hb - 1, hb)      # one ulp toward zero (sign-magnitude)
    return hb.view(torch.float16).float()


def scale_of(m):      # emap_common.h: mx6_scale_bits
    return 2.0 ** (torch.floor(torch.log2(m.clamp_min(2.0 ** -100) * 1.0666667)) - 2)


def q6_operand(x, mode):
    """(hi16, q6(hi), q6(lo) in true units) of an operand as the kernel builds its fragments"""
    hi = rtz16(x) if mode.startswith("rtz") else x.half().float()
    lo16 = ((x - hi) * 2048.0).half().float()
    hb, perm, K = blocks(hi)
    lb, _, _ = blocks(lo16)
    sh_ = scale_of(hb.abs().amax(-1, keepdim=True))
    sl = sh_ * 2.0 if mode.endswith("2x") else (sh_ if mode.endswith("1x") else scale_of(lb.abs().amax(-1, keepdim=True)))
    return hi, unblocks(q_e2m3(hb / sh_) * sh_, perm, K, x.shape), unblocks(q_e2m3(lb / sl) * sl / 2048.0, perm, K, x.shape)


_orig = PE.gemm


def gemm(W, x, passes):
    if passes.startswith("hh+mx:"):
        fmt = passes.split(":")[1]
        Wh, Wl = PE.split16(W)
        xh, xl = PE.split16(x)
        return xh @ Wh.T + (q_mx(xl, fmt) @ q_mx(Wh, fmt).T + q_mx(xh, fmt) @ q_mx(Wl, fmt).T) / PE.LO
    if passes.startswith("k6:"):
        Wh, Wh6, Wl6 = q6_operand(W, "rne")           # weights: RNE hi, own lo maximum (udf_mlp.hip: pack32_t_body)
        xh, xh6, xl6 = q6_operand(x, passes[3:])
        return xh @ Wh.T + xl6 @ Wh6.T + xh6 @ Wl6.T
    return _orig(W, x, passes)


PE.gemm = gemm


def main():
    torch.manual_seed(0)
    g2 = np.load(os.path.join(ROOT, "tests", "golden", "g2_mlp.npz"))
    kw = dict(d_in=3, d_out=1, d_hidden=256, n_layers=8, skip_in=(4,), multires=10, bias=0.5)
    state = synthetic.make_udf_state(seed=42, pert=0.02, **kw)
    cfg = O.UDFConfig(d_hidden=256, n_layers=8, multires=10)
    xg = torch.from_numpy(g2["x"])
    ref_u = torch.from_numpy(g2["d8w256L10.out"])[:, :1].double()
    ref_g = torch.from_numpy(g2["d8w256L10.grad"]).reshape(-1, 3).double()
    xr = torch.rand(4096, 3) * 2.4 - 1.2
    u64, g64 = O.udf_value_and_grad({k: v.double() for k, v in state.items()}, cfg, xr.double())
    rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())
    f3 = "hh+hl+lh"
    for fwd, bwd in [(f3, f3), (f3, "hh+mx:e4m3"), (f3, "hh+mx:e2m3"), (f3, "hh+mx:e3m2"), (f3, "k6:rtz2x"), (f3, "k6:rne1x"), (f3, "k6:rne"),
                     ("hh+mx:e4m3", "hh+mx:e4m3"), ("hh+mx:e2m3", "hh+mx:e2m3"), ("k6:rne", "k6:rne")]:
        u, g = PE.emulate(state, cfg, xg, fwd, bwd)
        ur, gr = PE.emulate(state, cfg, xr, fwd, bwd)
        print(json.dumps({"fwd": fwd, "bwd": bwd, "g2_udf": rel(u, ref_u), "g2_grad": rel(g, ref_g), "rand_udf": rel(ur, u64), "rand_grad": rel(gr, g64)}))


if __name__ == "__main__":
    main()
