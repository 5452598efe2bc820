"""CPU emulation of the split-precision arithmetic of the reverse-sweep MLP kernel (udf_mlp_rev32.inc), used to decide
which MFMA passes the 1e-4 parity bar actually needs BEFORE any kernel is written.

Every GEMM of the kernel is  z = Wh.xh + (Wh.xl + Wl.xh)/2^11  with f16 hi parts and f16 lo parts stored x2^11, fp32
accumulation.  Here the operands are rounded exactly as the kernel rounds them (torch .half()), the products are formed in
fp32 (a product of two f16 numbers is exact in fp32) and summed by an fp32 matmul.  Variants drop or coarsen passes:

    fwd / bwd pass sets:  "hh+hl+lh" (shipped f16x3), "hh+lh" (weights split only), "hh" (single pass),
                          "hh+x8" (cross terms with e4m3 operands, fixed power-of-two scale), "hh+x8b" (per-32-block scale)

Checked against tests/golden/g2_mlp.npz (the reference's own output) and, on a larger random point set, against the fp64
oracle.  Output: one JSON line per variant (max error relative to the max of the reference, as the tests measure it).

    python scripts/probes/precision_emulation.py [--points 4096]
"""
import argparse, json, os, sys
import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import emap_oracle as O          # noqa: E402  (a probe, not product code)
from emap_amd import synthetic                # noqa: E402

LO = 2048.0


def split16(x):
    h = x.half().float()
    l = ((x - h) * LO).half().float()
    return h, l


def q8(x, block=None):
    """e4m3 quantisation with a power-of-two scale: one for the whole tensor (block=None) or per 32 consecutive K values."""
    if block is None:
        m = x.abs().max().clamp_min(1e-30)
        s = 2.0 ** torch.floor(torch.log2(256.0 / m))
        return (x * s).clamp(-448, 448).to(torch.float8_e4m3fn).float() / s
    sh = x.shape
    K = sh[-1]
    pad = (-K) % block
    xp = F.pad(x, (0, pad)).reshape(*sh[:-1], -1, block)
    m = xp.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    s = 2.0 ** torch.floor(torch.log2(256.0 / m))
    y = ((xp * s).clamp(-448, 448).to(torch.float8_e4m3fn).float() / s).reshape(*sh[:-1], -1)
    return y[..., :K]


def gemm(W, x, passes):
    """x (P,K) @ W(O,K)^T in the emulated arithmetic."""
    Wh, Wl = split16(W)
    xh, xl = split16(x)
    z = xh @ Wh.T
    if passes == "hh":
        return z
    if passes == "hh+lh":
        return z + (xh @ Wl.T) / LO
    if passes == "hh+hl":
        return z + (xl @ Wh.T) / LO
    if passes == "hh+hl+lh":
        return z + (xl @ Wh.T + xh @ Wl.T) / LO
    if passes in ("hh+hl+l8h", "hh+hl+l8bh"):      # only the weights' lo part stored as e4m3 (fixed scale / per-32 block scale); activations stay f16 hi + lo
        blk = 32 if passes.endswith("bh") else None
        return z + (xl @ Wh.T + xh @ q8(Wl, blk).T) / LO
    if passes in ("hh+x8", "hh+x8b"):
        blk = 32 if passes.endswith("b") else None
        return z + (q8(xl, blk) @ q8(Wh, blk).T + q8(xh, blk) @ q8(Wl, blk).T) / LO
    raise ValueError(passes)


def emulate(state, cfg, x, fwd, bwd, stash16=True):
    Ws, bs = O._weights(state, cfg, torch.float32)
    xs = x * cfg.scale
    pe = O.positional_encoding(xs, cfg.multires)
    P, d0 = pe.shape
    s2 = float(1.0 / np.sqrt(2))
    a = pe
    sig = []
    ins = []
    for l in range(cfg.n_lin):
        W = Ws[l]
        if l in cfg.skip_in:
            a = torch.cat([a, pe], 1)
            W = W * s2                      # the 1/sqrt(2) of the concat is folded into the packed weights
        ins.append((W, a.shape[1]))
        z = gemm(W, a, fwd) + bs[l]
        if l < cfg.n_lin - 1:
            a = F.softplus(z, beta=100)
            s = torch.sigmoid(100.0 * z)
            if stash16:
                s = torch.round(s * 65535.0) / 65535.0
            sig.append(s)
        else:
            h = z[:, :1]
    # reverse sweep
    H = cfg.d_hidden
    da = Ws[-1][:1, :].expand(P, -1).clone()                      # delta a[last-1] = row 0 of the last layer (fp32 seed)
    if (cfg.n_lin - 1) in cfg.skip_in:
        raise NotImplementedError
    dpe = torch.zeros(P, d0)
    for l in range(cfg.n_lin - 2, -1, -1):
        dz = sig[l] * da
        W, kin = ins[l]
        d_in = gemm(W.T.contiguous(), dz, bwd)                    # (P, kin)
        if l in cfg.skip_in:
            dpe = dpe + d_in[:, kin - d0:]
            da = d_in[:, :kin - d0]
        elif l == 0:
            dpe = dpe + d_in
        else:
            da = d_in
    # J_PE^T
    g = dpe[:, :3].clone()
    for i in range(cfg.multires):
        f = 2.0 ** i
        base = 3 + 6 * i
        g = g + dpe[:, base:base + 3] * (f * torch.cos(xs * f)) - dpe[:, base + 3:base + 6] * (f * torch.sin(xs * f))
    udf = h.abs() / cfg.scale
    grad = torch.sign(h) * g
    return udf, grad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=4096)
    args = ap.parse_args()
    torch.manual_seed(0)
    g2 = np.load(os.path.join(ROOT, "tests", "golden", "g2_mlp.npz"))
    kw = dict(d_in=3, d_out=1, d_hidden=256, n_layers=8, skip_in=(4,), multires=10, bias=0.5)
    state = synthetic.make_udf_state(seed=42, pert=0.02, **kw)
    cfg = O.UDFConfig(d_hidden=256, n_layers=8, multires=10)
    xg = torch.from_numpy(g2["x"])
    ref_u = torch.from_numpy(g2["d8w256L10.out"])[:, :1].double()
    ref_g = torch.from_numpy(g2["d8w256L10.grad"]).reshape(-1, 3).double()
    xr = torch.rand(args.points, 3) * 2.4 - 1.2
    st64 = {k: v.double() for k, v in state.items()}
    u64, g64 = O.udf_value_and_grad(st64, cfg, xr.double())
    rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())
    variants = [("hh+hl+lh", "hh+hl+lh"), ("hh+hl+lh", "hh+lh"), ("hh+hl+lh", "hh+hl"), ("hh+hl+lh", "hh"),
                ("hh+x8", "hh+x8"), ("hh+x8b", "hh+x8b"), ("hh+hl+lh", "hh+x8"), ("hh+x8", "hh"), ("hh+x8b", "hh"),
                ("hh+lh", "hh+lh"), ("hh", "hh"), ("hh+hl+l8h", "hh+hl+l8h"), ("hh+hl+l8bh", "hh+hl+l8bh")]
    for fwd, bwd in variants:
        u, g = emulate(state, cfg, xg, fwd, bwd)
        ur, gr = emulate(state, cfg, xr, fwd, bwd)
        print(json.dumps({"fwd": fwd, "bwd": bwd,
                          "g2_udf": rel(u, ref_u), "g2_grad": rel(g, ref_g),
                          "rand_udf_vs_fp64": rel(ur, u64), "rand_grad_vs_fp64": rel(gr, g64)}))


if __name__ == "__main__":
    main()
