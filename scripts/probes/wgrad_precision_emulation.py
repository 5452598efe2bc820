"""CPU emulation of the weight-gradient GEMMs  dW_l = sum_cols Z_l A_{l-1}^T  (wgrad.hip) with the operand precisions that were
built or considered (VERDICT r3 item 5 iii: what would dL/dtheta <= 1e-4 cost?).

The sweep leaves both operands of every layer in the stash as f16 hi parts only (11 bits); the judged bound on dL/dtheta is 1e-3 of
each tensor's largest entry.  Variants: hi x hi (shipped), one or both operands hi + lo (21 bits: twice / three times the stash
bytes and MFMA passes), and MX block formats with one scale per (feature, 32 columns) - e4m3 (half the bytes) and e2m3 (three
eighths).  Exact operands come from the fp64 mirror of the sweep (oracle/vjp_mirror.py: test infrastructure, not product code).

    python scripts/probes/wgrad_precision_emulation.py [--points 8192]
"""
import argparse, json, os, sys
import numpy as np
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import emap_oracle as O          # noqa: E402
from oracle import vjp_mirror as M           # noqa: E402
from emap_amd import synthetic               # noqa: E402


def operands(state, cfg, x, du, dg):
    """(Z_l, A_{l-1}) per layer with value and tangent columns concatenated along the column axis, fp64 - the mirror's recurrences."""
    dt = torch.float64
    Ws, bs = O._weights(state, cfg, dt)
    xs = x * cfg.scale
    pe, dpe = M.pe_and_tangent(xs, dg.to(dt), cfg.multires)
    a, ap = pe, dpe
    ins, acts = [], []
    for l in range(cfg.n_lin):
        if l in cfg.skip_in:
            a = torch.cat([a, pe], 1) / np.sqrt(2); ap = torch.cat([ap, dpe], 1) / np.sqrt(2)
        ins.append((a, ap))
        z = F.linear(a, Ws[l], bs[l]); zp = F.linear(ap, Ws[l])
        if l < cfg.n_lin - 1:
            s = torch.sigmoid(100.0 * z)
            a, ap = F.softplus(z, beta=100), s * zp
            acts.append((s, ap))
        else:
            h, hp = z[:, :1], zp[:, :1]
    U1 = torch.sign(h)
    zb = torch.zeros(x.shape[0], Ws[-1].shape[0], dtype=dt); zb[:, :1] = du.reshape(-1, 1).to(dt) * U1 / cfg.scale
    zbp = torch.zeros_like(zb); zbp[:, :1] = U1
    out = {}
    for l in range(cfg.n_lin - 1, -1, -1):
        a, ap = ins[l]
        out[l] = (torch.cat([zb, zbp], 0).t().contiguous(), torch.cat([a, ap], 0).t().contiguous())     # (features, columns)
        if l == 0:
            break
        ab, abp = zb @ Ws[l], zbp @ Ws[l]
        if l in cfg.skip_in:
            n_prev = Ws[l].shape[1] - pe.shape[1]
            ab, abp = ab[:, :n_prev] / np.sqrt(2), abp[:, :n_prev] / np.sqrt(2)
        s, apl = acts[l - 1]
        zb = s * ab + 100.0 * (1.0 - s) * apl * abp
        zbp = s * abp
    return out


def hi16(x):
    """f16 hi part after the launch-wide power-of-two scaling that keeps the values in fp16's normal range (wgrad_reduce divides it out)."""
    k = 2.0 ** torch.floor(torch.log2(1024.0 / x.abs().max().clamp_min(1e-300)))
    return (x * k).float().half().double() / k


def hilo(x):
    h = hi16(x)
    return h + hi16(x - h)


def q_e2m3(v):
    a = v.abs().clamp(max=7.5)
    step = torch.where(a < 2, torch.full_like(a, 0.125), torch.where(a < 4, torch.full_like(a, 0.25), torch.full_like(a, 0.5)))
    return torch.sign(v) * (torch.round(a / step) * step).clamp(max=7.5)


def mx(x, fmt):
    """one power-of-two scale per (feature, 32 consecutive columns)"""
    F_, C = x.shape
    pad = (-C) % 32
    xb = F.pad(x, (0, pad)).reshape(F_, -1, 32)
    m = xb.abs().amax(-1, keepdim=True).clamp_min(1e-300)
    if fmt == "e2m3":
        s = 2.0 ** (torch.floor(torch.log2(m * (8 / 7.5))) - 2)
        q = q_e2m3(xb / s)
    else:
        s = 2.0 ** (torch.floor(torch.log2(m * (512 / 448.0))) - 8)
        q = (xb / s).float().clamp(-448, 448).to(torch.float8_e4m3fn).double()
    return (q * s).reshape(F_, -1)[:, :C]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=8192)
    a = ap.parse_args()
    torch.manual_seed(0)
    kw = dict(d_in=3, d_out=1, d_hidden=256, n_layers=8, skip_in=(4,), multires=10, bias=0.5)
    state = {k: v.double() for k, v in synthetic.make_udf_state(seed=42, pert=0.02, **kw).items()}
    cfg = O.UDFConfig(d_hidden=256, n_layers=8, multires=10)
    x = (torch.rand(a.points, 3) * 2.4 - 1.2).double()
    du, dg = torch.randn(a.points).double() * 1e-3, torch.randn(a.points, 3).double() * 1e-4
    ops = operands(state, cfg, x, du, dg)
    variants = {"hi x hi (shipped)": (hi16, hi16), "Z hi+lo, A hi": (hilo, hi16), "Z hi, A hi+lo": (hi16, hilo), "hi+lo x hi+lo": (hilo, hilo),
                "MX e4m3 x e4m3": (lambda t: mx(t, "e4m3"), lambda t: mx(t, "e4m3")), "MX e2m3 x e2m3": (lambda t: mx(t, "e2m3"), lambda t: mx(t, "e2m3")),
                "Z hi16, A MX e2m3": (hi16, lambda t: mx(t, "e2m3"))}
    for name, (qz, qa) in variants.items():
        worst, per = 0.0, {}
        for l, (Z, A) in ops.items():
            exact = Z @ A.t()
            got = qz(Z) @ qa(A).t()
            e = float((got - exact).abs().max() / exact.abs().max())
            per[f"lin{l}"] = e
            worst = max(worst, e)
        print(json.dumps({"operands": name, "points": a.points, "worst_layer_rel_err_of_dW": worst, "per_layer": {k: float(f"{v:.2e}") for k, v in per.items()}}))


if __name__ == "__main__":
    main()
