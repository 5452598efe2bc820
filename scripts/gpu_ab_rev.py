#!/usr/bin/env python3
"""Same-box A/B of the value + grad_x MLP kernels: emap_set_grad_mode in {rev (reverse sweep, 32x32x16 tiles), fwd (forward-mode tangents)};
EMAP_HIP_LIB selects another build of the library (scripts/build_variant.sh) for A/B of kernel changes.

Interleaved rounds in ONE process (emap_set_grad_mode between the calls), HIP-event medians, and the agreement of every variant
with the fp64 CPU oracle on the same points (sizes the oracle finishes in seconds) and with the reference golden g2.

    python scripts/gpu_ab_rev.py [f16x3 bf16x3 ...] [--points 65536] [--rounds 15]
"""
import argparse, json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import emap_amd                                   # noqa: E402
from emap_amd import synthetic                    # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("modes", nargs="*", default=["f16x3"])
    ap.add_argument("--points", type=int, default=65536)
    ap.add_argument("--rounds", type=int, default=15)
    ap.add_argument("--variants", default="rev,fwd")
    ap.add_argument("--no-oracle", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    kw = dict(d_in=3, d_out=1, d_hidden=256, n_layers=8, skip_in=(4,), multires=10, bias=0.5)
    state = synthetic.make_udf_state(seed=42, pert=0.02, **kw)
    g2 = np.load(os.path.join(ROOT, "tests", "golden", "g2_mlp.npz"))
    xg = torch.from_numpy(g2["x"])
    ref_u = torch.from_numpy(g2["d8w256L10.out"])[:, :1].double()
    ref_g = torch.from_numpy(g2["d8w256L10.grad"]).reshape(-1, 3).double()
    torch.manual_seed(3)
    x = torch.rand(args.points, 3) * 2.4 - 1.2
    x[:256] = xg
    n_or = 2048
    if not args.no_oracle:
        from oracle import emap_oracle as O       # the checker, not the thing measured
        cfg = O.UDFConfig(d_hidden=256, n_layers=8, multires=10)
        u64, g64 = O.udf_value_and_grad({k: v.double() for k, v in state.items()}, cfg, x[:n_or].double())
    xd = x.to(dev)
    rel = lambda a, b: float((a.double().cpu() - b).abs().max() / b.abs().max())
    variants = args.variants.split(",")
    for prec in args.modes:
        net = emap_amd.UDFNetwork(scale=1.0, precision=prec, **kw)
        net.load_state_dict(state)
        net = net.to(dev)
        out = {"prec": prec, "points": args.points, "lib": emap_amd._lib.LIB_PATH if hasattr(emap_amd, "_lib") else None}
        res = {}
        with torch.no_grad():
            for v in variants:
                emap_amd._lib.lib().emap_set_grad_mode({"fwd": 0, "rev": 1}[v])
                u, g = net.hip_udf(xd, with_grad=True)
                torch.cuda.synchronize()
                res[v] = (u.clone(), g.clone())
                e = {"g2_udf": rel(u[:256], ref_u), "g2_grad": rel(g[:256], ref_g), "finite": bool(torch.isfinite(u).all() and torch.isfinite(g).all())}
                if not args.no_oracle:
                    e["oracle_udf"] = rel(u[:n_or], u64)
                    e["oracle_grad"] = rel(g[:n_or], g64)
                u2, g2_ = net.hip_udf(xd, with_grad=True)
                e["bit_stable"] = bool(torch.equal(u, u2) and torch.equal(g, g2_))
                out[v] = e
            base = variants[0]
            for v in variants[1:]:
                out[f"{base}_vs_{v}"] = {"udf": rel(res[base][0], res[v][0].double().cpu()), "grad": rel(res[base][1], res[v][1].double().cpu())}
            # interleaved timing
            times = {v: [] for v in variants}
            for v in variants:                      # warm-up
                emap_amd._lib.lib().emap_set_grad_mode({"fwd": 0, "rev": 1}[v])
                for _ in range(3):
                    net.hip_udf(xd, with_grad=True)
            torch.cuda.synchronize()
            for _ in range(args.rounds):
                for v in variants:
                    emap_amd._lib.lib().emap_set_grad_mode({"fwd": 0, "rev": 1}[v])
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    for _ in range(4):
                        net.hip_udf(xd, with_grad=True)
                    e.record()
                    torch.cuda.synchronize()
                    times[v].append(s.elapsed_time(e) / 4 * 1e3)
            for v in variants:
                t = sorted(times[v])
                out[v]["us_median"] = round(t[len(t) // 2], 1)
                out[v]["us_min"] = round(t[0], 1)
        emap_amd._lib.lib().emap_set_grad_mode(-1)
        print(json.dumps(out))


main()
