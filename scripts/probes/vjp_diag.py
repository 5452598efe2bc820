"""Diagnostic: per-tensor error of emap_udf_vjp vs the fp64 mirror for u-only / g-only / combined seeds."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from test_gpu_backward import _hip_vjp, _mirror_param_grads
from test_gpu_parity import mk

for name in ("d4w128L10", "d8w256L10"):
    net, state, cfg = mk(name, "f16x3")
    gen = torch.Generator().manual_seed(9)
    P = 300
    x = torch.rand(P, 3, generator=gen) * 2 - 1
    wu, wg = torch.randn(P, generator=gen), torch.randn(P, 3, generator=gen) * 0.1
    for label, du, dg in (("u-only", wu, torch.zeros(P, 3)), ("g-only", torch.zeros(P), wg), ("both", wu, wg)):
        got = _hip_vjp(net, x, du, dg)
        ref = _mirror_param_grads(state, cfg, x, du, dg)
        rows = []
        for k, r in ref.items():
            e = float((got[k].double() - r).abs().max()); m = float(r.abs().max())
            rows.append((e / (m + 1e-30), k.replace("parametrizations.weight.original", "o"), m))
        rows.sort(reverse=True)
        print(name, label, " | ".join(f"{k}:{e:.1e}(max {m:.2g})" for e, k, m in rows[:5]))
