#!/usr/bin/env python3
"""Round 6, item 5: render() from a hipGraph with render_core's tail as its own launch (0) or fused into the value + grad_x kernel (1).
Run ON THE GPU BOX:  python scripts/r6/gpu_comp_ab.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import emap_amd
from emap_amd import synthetic, _lib

dev = torch.device("cuda:0")
kw = dict(d_in=3, d_out=1, d_hidden=256, n_layers=8, skip_in=(4,), multires=10, bias=0.5)
net = emap_amd.UDFNetwork(scale=1.0, precision="f16x3", **kw)
net.load_state_dict(synthetic.make_udf_state(seed=42, pert=0.02, **kw))
net = net.to(dev)
devn = emap_amd.SingleVarianceNetwork(0.3).to(dev)
bet = emap_amd.BetaNetwork(0.5, 0.3, 0.3, 5e-5, True, True, False).to(dev)
L = _lib.lib()


def timed(fn, steps=200, warmup=100):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


for N, ns, ni, K in [(512, 64, 64, 4), (1024, 64, 64, 4), (1024, 64, 50, 5), (4096, 64, 64, 4)]:
    r = emap_amd.UDFRendererBlending(None, net, devn, bet, ns, ni, 0, K, 1.0, device=dev)
    ro, rd, near, far, ds = [t.contiguous().to(dev) for t in synthetic.make_rays(N, seed=1)]
    tr = synthetic.make_t_rand(N).to(dev)
    row = {}
    for rep in range(3):
        for mode in (0, 1):
            L.emap_set_fused_composite(mode)
            for red in (False, True):
                g = r.capture(ro, rd, near, far, ds, cos_anneal_ratio=1.0, flip_saturation=0.9, t_rand=tr, reduced=red)
                row.setdefault((mode, red), []).append(timed(g))
                del g
    L.emap_set_fused_composite(1)
    S = ns + ni
    print(f"N={N:5d} S={S}:  " + "  ".join(f"{'fused' if m else 'separate'}{' reduced' if red else ''}: {min(v):.4f} ms" for (m, red), v in row.items()), flush=True)
