#!/usr/bin/env python3
"""Round 6, item 4: render() from a hipGraph with importance_sample as the launch chain (0), by the launcher's size rule (1) and fused at
every size (2), per launch shape.  Run ON THE GPU BOX:  python scripts/r6/gpu_fused_ab.py > gpurun_out/r6_fused_ab.txt"""
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


def timed(fn, steps=60, warmup=20):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


if len(sys.argv) > 1:      # e.g. "640,64,50,5 768,64,64,4": N, n_samples, n_importance, up_sample_steps
    shapes_arg = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
shapes = [(512, 64, 64, 4), (1024, 64, 64, 4), (1024, 64, 50, 5), (2048, 64, 64, 4), (2048, 64, 50, 5), (4096, 64, 64, 4), (4096, 64, 50, 5), (8192, 64, 64, 4)]
if len(sys.argv) > 1:
    shapes = shapes_arg
for N, ns, ni, K in shapes:
    r = emap_amd.UDFRendererBlending(None, net, devn, bet, ns, ni, 0, K, 1.0, device=dev)
    ro, rd, near, far, ds = [t.contiguous().to(dev) for t in synthetic.make_rays(N, seed=1)]
    tr = synthetic.make_t_rand(N).to(dev)
    row = {}
    for rep in range(2):                      # interleaved twice: box drift shows as a difference between the repeats
        for mode in (0, 1, 2):
            L.emap_set_fused_sampling(mode)
            g = r.capture(ro, rd, near, far, ds, cos_anneal_ratio=1.0, flip_saturation=0.9, t_rand=tr)
            row.setdefault(mode, []).append(timed(g))
            del g
    L.emap_set_fused_sampling(1)
    S = ns + ni
    print(f"N={N:5d} S={S} (m={ni // K:2d}, K={K}):  " + "  ".join(
        f"mode {m}: {min(v):.4f} ms ({N * S / min(v) / 1e3:.3e}/s)" for m, v in row.items()), flush=True)
