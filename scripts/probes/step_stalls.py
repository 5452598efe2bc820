#!/usr/bin/env python3
"""Probe: per-step HIP-event times of N consecutive render() calls; prints the slowest steps and where they are."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
dev = torch.device("cuda:0")
r, state, kw = bench.build_renderer(dev, sys.argv[2] if len(sys.argv) > 2 else "f16x3")
from emap_amd import synthetic
ro, rd, near, far, ds = [t.to(dev) for t in synthetic.make_rays(512, seed=1)]
tr = synthetic.make_t_rand(512).to(dev)
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
host = []
with torch.no_grad():
    for s, e in ev:
        t0 = time.perf_counter()
        s.record(); r.render(ro, rd, near, far, ds, cos_anneal_ratio=1.0, flip_saturation=0.9, t_rand=tr); e.record()
        host.append(time.perf_counter() - t0)
torch.cuda.synchronize()
ts = [s.elapsed_time(e) for s, e in ev]
order = sorted(range(n), key=lambda i: -ts[i])[:8]
print("median ms", sorted(ts)[n // 2], "mean", sum(ts) / n)
print("slowest device steps:", [(i, round(ts[i], 2)) for i in order])
oh = sorted(range(n), key=lambda i: -host[i])[:8]
print("slowest host enqueue (ms):", [(i, round(host[i] * 1e3, 2)) for i in oh])
