#!/usr/bin/env python3
"""Per-job workgroup durations of wgrad_kernel (a -DEMAP_WGRAD_TIMING build: EMAP_VARIANT_UNITS=wgrad scripts/build_variant.sh wtime -DEMAP_WGRAD_TIMING)."""
import ctypes as C, json, os, subprocess, sys
import numpy as np
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
x = (torch.rand(65536, 3, device=dev) * 2 - 1)
for _ in range(3):
    for p in net.parameters():
        p.grad = None
    u, _, _ = net.udf(x)
    g = net.gradient(x)
    (u.sum() + (g * g).sum()).backward()
torch.cuda.synchronize()
L = _lib.lib()
n = 1024 * 2
buf = (C.c_longlong * n)()
assert L.emap_debug_wgrad_times(buf, n) == 0
t = np.frombuffer(buf, dtype=np.int64).reshape(1024, 2)
t = t[t[:, 1] > 0]
for j in sorted(set(t[:, 0])):
    d = t[t[:, 0] == j][:, 1] / 100.0
    print(f"job {j}: {len(d)} slices, us median {np.median(d):.1f} min {d.min():.1f} max {d.max():.1f}")
print("all:", len(t), "workgroups, max", t[:, 1].max() / 100.0, "us")
