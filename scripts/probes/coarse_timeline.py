#!/usr/bin/env python3
"""s_memtime stamps of hidden layer 2 of the COARSE value pass (udf_mlp_fs2_kernel<256,f16x3,4,false,4>: 64-point tiles, 4 waves x 2 pairs, two workgroups per CU;
32 768 points) in a -DEMAP_TIMELINE -DEMAP_TIMELINE_COARSE build.  Per pair pi: 5 + s + 24 pi = fragments of K-step s landed, 1 + 24 pi = K-loop done,
20 + 24 pi = epilogue done; 0 layer start, 2 both pairs done, 3 outputs stored (after barrier A), 4 barrier B passed."""
import ctypes as C, os, sys
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
x = torch.rand(32768, 3, device=dev) * 2 - 1
with torch.no_grad():
    for _ in range(5):
        net.hip_udf(x, with_grad=False)
torch.cuda.synchronize()
L = C.CDLL(_lib.LIB_PATH)
n = 32 * 8 * 64
buf = (C.c_longlong * n)()
assert L.emap_debug_fs2_timeline(buf, n) == 0
t = np.frombuffer(buf, dtype=np.int64).reshape(32, 8, 64)[:, :4]
def seg(a, b):
    d = (t[:, :, b] - t[:, :, a]).reshape(-1)
    return f"median {int(np.median(d)):6d}  min {int(d.min()):6d}  max {int(d.max()):6d}"
rows = [("pair 0 K-loop", 0, 1), ("pair 0 epilogue", 1, 20), ("pair 1 K-loop", 20, 25), ("pair 1 epilogue", 25, 44), ("wait barrier A + stores", 2, 3), ("wait barrier B", 3, 4),
        ("layer total", 0, 4), ("PE block", 16, 17), ("9 layers", 17, 18), ("tile total", 16, 18)]
rows[2:2] = [(f"pair 0 K-step {k}", (0 if k == 0 else 4 + k), 5 + k) for k in range(8)]
for k, a, b in rows:
    print(f"{k:28s}", seg(a, b))
