#!/usr/bin/env python3
"""s_memtime stamps of a -DEMAP_TIMELINE build of udf_mlp_vjp_kernel (python scripts/probes/vjp_timeline_instrument.py; scripts/build_variant.sh vtl -DEMAP_TIMELINE; git checkout emap_amd/csrc):
second tile of workgroups 0..31, every wave.  Forward layer 2: 0 start, 1 K-loop done, 2 epilogue done, 3 stash + slab stores issued, 4 barrier A,
5 exchange written, 6 barrier B; backward step b = 3: 8 start, 9 K-loop done, 10 slab landed, 11 epilogue done, 7 stash issued, 12 / 13 / 14 = barrier A /
exchange / barrier B; 16 tile start, 17 forward sweep done, 18 last layer + seeds done, 19 reverse sweep done."""
import ctypes as C, json, os, sys
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
n = 32 * 8 * 64
buf = (C.c_longlong * n)()
assert L.emap_debug_vjp_timeline(buf, n) == 0
t = np.frombuffer(buf, dtype=np.int64).reshape(32, 8, 64)
def seg(a, b):
    d = (t[:, :, b] - t[:, :, a]).reshape(-1)
    return {"median": int(np.median(d)), "min": int(d.min()), "max": int(d.max())}
# round 5 stamps: forward layer 2: 0 start, 1 K-loop done, 2 head request + epilogue done, 3 stash transposition + stores issued (publish entered), 4 exchange fragments
# written, 5 MX block conversion + entries written (SMX builds; else = 4), 6 barrier passed; reverse step b = 3: 8 start, 9 K-loop done, 10 slab wait + head request + epilogue
# done, 11 stash issued, 12 fragments written, 13 conversion done, 14 barrier; 16 tile start, 17 forward sweep done, 18 last layer + seeds, 19 reverse sweep done
names = [("fwd K-loop", 0, 1), ("fwd head request + epilogue", 1, 2), ("fwd stash transposition + stores issued", 2, 3), ("fwd exchange fragment writes", 3, 4),
         ("fwd MX block conversion + entry writes", 4, 5), ("fwd wait barrier", 5, 6), ("fwd layer total", 0, 6),
         ("bwd K-loop", 8, 9), ("bwd slab wait + head request + epilogue", 9, 10), ("bwd stash", 10, 11), ("bwd exchange fragment writes", 11, 12),
         ("bwd MX block conversion + entry writes", 12, 13), ("bwd wait barrier", 13, 14), ("bwd step total", 8, 14),
         ("PE + forward sweep", 16, 17), ("last layer + seeds", 17, 18), ("reverse sweep", 18, 19), ("tile total", 16, 19)]
print(json.dumps({k: seg(a, b) for k, a, b in names}, indent=1))
