#!/usr/bin/env python3
"""s_memtime stamps of the narrow value pass (udf_mlp_fs2_kernel<256,f16x3,2,false,8>, 8192 points) in a -DEMAP_TIMELINE build
(round 6: the hooks are in the tree, `scripts/build_variant.sh tl -DEMAP_TIMELINE`, run with EMAP_HIP_LIB=emap_amd/lib/tl/libemap_hip.so): layer 2:
0 start, 5 + s fragments of K-step s landed, 1 K-loop done (+ next prologue issued), 2 epilogue done, 3 outputs stored, 4 barrier passed; 16 tile start,
17 PE done, 18 last layer done."""
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
x = torch.rand(8192, 3, device=dev) * 2 - 1
with torch.no_grad():
    for _ in range(5):
        net.hip_udf(x, with_grad=False)
torch.cuda.synchronize()
L = _lib.lib()
n = 32 * 8 * 64
buf = (C.c_longlong * n)()
import ctypes
L = ctypes.CDLL(_lib.LIB_PATH)
assert L.emap_debug_fs2_timeline(buf, n) == 0
t = np.frombuffer(buf, dtype=np.int64).reshape(32, 8, 64)
def seg(a, b):
    d = (t[:, :, b] - t[:, :, a]).reshape(-1)
    return {"median": int(np.median(d)), "min": int(d.min()), "max": int(d.max())}
names = [(f"K-step {k}: start -> fragments landed", (0 if k == 0 else 4 + k), 5 + k) for k in range(8)] + [("K-loop (+ next prologue issue)", 0, 1), ("epilogue", 1, 2), ("output stores", 2, 3), ("wait barrier", 3, 4), ("layer total", 0, 4),
         ("PE block", 16, 17), ("9 layers", 17, 18), ("tile total", 16, 18)]
print(json.dumps({k: seg(a, b) for k, a, b in names}, indent=1))
