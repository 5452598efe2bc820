#!/usr/bin/env python3
"""s_memtime stamps of hidden layer 2 of the FIRST MLP pass inside the fused importance-sampling kernel (udf_mlp_fs2_kernel<256,f16x3,2,false,8,IS>, 512 rays)
in a -DEMAP_TIMELINE -DEMAP_TIMELINE_IS build (scripts/build_variant.sh tlis -DEMAP_TIMELINE -DEMAP_TIMELINE_IS [-DEMAP_IS_RING=0]); stamps as in fs2_timeline.py."""
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
r = emap_amd.UDFRendererBlending(None, net, emap_amd.SingleVarianceNetwork(0.3).to(dev), emap_amd.BetaNetwork(0.5, 0.3, 0.3, 5e-5, True, True, False).to(dev),
                                 64, 64, 0, 4, 1.0, device=dev)
ro, rd, near, far, ds = [t.to(dev) for t in synthetic.make_rays(512, seed=1)]
with torch.no_grad():
    for _ in range(5):
        r.render(ro, rd, near, far, ds, cos_anneal_ratio=1.0, perturb_overwrite=0, flip_saturation=0.9)
torch.cuda.synchronize()
L = C.CDLL(_lib.LIB_PATH)
n = 32 * 8 * 64
buf = (C.c_longlong * n)()
assert L.emap_debug_fs2_timeline(buf, n) == 0
t = np.frombuffer(buf, dtype=np.int64).reshape(32, 8, 64)
def seg(a, b):
    d = (t[:, :, b] - t[:, :, a]).reshape(-1)
    return {"median": int(np.median(d)), "min": int(d.min()), "max": int(d.max())}
names = [(f"K-step {k}", (0 if k == 0 else 4 + k), 5 + k) for k in range(8)] + [("K-loop (+ next prologue issue)", 0, 1), ("epilogue", 1, 2), ("output stores", 2, 3),
         ("wait barrier", 3, 4), ("layer total", 0, 4), ("PE block", 16, 17), ("9 layers", 17, 18), ("tile total", 16, 18)]
for k, a, b in names:
    print(f"{k:34s}", seg(a, b))
