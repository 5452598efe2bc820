#!/usr/bin/env python3
"""Reads the s_memtime stamps of a -DEMAP_TIMELINE build of udf_mlp_rev32_kernel (scripts/build_variant.sh tl -DEMAP_TIMELINE;
EMAP_HIP_LIB=emap_amd/lib/tl/libemap_hip.so) and prints, per phase, the median / min / max cycles over the recorded waves.

Stamps (second tile of workgroups 0..31, every wave): forward layer 2: 0 start tile-pair 0, 1 K-loop issued, 2 epilogue 0 done,
3 K-loop 1 issued, 4 epilogue 1 done, 5 barrier A passed, 6 exchange written, 7 barrier B passed; 24..31 the K32-steps of K-loop 0;
backward layer 3: 8..15 likewise; 16 tile start, 17 PE done, 18 forward sweep done, 19 last layer, 20 reverse sweep, 21 PE rows, 22 end."""
import ctypes as C, json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import emap_amd
from emap_amd import synthetic, _lib

dev = torch.device("cuda:0")
kw = dict(d_in=3, d_out=1, d_hidden=256, n_layers=8, skip_in=(4,), multires=10, bias=0.5)
net = emap_amd.UDFNetwork(scale=1.0, precision="f16x3", **kw)
net.load_state_dict(synthetic.make_udf_state(seed=42, pert=0.02, **kw))
net = net.to(dev)
x = torch.rand(65536, 3, device=dev) * 2 - 1
with torch.no_grad():
    for _ in range(5):
        net.hip_udf(x, with_grad=True)
torch.cuda.synchronize()
L = _lib.lib()
n = 32 * 4 * 64
buf = (C.c_longlong * n)()
rc = L.emap_debug_timeline(buf, n)
assert rc == 0, rc
t = np.frombuffer(buf, dtype=np.int64).reshape(32, 4, 64)
def seg(a, b):
    d = (t[:, :, b] - t[:, :, a]).reshape(-1)
    return {"median": int(np.median(d)), "min": int(d.min()), "max": int(d.max())}
names = [("fwd K-loop tile-pair 0", 0, 1), ("fwd epilogue 0", 1, 2), ("fwd K-loop 1", 2, 3), ("fwd epilogue 1", 3, 4), ("fwd wait barrier A", 4, 5),
         ("fwd exchange writes", 5, 6), ("fwd wait barrier B", 6, 7), ("fwd layer total", 0, 7),
         ("bwd K-loop 0", 8, 9), ("bwd epilogue 0", 9, 10), ("bwd K-loop 1", 10, 11), ("bwd epilogue 1", 11, 12), ("bwd wait barrier A", 12, 13),
         ("bwd exchange writes", 13, 14), ("bwd wait barrier B", 14, 15), ("bwd layer total", 8, 15),
         ("PE block", 16, 17), ("forward sweep", 17, 18), ("last layer", 18, 19), ("reverse sweep", 19, 20), ("layer-0 PE rows", 20, 21), ("reduction + output", 21, 22),
         ("tile total", 16, 22)]
out = {k: seg(a, b) for k, a, b in names}
for s in range(7):
    out[f"fwd K-loop 0 step {s}->{s+1}"] = seg(24 + s, 25 + s)
out["fwd K-loop 0 last step -> issued"] = seg(31, 1)
print(json.dumps(out, indent=1))
# per-workgroup view of one CU's worth: wave 0 of each workgroup, start phase of the forward layer relative to workgroup 0
print("fwd layer start (wave 0) per workgroup, relative to min:", ((t[:, 0, 0] - t[:, 0, 0].min())).tolist())
