#!/usr/bin/env python3
"""Reads the s_memtime stamps of a -DEMAP_TIMELINE build of udf_mlp_rev32_kernel (git apply scripts/probes/rev32_timeline.patch; scripts/build_variant.sh tl -DEMAP_TIMELINE; git apply -R ...;
EMAP_HIP_LIB=emap_amd/lib/tl/libemap_hip.so) and prints, per phase, the median / min / max cycles over the recorded waves.

Stamps (second tile of workgroups 0..31, every wave): forward layer 2: 0 start tile-pair 0, 1 K-loop issued, 2 epilogue 0 done,
3 K-loop 1 issued, 4 epilogue 1 done, 5 barrier A passed, 6 exchange written, 7 barrier B passed; (the K32-step stamps of round 3 are gone);
backward layer 3: 8..15 likewise; 16 tile start, 17 PE done, 18 forward sweep done, 19 last layer, 20 reverse sweep, 21 PE rows, 22 end;
round 5 (MX builds): 23 / 24 = mx_finish (block maxima -> scales, the two fp6 conversions per column tile) done in the forward / backward layer.
usage: rev32_timeline.py [--prec f16x3|f16x3m] [--zero] [--w16]"""
import ctypes as C, json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import emap_amd
from emap_amd import synthetic, _lib

dev = torch.device("cuda:0")
kw = dict(d_in=3, d_out=1, d_hidden=256, n_layers=8, skip_in=(4,), multires=10, bias=0.5)
PREC = sys.argv[sys.argv.index("--prec") + 1] if "--prec" in sys.argv else "f16x3"
net = emap_amd.UDFNetwork(scale=1.0, precision=PREC, **kw)
state = synthetic.make_udf_state(seed=42, pert=0.02, **kw)
if "--zero" in sys.argv:     # same instruction stream on all-zero operands: what the clock does without data toggling
    state = {k: v * 0 for k, v in state.items()}
if "--w16" in sys.argv:      # weights exactly representable in fp16 after the weight-norm: the lo fragments are all zero
    for k in list(state):
        if k.endswith("original1"):
            g = state[k.replace("original1", "original0")]
            w = (g * state[k] / state[k].norm(dim=1, keepdim=True)).half().float()
            state[k] = w
            state[k.replace("original1", "original0")] = w.norm(dim=1, keepdim=True)
net.load_state_dict(state)
net = net.to(dev)
x = torch.rand(65536, 3, device=dev) * 2 - 1
if "--zero" in sys.argv:
    x = x * 0
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
mxf = bool((t[:, :, 23] > 0).all())      # the forward layer published through publish6 (f16x3m)
mxb = bool((t[:, :, 24] > 0).all())
names = [("fwd K-loop tile-pair 0", 0, 1), ("fwd epilogue 0", 1, 2), ("fwd K-loop 1", 2, 3), ("fwd epilogue 1", 3, 4)] + \
        ([("fwd mx_finish (scales + fp6 conversions)", 4, 23), ("fwd wait barrier A", 23, 5)] if mxf else [("fwd wait barrier A", 4, 5)]) + [
         ("fwd exchange writes", 5, 6), ("fwd wait barrier B", 6, 7), ("fwd layer total", 0, 7),
         ("bwd K-loop 0", 8, 9), ("bwd epilogue 0", 9, 10), ("bwd K-loop 1", 10, 11), ("bwd epilogue 1", 11, 12)] + \
        ([("bwd mx_finish (scales + fp6 conversions)", 12, 24), ("bwd wait barrier A", 24, 13)] if mxb else [("bwd wait barrier A", 12, 13)]) + [
         ("bwd exchange writes", 13, 14), ("bwd wait barrier B", 14, 15), ("bwd layer total", 8, 15),
         ("PE block", 16, 17), ("forward sweep", 17, 18), ("last layer", 18, 19), ("reverse sweep", 19, 20), ("layer-0 PE rows", 20, 21), ("reduction + output", 21, 22),
         ("tile total", 16, 22)]
out = {"precision": PREC}
out.update({k: seg(a, b) for k, a, b in names})
rt = (t[:, :, 33] - t[:, :, 32]).reshape(-1).astype(np.float64)          # s_memrealtime: constant 100 MHz
ck = (t[:, :, 22] - t[:, :, 16]).reshape(-1).astype(np.float64)
out["s_memtime ticks per s_memrealtime tick (x 100 MHz = tick rate)"] = float(np.median(ck / rt))
out["tile total in us (s_memrealtime)"] = float(np.median(rt) / 100.0)
s_ev, e_ev = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
with torch.no_grad():
    s_ev.record()
    for _ in range(10):
        net.hip_udf(x, with_grad=True)
    e_ev.record()
torch.cuda.synchronize()
out["kernel us (10 back-to-back launches, HIP events)"] = s_ev.elapsed_time(e_ev) * 100.0
print(json.dumps(out, indent=1))
# per-workgroup view of one CU's worth: wave 0 of each workgroup, start phase of the forward layer relative to workgroup 0
print("fwd layer start (wave 0) per workgroup, relative to min:", ((t[:, 0, 0] - t[:, 0, 0].min())).tolist())
