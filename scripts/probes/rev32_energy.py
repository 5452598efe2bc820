#!/usr/bin/env python3
"""What the shader clock does under udf_mlp_rev32_kernel as a function of the DATA (the kernel sits at the 1400 W package power cap on real
data; scripts/probes/power_sample.sh): same instruction stream, same cycle count, operands zeroed selectively.  Needs the -DEMAP_TIMELINE
build (scripts/probes/rev32_timeline.py).  One process, configurations interleaved, several rounds."""
import ctypes as C, json, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import emap_amd
from emap_amd import synthetic, _lib

dev = torch.device("cuda:0")
kw = dict(d_in=3, d_out=1, d_hidden=256, n_layers=8, skip_in=(4,), multires=10, bias=0.5)
base = synthetic.make_udf_state(seed=42, pert=0.02, **kw)


def w16(state):   # weights exactly representable in fp16 after the weight norm: every lo fragment is zero
    st = dict(state)
    for k in list(st):
        if k.endswith("original1"):
            g = st[k.replace("original1", "original0")]
            w = (g * st[k] / st[k].norm(dim=1, keepdim=True)).half().float()
            st[k] = w
            st[k.replace("original1", "original0")] = w.norm(dim=1, keepdim=True)
    return st


def mk(state):
    net = emap_amd.UDFNetwork(scale=1.0, precision="f16x3", **kw)
    net.load_state_dict(state)
    return net.to(dev)


xr = torch.rand(65536, 3, device=dev) * 2 - 1
cfgs = {"real": (mk(base), xr), "lo(W)=0": (mk(w16(base)), xr), "x=0 (all columns equal)": (mk(base), xr * 0),
        "W=0,b=0": (mk({k: v * 0 for k, v in base.items()}), xr), "all zero": (mk({k: v * 0 for k, v in base.items()}), xr * 0)}
L = _lib.lib()
n = 32 * 4 * 64
buf = (C.c_longlong * n)()
rows = {k: [] for k in cfgs}
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    for name, (net, x) in cfgs.items():
        with torch.no_grad():
            for _ in range(6):
                net.hip_udf(x, with_grad=True)
        torch.cuda.synchronize()
        assert L.emap_debug_timeline(buf, n) == 0
        t = np.frombuffer(buf, dtype=np.int64).reshape(32, 4, 64)
        rt = (t[:, :, 33] - t[:, :, 32]).reshape(-1).astype(np.float64)
        ck = (t[:, :, 22] - t[:, :, 16]).reshape(-1).astype(np.float64)
        rows[name].append((float(np.median(ck)), float(np.median(ck / rt)) * 100.0, float(np.median(rt)) / 100.0))
for name, r in rows.items():
    a = np.array(r)
    print(json.dumps({"data": name, "cycles_per_tile": int(np.median(a[:, 0])), "shader_MHz": [round(v) for v in a[:, 1]],
                      "tile_us": [round(v, 1) for v in a[:, 2]], "median_MHz": round(float(np.median(a[:, 1]))), "median_tile_us": round(float(np.median(a[:, 2])), 1)}))
