#!/usr/bin/env python3
"""Reads the s_memtime stamps of a -DEMAP_SAMPLER_TIMELINE build of sampler_step_kernel (git apply scripts/probes/sampler_timeline.patch;
EMAP_VARIANT_UNITS=sampler scripts/build_variant.sh stl -DEMAP_SAMPLER_TIMELINE; git apply -R ...; EMAP_HIP_LIB=emap_amd/lib/stl/libemap_hip.so) after one
512-ray render: per phase of the LAST importance-sampling step (n = 112 -> 128 samples), median / min / max ticks over rays 0..63 (one wave per ray).
Stamps: 0 kernel entry, 1 inputs merged in LDS, 2 radii + true_cos, 3 occlusion alphas, 4 scan (visibility), 5 sdf2alpha both signs, 6 scan (transmittance),
7 weights, 8 sample_pdf (sum, normalise, scan, binary search, interpolation), 9 new samples stored."""
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
devn = emap_amd.SingleVarianceNetwork(0.3).to(dev)
bet = emap_amd.BetaNetwork(0.5, 0.3, 0.3, 5e-5, True, True, False).to(dev)
r = emap_amd.UDFRendererBlending(None, net, devn, bet, 64, 64, 0, 4, 1.0, device=dev)
ro, rd, near, far, ds = [v.to(dev) for v in synthetic.make_rays(512, seed=1)]
tr = synthetic.make_t_rand(512, seed=7).to(dev)
with torch.no_grad():
    for _ in range(5):
        r.render(ro, rd, near, far, ds, cos_anneal_ratio=1.0, flip_saturation=0.9, t_rand=tr)
torch.cuda.synchronize()
n = 2 * 64 * 16
buf = (C.c_longlong * n)()
L = _lib.lib()
L.emap_debug_sampler_timeline.restype = C.c_int
L.emap_debug_sampler_timeline.argtypes = [C.c_void_p, C.c_int]
assert L.emap_debug_sampler_timeline(buf, n) == 0
t = np.frombuffer(buf, dtype=np.int64).reshape(2, 64, 16)[0]
names = ["loads + merge of the previous step's samples", "radii, true_cos (1 division per interval)", "occlusion alphas (exp x2, division)", "scan: visibility product (fp64)",
         "sdf2alpha for +-udf (4 sigmoids, 2 divisions)", "scan: transmittance (fp64)", "weights", "sample_pdf: sum, normalise, scan, binary search, interpolation", "store"]
out = {}
for i, nm in enumerate(names):
    d = t[:, i + 1] - t[:, i]
    out[nm] = {"median": int(np.median(d)), "min": int(d.min()), "max": int(d.max())}
tot = t[:, 9] - t[:, 0]
out["kernel body total (ticks of s_memtime, 100 MHz x ~15-19 on these boxes: see profiles/r03_probe_clock.txt)"] = {"median": int(np.median(tot)), "min": int(tot.min()), "max": int(tot.max())}
print(json.dumps(out, indent=1))
