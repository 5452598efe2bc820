#!/usr/bin/env python3
"""Median time of ONE value launch (udf only) of the d8 w256 network over a range of point counts: where the tile-geometry rule of
udf_mlp_kernel.inc:launch_mlp_fs2_mode switches (A/B of that rule: EMAP_HIP_LIB selects the library).
usage: python scripts/gpu_time_value_sweep.py [precision] [P ...]"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import emap_amd
from emap_amd import synthetic

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
Ps = [int(v) for v in sys.argv[2:]] or [4096, 6144, 8192, 9216, 10240, 12288, 14336, 16384, 20480, 24576, 32768]
dev = torch.device("cuda:0")
kw = dict(d_in=3, d_out=1, d_hidden=256, n_layers=8, skip_in=(4,), multires=10, bias=0.5)
net = emap_amd.UDFNetwork(scale=1.0, precision=prec, **kw)
net.load_state_dict(synthetic.make_udf_state(seed=42, pert=0.02, **kw))
net = net.to(dev)
x = torch.rand(max(Ps), 3, device=dev) * 2 - 1
out = {}
with torch.no_grad():
    for P in Ps:
        xs = x[:P].contiguous()
        for _ in range(5):
            net.hip_udf(xs)
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(60)]
        for s, e in ev:
            s.record(); net.hip_udf(xs); e.record()
        torch.cuda.synchronize()
        ts = sorted(s.elapsed_time(e) for s, e in ev)
        out[P] = round(ts[len(ts) // 2] * 1e3, 1)
print(json.dumps({"prec": prec, "lib": os.environ.get("EMAP_HIP_LIB", "default"), "value_launch_us_by_points": out}))
