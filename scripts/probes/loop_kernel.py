#!/usr/bin/env python3
"""Runs one kernel of the path in a tight loop for a few seconds (for scripts/probes/power_sample.sh):
   loop_kernel.py rev|value|train [--zero] [--seconds 6]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import emap_amd
from emap_amd import synthetic
dev = torch.device("cuda:0")
kw = dict(d_in=3, d_out=1, d_hidden=256, n_layers=8, skip_in=(4,), multires=10, bias=0.5)
state = synthetic.make_udf_state(seed=42, pert=0.02, **kw)
if "--zero" in sys.argv:
    state = {k: v * 0 for k, v in state.items()}
net = emap_amd.UDFNetwork(scale=1.0, precision="f16x3", **kw)
net.load_state_dict(state)
net = net.to(dev)
x = torch.rand(65536, 3, device=dev) * 2 - 1
if "--zero" in sys.argv:
    x = x * 0
secs = float(sys.argv[sys.argv.index("--seconds") + 1]) if "--seconds" in sys.argv else 6.0
grad = sys.argv[1] != "value"
t0 = time.time(); n = 0
with torch.no_grad():
    while time.time() - t0 < secs:
        for _ in range(50):
            net.hip_udf(x, with_grad=grad)
        torch.cuda.synchronize(); n += 50
print(f"{n} launches, {(time.time() - t0) / n * 1e6:.1f} us per launch (host clock, includes launch overhead)")
