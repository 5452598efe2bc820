#!/usr/bin/env python3
"""Run one MLP kernel configuration a few times (for rocprofv3 PMC passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import emap_amd
from conftest import net_state
dev = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
wg = (sys.argv[2] == "grad") if len(sys.argv) > 2 else True
P = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
kw, state = net_state("d8w256L10")
net = emap_amd.UDFNetwork(scale=1.0, precision=prec, **kw); net.load_state_dict(state); net = net.to(dev)
x = torch.rand(P, 3, device=dev) * 2 - 1
with torch.no_grad():
    net.hip_udf(x, with_grad=wg)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): net.hip_udf(x, with_grad=wg)
    e1.record()
torch.cuda.synchronize()
print(f"{prec} grad={int(wg)} P={P}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us")
