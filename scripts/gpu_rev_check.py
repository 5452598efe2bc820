#!/usr/bin/env python3
"""A/B check of the reverse-mode grad kernel against the forward-mode one (run twice: EMAP_GRAD_MODE=fwd, then rev)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import emap_amd
from conftest import net_state
dev = torch.device("cuda:0")
mode = os.environ.get("EMAP_GRAD_MODE", "rev")
out = {}
for name in ["d8w256L10", "d8w256L6"]:
    for prec in ["f16x3", "bf16x3", "bf16", "f16"]:
        kw, state = net_state(name)
        net = emap_amd.UDFNetwork(scale=1.0, precision=prec, **kw); net.load_state_dict(state); net = net.to(dev)
        g = torch.Generator().manual_seed(5)
        for P in [1, 64, 777, 4099, 65536]:
            x = (torch.rand(P, 3, generator=g) * 2 - 1).to(dev)
            with torch.no_grad():
                u, gr = net.hip_udf(x, with_grad=True)
            torch.cuda.synchronize()
            out[(name, prec, P)] = (u.cpu(), gr.cpu())
            print(name, prec, P, "ok", float(u.abs().max()), float(gr.abs().max()), flush=True)
path = "/tmp/revcheck_%s.pt" % mode
torch.save(out, path)
other = "/tmp/revcheck_%s.pt" % ("fwd" if mode == "rev" else "rev")
if os.path.exists(other):
    ref = torch.load(other)
    for k in out:
        du = (out[k][0] - ref[k][0]).abs().max() / ref[k][0].abs().max()
        dg = (out[k][1] - ref[k][1]).abs().max() / ref[k][1].abs().max()
        print(k, "udf rel diff %.2e  grad rel diff %.2e" % (float(du), float(dg)))
