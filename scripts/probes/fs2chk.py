import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch, emap_amd
from conftest import net_state
kw, state = net_state("d8w256L10")
g = torch.Generator().manual_seed(5)
x = (torch.rand(65536, 3, generator=g) * 2 - 1).cuda()
nets = {}
for prec in ["f16x3", "bf16x3"]:
    net = emap_amd.UDFNetwork(scale=1.0, precision=prec, **kw); net.load_state_dict(state); nets[prec] = net.cuda()
with torch.no_grad():
    ref = nets["f16x3"].hip_udf(x, with_grad=False)[0].flatten().cpu()
    for rep in range(4):
        u = nets["bf16x3"].hip_udf(x, with_grad=False)[0].flatten().cpu()
        ug = nets["bf16x3"].hip_udf(x, with_grad=True)[0].flatten().cpu()
        for name, v in (("value", u), ("grad-call udf", ug)):
            d = (v - ref).abs() / ref.abs().max()
            bad = torch.nonzero(d > 1e-3).flatten()
            print(rep, name, "bad:", len(bad), "max rel", float(d.max()), "first tiles(64):", sorted(set((bad // 64).tolist()))[:8])
