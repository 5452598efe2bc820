"""Error of the reverse-mode kernel against the CPU oracle (fp32 torch) on 4096 of 20000 points, per precision mode."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch, emap_amd
from conftest import net_state
sys.path.insert(0, 'oracle')
import emap_oracle as O
for name in ["d8w256L10", "d8w256L10_init"]:
    kw, state = net_state(name)
    cfg = O.UDFConfig(d_hidden=kw["d_hidden"], n_layers=kw["n_layers"], multires=kw["multires"], scale=1.0)
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(20000, 3, generator=g) * 2 - 1)
    ur, gr = O.udf_value_and_grad(state, cfg, x[:4096])
    for prec in ["f16x3", "bf16", "f16"]:
        net = emap_amd.UDFNetwork(scale=1.0, precision=prec, **kw); net.load_state_dict(state); net = net.cuda()
        with torch.no_grad():
            u, gd = net.hip_udf(x.cuda(), with_grad=True)
            uf, gf = net.hip_udf(x[:4096].cuda(), with_grad=True)
        rel = lambda a, b: float((a.cpu().double() - b.double()).abs().max() / b.double().abs().max())
        print(name, prec, "reverse: udf %.1e grad %.1e | forward-mode: udf %.1e grad %.1e" % (rel(u[:4096], ur), rel(gd[:4096], gr), rel(uf, ur), rel(gf, gr)))
