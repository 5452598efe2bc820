import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import emap_amd
from conftest import net_state
from test_gpu_backward import _hip_vjp, _mirror_param_grads
from oracle import emap_oracle as O
for name, prec in (("d4w128L10", "f16x3"), ("d8w256L10", "f16x3e"), ("d8w256L10", "f16x3")):
    kw, state = net_state(name)
    net = emap_amd.UDFNetwork(scale=1.0, precision=prec, **kw)
    net.load_state_dict(state)
    net = net.to("cuda:0")
    cfg = O.UDFConfig(d_hidden=kw["d_hidden"], n_layers=kw["n_layers"], multires=kw["multires"])
    gen = torch.Generator().manual_seed(11)
    for P in (32, 777):
        x = torch.rand(P, 3, generator=gen) * 2 - 1
        du = torch.randn(P, generator=gen) * 1e-3
        dg = torch.randn(P, 3, generator=gen) * 1e-4
        got = _hip_vjp(net, x, du, dg)
        ref = _mirror_param_grads(state, cfg, x, du, dg)
        print(name, prec, P, {k.replace("parametrizations.weight.original", "w"): (int(torch.isnan(v).sum()), f"{float((v.double()-ref[k].double()).abs().max()/ref[k].abs().max()):.1e}") for k, v in got.items()})
