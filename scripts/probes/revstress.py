import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch, emap_amd
from conftest import net_state
kw, state = net_state("d8w256L10")
precs = sys.argv[1].split(",")
for prec in precs:
    net = emap_amd.UDFNetwork(scale=1.0, precision=prec, **kw); net.load_state_dict(state); net = net.cuda()
    g = torch.Generator().manual_seed(7)
    for P in [32768, 65536, 200000, 524288]:
        x = (torch.rand(P, 3, generator=g) * 2 - 1).cuda()
        outs = []
        with torch.no_grad():
            for rep in range(6):
                u, gr = net.hip_udf(x, with_grad=True)
                outs.append((u.clone(), gr.clone()))
        torch.cuda.synchronize()
        nd_u = sum(int((outs[i][0] != outs[0][0]).sum()) for i in range(1, 6))
        nd_g = sum(int((outs[i][1] != outs[0][1]).sum()) for i in range(1, 6))
        print(prec, P, "nondeterministic elements over 5 repeats: udf", nd_u, "grad", nd_g, flush=True)
