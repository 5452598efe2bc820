import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch, emap_amd
from conftest import net_state
mode = os.environ.get("EMAP_GRAD_MODE", "rev")
kw, state = net_state("d8w256L10")
res = {}
for prec in ["bf16x3", "f16x3"]:
    net = emap_amd.UDFNetwork(scale=1.0, precision=prec, **kw); net.load_state_dict(state); net = net.cuda()
    g = torch.Generator().manual_seed(5)
    for P in [64, 777, 65536]:
        x = (torch.rand(P, 3, generator=g) * 2 - 1).cuda()
        for rep in range(4):
            with torch.no_grad(): u, gr = net.hip_udf(x, with_grad=True)
            torch.cuda.synchronize()
            res[(prec, P, rep, len(res))] = u.cpu().flatten()
torch.save(res, "/tmp/revdbg2_%s.pt" % mode)
if mode == "rev":
    ref = torch.load("/tmp/revdbg2_fwd.pt")
    for k in res:
        du = (res[k] - ref[k]).abs() / ref[k].abs().max()
        bad = torch.nonzero((du > 1e-3) | ~torch.isfinite(du)).flatten()
        tl = torch.tensor(sorted(set((bad // 64).tolist())))
        hist = [int(((tl >= a) & (tl < a + 128)).sum()) for a in range(0, 1024, 128)] if len(tl) else []
        lanes = sorted(set((bad % 16).tolist()))
        if len(bad): print("   sample bad:", [(int(i), float(res[k][i]), float(ref[k][i])) for i in bad[:6]], "nonfinite:", int((~torch.isfinite(res[k])).sum()))
        print(k, "bad udf:", len(bad), "of", len(du), "bad tiles per 128:", hist, "j:", lanes)
