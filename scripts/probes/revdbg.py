import os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch, emap_amd
from conftest import net_state
mode = os.environ.get("EMAP_GRAD_MODE", "rev")
kw, state = net_state("d8w256L10")
res = {}
for prec in ["bf16x3"]:
    net = emap_amd.UDFNetwork(scale=1.0, precision=prec, **kw); net.load_state_dict(state); net = net.cuda()
    g = torch.Generator().manual_seed(5)
    for P in [32768]:
        x = (torch.rand(P, 3, generator=g) * 2 - 1).cuda()
        for rep in range(2):
            with torch.no_grad(): u, gr = net.hip_udf(x, with_grad=True)
            torch.cuda.synchronize()
            res[(prec, P, rep)] = (u.cpu().flatten(), gr.cpu().reshape(-1, 3))
torch.save(res, "/tmp/revdbg_%s.pt" % mode)
for (prec, P, rep) in list(res):
    if rep == 1:
        d = (res[(prec, P, 1)][1] != res[(prec, P, 0)][1])
        print(prec, P, "rep0 vs rep1: differing grad components per axis:", d.sum(dim=0).tolist())
if mode == "rev":
    ref = torch.load("/tmp/revdbg_fwd.pt")
    for k in res:
        du = (res[k][0] - ref[k][0]).abs() / ref[k][0].abs().max()
        dg = (res[k][1] - ref[k][1]).abs().max(dim=1).values / ref[k][1].abs().max()
        tol = 2e-2 if k[0] == "bf16" else 1e-3
        bad = torch.nonzero((du > tol) | (dg > 10 * tol) | ~torch.isfinite(du)).flatten()
        if len(bad): print("   ", [(int(i), [round(float(v), 3) for v in res[k][1][i]], [round(float(v), 3) for v in ref[k][1][i]]) for i in bad[:5]])
        badu = torch.nonzero((du > tol) | ~torch.isfinite(du)).flatten()
        print(k, "bad points:", len(bad), "of", len(du), "bad udf:", len(badu), "first:", bad[:8].tolist(), "tiles:", sorted(set((bad // 64).tolist()))[:12])
