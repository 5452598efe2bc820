"""Run-to-run determinism stress of the large MLP kernels (DESIGN.md par. 3.1 open issue): repeated launches on the same
inputs must be bit-identical.  usage: determinism.py <prec> [rev|vjp] [points] [reps]   (EMAP_GRAD_MODE=rev forces the reverse kernel)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
prec, what = sys.argv[1], sys.argv[2]
P = int(sys.argv[3]) if len(sys.argv) > 3 else 524288
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 8
from test_gpu_parity import mk
net, state, cfg = mk("d8w256L10", prec)
gen = torch.Generator().manual_seed(5)
x = (torch.rand(P, 3, generator=gen) * 2 - 1).cuda()
if what == "rev":
    ref = None; bad = 0
    for i in range(reps):
        u, g = net.hip_udf(x, with_grad=True)
        torch.cuda.synchronize()
        if ref is None: ref = (u.clone(), g.clone())
        else:
            nd = int((g != ref[1]).any(dim=1).sum()) + int((u != ref[0]).sum())
            bad += nd
    print(f"{prec} rev P={P} reps={reps}: differing points over all repeats: {bad}")
else:
    from test_gpu_backward import _hip_vjp
    du = torch.randn(P, generator=gen) * 1e-3; dg = torch.randn(P, 3, generator=gen) * 1e-4
    ref = None; bad = 0
    for i in range(reps):
        out = _hip_vjp(net, x.cpu(), du, dg)
        flat = torch.cat([v.reshape(-1) for v in out.values()])
        if ref is None: ref = flat
        else: bad += int((flat != ref).sum())
    print(f"{prec} vjp P={P} reps={reps}: differing gradient entries over all repeats: {bad}")
