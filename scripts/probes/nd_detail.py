"""Where and how much do two identical launches of the bf16x3 reverse kernel differ?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from test_gpu_parity import mk
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
P = int(sys.argv[2]) if len(sys.argv) > 2 else 131072
from emap_amd import _lib
_lib.lib().emap_set_grad_mode(1)   # reverse sweep regardless of size (the environment variable is read once, at load)
net, state, cfg = mk("d8w256L10", prec)
gen = torch.Generator().manual_seed(5)
x = (torch.rand(P, 3, generator=gen) * 2 - 1).cuda()
outs = []
for i in range(4):
    u, g = net.hip_udf(x, with_grad=True); torch.cuda.synchronize(); outs.append((u.clone().cpu(), g.clone().cpu()))
_lib.lib().emap_set_grad_mode(0)   # forward-mode tangents
uf, gf = net.hip_udf(x, with_grad=True); uf, gf = uf.cpu(), gf.cpu()
e_all = (outs[0][1] - gf).abs().max(dim=1).values
print(f"run 0 vs forward-mode kernel: max |dgrad| {float(e_all.max()):.3e}, rel to max {float(e_all.max() / gf.abs().max()):.2e}; udf max diff {float((outs[0][0] - uf).abs().max()):.2e}")
for i in range(1, 4):
    du = (outs[i][0] != outs[0][0]).reshape(-1); dg = (outs[i][1] != outs[0][1]).any(dim=1)
    d = (outs[i][1] - outs[0][1]).abs()
    idx = torch.nonzero(dg).reshape(-1)
    print(f"run {i} vs 0: udf differs at {int(du.sum())}, grad at {int(dg.sum())} points; max |dgrad| {float(d.max()):.3e} (|grad| max {float(outs[0][1].abs().max()):.2f})")
    if len(idx):
        ct = (idx % 64) // 16
        tile = idx // 64
        print("   by column tile (ct):", torch.bincount(ct, minlength=4).tolist(), " distinct tiles:", int(tile.unique().numel()), "of", P // 64,
              " tile%512 histogram head:", torch.bincount(tile % 8, minlength=8).tolist())
        # do whole 16-point groups differ together?
        grp = idx // 16
        cnt = torch.bincount(grp)
        print("   16-point groups hit:", int((cnt > 0).sum()), " fully (16/16):", int((cnt == 16).sum()))
        # size of the error relative to forward-mode result
        e0 = (outs[0][1] - gf).abs().max(dim=1).values; ei = (outs[i][1] - gf).abs().max(dim=1).values
        print(f"   error vs forward-mode kernel at differing points: run0 max {float(e0[idx].max()):.3e} median {float(e0[idx].median()):.3e}; run{i} max {float(ei[idx].max()):.3e} median {float(ei[idx].median()):.3e}; at equal points max {float(e0[~dg].max()):.3e}")
