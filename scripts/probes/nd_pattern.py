import os, sys, collections
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch, emap_amd
from conftest import net_state
kw, state = net_state("d8w256L10")
net = emap_amd.UDFNetwork(scale=1.0, precision="bf16x3", **kw); net.load_state_dict(state); net = net.cuda()
g = torch.Generator().manual_seed(7)
x = (torch.rand(32768, 3, generator=g) * 2 - 1).cuda()
with torch.no_grad():
    outs = [net.hip_udf(x, with_grad=True)[1].cpu().reshape(-1, 3) for _ in range(4)]
ref = outs[0]
for r in range(1, 4):
    d = outs[r] - ref
    idx = torch.nonzero(d.abs() > 0)
    pts = idx[:, 0]; ax = idx[:, 1]
    print("rep", r, "diff elems", len(idx), "axis hist", torch.bincount(ax, minlength=3).tolist(),
          "ct hist", torch.bincount((pts % 64) // 16, minlength=4).tolist(), "j hist", torch.bincount(pts % 16, minlength=16).tolist())
    mags = d[pts, ax].abs()
    print("   |delta| quantiles", [float(mags.quantile(q)) for q in (0.1, 0.5, 0.9, 1.0)], " tiles with diffs:", len(set((pts // 64).tolist())), "of 512;",
          "WG parity hist (tile<256 / >=256):", int((pts // 64 < 256).sum()), int((pts // 64 >= 256).sum()))
    # per tile: how many points differ
    per_tile = torch.bincount(pts // 64, minlength=512)
    print("   points-differing-per-affected-tile hist:", collections.Counter(per_tile[per_tile > 0].tolist()).most_common(8))
