import os, sys, json
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import emap_amd
from emap_amd import synthetic
dev = torch.device("cuda:0")
kw = dict(d_in=3, d_out=1, d_hidden=256, n_layers=8, skip_in=(4,), multires=10, bias=0.5)
net = emap_amd.UDFNetwork(scale=1.0, precision="f16x3", **kw)
net.load_state_dict(synthetic.make_udf_state(seed=42, pert=0.02, **kw))
net = net.to(dev)
L = emap_amd._lib.lib()
P = 65536
torch.manual_seed(3)
x = (torch.rand(P, 3) * 2.4 - 1.2).to(dev)
np.set_printoptions(precision=4, suppress=True, linewidth=250)
with torch.no_grad():
    L.emap_set_grad_mode(0)
    uf, gf = net.hip_udf(x, with_grad=True)
    L.emap_set_grad_mode(1)
    runs = [net.hip_udf(x, with_grad=True)[1].cpu().numpy() for _ in range(6)]
gf = gf.cpu().numpy()
shown = 0
for i in range(6):
    e = np.abs(runs[i] - gf).max(axis=1).reshape(-1, 64)
    for t in np.nonzero(e.max(axis=1) > 1e-3 * np.abs(gf).max())[0]:
        if shown >= 4: break
        shown += 1
        bad = np.nonzero(e[t] > 1e-3 * np.abs(gf).max())[0]
        print("run", i, "tile", t, "bad lanes", bad.tolist())
        j0 = bad[0]
        print(" ref  ", gf[t * 64 + j0: t * 64 + j0 + 4].round(4).tolist())
        print(" got  ", runs[i][t * 64 + j0: t * 64 + j0 + 4].round(4).tolist())
        r = runs[i][t * 64: t * 64 + 64] / gf[t * 64: t * 64 + 64]
        print(" ratio got/ref, comp 0, all 64 lanes:", r[:, 0].round(3).tolist())
        # does the wrong value equal the right value of some other point?
        for j in bad[:2]:
            d = np.abs(gf - runs[i][t * 64 + j]).max(axis=1)
            print(f"  lane {j}: nearest ref point {int(d.argmin())} (own {t*64+j}) dist {d.min():.4g}")
