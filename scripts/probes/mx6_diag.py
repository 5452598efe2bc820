"""Where does the reverse sweep (current EMAP_HIP_LIB) differ from the forward-mode kernel?  Per 64-point tile and per run."""
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
P = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
torch.manual_seed(3)
x = (torch.rand(P, 3) * 2.4 - 1.2).to(dev)
with torch.no_grad():
    L.emap_set_grad_mode(0)
    uf, gf = net.hip_udf(x, with_grad=True)
    L.emap_set_grad_mode(1)
    runs = [net.hip_udf(x, with_grad=True) for _ in range(4)]
torch.cuda.synchronize()
print(json.dumps({"err_word_after_runs": hex(int(net.err_word(dev).item()))}))
gmax = float(gf.abs().max())
for i, (u, g) in enumerate(runs):
    e = ((g - gf).abs().max(dim=1).values / gmax).cpu().numpy()
    et = e.reshape(-1, 64).max(axis=1)
    bad = np.nonzero(et > 2e-4)[0]
    ec = e.reshape(-1, 2, 32).max(axis=2)      # per column tile of 32
    print(json.dumps({"run": i, "max_rel": float(e.max()), "median_tile_err": float(np.median(et)), "bad_tiles": int(bad.size), "n_tiles": int(et.size),
                      "first_bad": bad[:16].tolist(), "bad_mod_512": sorted(set((bad % 512).tolist()))[:16],
                      "bad_col0": int((ec[:, 0] > 2e-4).sum()), "bad_col1": int((ec[:, 1] > 2e-4).sum()),
                      "udf_equal_fwd": float((u - uf).abs().max() / uf.abs().max()),
                      "same_as_run0": bool(torch.equal(g, runs[0][1]))}))
    if i > 0:      # where do two runs of the SAME kernel differ?  (independent of the precision of the variant)
        d = ((g - runs[0][1]).abs().max(dim=1).values / gmax).cpu().numpy().reshape(-1, 64)
        dt = np.nonzero(d.max(axis=1) > 0)[0]
        print(json.dumps({"run": i, "tiles_differing_from_run0": int(dt.size), "lanes_of_first": [np.nonzero(d[t] > 0)[0].tolist() for t in dt[:3]],
                          "max_diff_first": [float(d[t].max()) for t in dt[:3]]}))
