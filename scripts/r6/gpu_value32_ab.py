#!/usr/bin/env python3
"""Round 6: the wide value launches as the forward sweep of the 32x32 kernel (emap_set_value_tile_mode(1), udf_mlp_rev32.inc VAL) against
udf_mlp_fs2_kernel (0): udf of both forms against the fp64 oracle and against each other, time of one value launch over a range of point
counts, and the 512 / 1024 / 4096-ray render from a graph in both forms (same process, same box, interleaved).
usage: python scripts/r6/gpu_value32_ab.py [precision]"""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import emap_amd
from emap_amd import synthetic, _lib
from oracle import emap_oracle as O

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
L = _lib.lib()
dev = torch.device("cuda:0")
kw = dict(d_in=3, d_out=1, d_hidden=256, n_layers=8, skip_in=(4,), multires=10, bias=0.5)
state = synthetic.make_udf_state(seed=42, pert=0.02, **kw)
net = emap_amd.UDFNetwork(scale=1.0, precision=prec, **kw)
net.load_state_dict(state)
net = net.to(dev)
res = {"prec": prec}

# ---- values ----
P = 40000 + 37
x = torch.rand(P, 3) * 2 - 1
cfg = O.UDFConfig(d_hidden=256, n_layers=8, multires=10)
ref = O.udf_value_and_grad({k: v.double() for k, v in state.items()}, cfg, x[:4096].double())[0].float()
with torch.no_grad():
    vals = {}
    for mode in (0, 1):
        L.emap_set_value_tile_mode(mode)
        vals[mode] = net.hip_udf(x.to(dev))[0].cpu().clone()
    L.emap_set_value_tile_mode(0)
d = (vals[0] - vals[1]).abs()
res["value_forms_max_abs_diff"] = float(d.max())
res["value_forms_max_norm_diff"] = float(d.max() / vals[0].abs().max())
if ref is not None:
    for mode in (0, 1):
        res[f"mode{mode}_vs_fp64_oracle_maxnorm"] = float((vals[mode][:4096].reshape(-1) - ref.reshape(-1)).abs().max() / ref.abs().max())

# ---- one value launch ----
Ps = [32768, 40960, 65536, 131072, 262144]
xg = torch.rand(max(Ps), 3, device=dev) * 2 - 1
tv = {}
with torch.no_grad():
    for Pn in Ps:
        xs = xg[:Pn].contiguous()
        for mode in (0, 1, 0, 1):
            L.emap_set_value_tile_mode(mode)
            for _ in range(5):
                net.hip_udf(xs)
            torch.cuda.synchronize()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(60)]
            for s, e in ev:
                s.record(); net.hip_udf(xs); e.record()
            torch.cuda.synchronize()
            ts = sorted(s.elapsed_time(e) for s, e in ev)
            tv.setdefault(Pn, {}).setdefault(mode, []).append(round(ts[len(ts) // 2] * 1e3, 1))
res["value_launch_us"] = tv

# ---- render from a graph ----
devn = emap_amd.SingleVarianceNetwork(0.3).to(dev)
bet = emap_amd.BetaNetwork(0.5, 0.3, 0.3, 5e-5, True, True, False).to(dev)
tr = {}
for N in (512, 1024, 4096):
    ro, rd, near, far, ds = [t.to(dev) for t in synthetic.make_rays(N, seed=1)]
    for mode in (0, 1, 0, 1):
        L.emap_set_value_tile_mode(mode)
        r = emap_amd.UDFRendererBlending(None, net, devn, bet, 64, 64, 0, 4, 0.0, device=dev)
        with torch.no_grad():
            for _ in range(3):
                out = r.render(ro, rd, near, far, ds, cos_anneal_ratio=1.0, perturb_overwrite=0, flip_saturation=0.9)
            torch.cuda.synchronize()
            g = r.capture(ro, rd, near, far, ds, cos_anneal_ratio=1.0, flip_saturation=0.9).graph
            reps = 200 if N <= 1024 else 60
            for _ in range(20):
                g.replay()
            torch.cuda.synchronize()
            s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(reps):
                g.replay()
            e.record(); torch.cuda.synchronize()
            tr.setdefault(N, {}).setdefault(mode, []).append(round(s.elapsed_time(e) / reps, 4))
L.emap_set_value_tile_mode(0)
res["render_ms"] = tr
print(json.dumps(res))
