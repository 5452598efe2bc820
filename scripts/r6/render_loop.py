#!/usr/bin/env python3
"""Round 6: render() of one launch shape in an eager loop - the command rocprofv3 --kernel-trace --stats is wrapped around for a per-kernel
breakdown of shapes other than the benchmark batch.  usage: render_loop.py N n_samples n_importance up_sample_steps [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import emap_amd
from emap_amd import synthetic
N, ns, ni, K = [int(v) for v in sys.argv[1:5]]
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 50
dev = torch.device("cuda:0")
kw = dict(d_in=3, d_out=1, d_hidden=256, n_layers=8, skip_in=(4,), multires=10, bias=0.5)
net = emap_amd.UDFNetwork(scale=1.0, precision="f16x3", **kw)
net.load_state_dict(synthetic.make_udf_state(seed=42, pert=0.02, **kw))
net = net.to(dev)
devn = emap_amd.SingleVarianceNetwork(0.3).to(dev)
bet = emap_amd.BetaNetwork(0.5, 0.3, 0.3, 5e-5, True, True, False).to(dev)
r = emap_amd.UDFRendererBlending(None, net, devn, bet, ns, ni, 0, K, 1.0, device=dev)
ro, rd, near, far, ds = [t.contiguous().to(dev) for t in synthetic.make_rays(N, seed=1)]
tr = synthetic.make_t_rand(N).to(dev)
with torch.no_grad():
    for _ in range(iters):
        r.render(ro, rd, near, far, ds, cos_anneal_ratio=1.0, flip_saturation=0.9, t_rand=tr)
torch.cuda.synchronize()
