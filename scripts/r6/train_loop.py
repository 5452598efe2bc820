#!/usr/bin/env python3
"""Round 6: the native training step (emap_amd.parallel.Trainer) of one launch shape in an eager loop - the command rocprofv3 --kernel-trace --stats is
wrapped around for a per-kernel breakdown.  usage: train_loop.py N n_samples n_importance up_sample_steps [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import emap_amd
from emap_amd import synthetic
from emap_amd.parallel import Trainer
N, ns, ni, K = [int(v) for v in sys.argv[1:5]]
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 40
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
te = synthetic.make_true_edge(N, seed=11).to(dev)
trainer = Trainer(r, lr_geo=1e-4, lr=5e-4, edge_weight=1.0, igr_weight=0.1, igr_ns_weight=0.0)
batch = {"rays_o": ro, "rays_d": rd, "near": near, "far": far, "depth_scale": ds, "cos_anneal_ratio": 1.0, "flip_saturation": 0.9, "t_rand": tr}
for _ in range(iters):
    trainer.step(batch, te, n_rays_global=N)
torch.cuda.synchronize()
r.check_errors()
