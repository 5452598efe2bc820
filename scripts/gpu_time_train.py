#!/usr/bin/env python3
"""Timing of one training step's pieces on one GPU (512 rays x 128 samples by default): forward render, forward + HIP
backward through autograd (the drop-in path), and the direct emap_render_bwd call.  Prints one JSON line."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import emap_amd  # noqa: E402
from emap_amd import synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=512)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--precision", default="f16x3")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    kw = dict(d_in=3, d_out=1, d_hidden=256, n_layers=8, skip_in=(4,), multires=10, bias=0.5)
    net = emap_amd.UDFNetwork(scale=1.0, precision=a.precision, **kw)
    net.load_state_dict(synthetic.make_udf_state(seed=42, pert=0.02, **kw))
    net = net.to(dev)
    devn = emap_amd.SingleVarianceNetwork(0.3).to(dev)
    bet = emap_amd.BetaNetwork(0.5, 0.3, 0.3, 5e-5, True, True, False).to(dev)
    r = emap_amd.UDFRendererBlending(None, net, devn, bet, 64, 64, 0, 4, 1.0, device=dev)
    N = a.rays
    ro, rd, near, far, ds = [v.to(dev) for v in synthetic.make_rays(N, seed=1)]
    te = synthetic.make_true_edge(N, seed=11).to(dev)
    tr = synthetic.make_t_rand(N, seed=7).to(dev)
    kwr = dict(cos_anneal_ratio=1.0, flip_saturation=0.9, t_rand=tr)

    def timed(fn, iters):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for s, e in ev:
            s.record(); fn(); e.record()
        torch.cuda.synchronize()
        ts = sorted(s.elapsed_time(e) for s, e in ev)
        return ts[len(ts) // 2]

    def fwd():
        with torch.no_grad():
            return r.render(ro, rd, near, far, ds, **kwr)

    def step_autograd():
        for p in list(net.parameters()) + [devn.variance, bet.beta, bet.gamma]:
            p.grad = None
        out = r.render(ro, rd, near, far, ds, **kwr)
        loss = ((out["edge"] - te) ** 2).mean() + 0.1 * out["gradient_error"]
        loss.backward()

    call = r._prepare(ro, rd, near, far, ds, 1.0, -1, None, 0.9, tr)
    v = r._render_hip(call)
    flat = torch.empty(r._layout().numel, device=dev)
    d_edge = 2 * (v["edge"] - te.view(-1)) / N
    wge = torch.tensor([0.1], device=dev)

    def bwd_direct():
        r.backward_into(call, v, d_edge, None, wge, None, flat=flat)

    res = {"rays": N, "samples": 128, "precision": a.precision,
           "fwd_ms": timed(fwd, a.iters), "step_autograd_ms": timed(step_autograd, a.iters), "bwd_direct_ms": timed(bwd_direct, a.iters)}
    res["train_ray_samples_per_s"] = N * 128 / (res["step_autograd_ms"] * 1e-3)
    r.check_errors()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
