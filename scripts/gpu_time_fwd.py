#!/usr/bin/env python3
"""A/B timing helper: median ms of the forward render (512 x 128) and of single MLP launches (value+grad 65536 points, value 32768 / 8192)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import emap_amd
from emap_amd import synthetic

def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
    dev = torch.device("cuda:0")
    kw = dict(d_in=3, d_out=1, d_hidden=256, n_layers=8, skip_in=(4,), multires=10, bias=0.5)
    net = emap_amd.UDFNetwork(scale=1.0, precision=prec, **kw)
    net.load_state_dict(synthetic.make_udf_state(seed=42, pert=0.02, **kw))
    net = net.to(dev)
    devn = emap_amd.SingleVarianceNetwork(0.3).to(dev)
    bet = emap_amd.BetaNetwork(0.5, 0.3, 0.3, 5e-5, True, True, False).to(dev)
    r = emap_amd.UDFRendererBlending(None, net, devn, bet, 64, 64, 0, 4, 1.0, device=dev)
    N = 512
    ro, rd, near, far, ds = [v.to(dev) for v in synthetic.make_rays(N, seed=1)]
    tr = synthetic.make_t_rand(N, seed=7).to(dev)
    x = torch.rand(65536, 3, device=dev) * 2 - 1

    def timed(fn, iters=40):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(iters)]
        for s, e in ev:
            s.record(); fn(); e.record()
        torch.cuda.synchronize()
        ts = sorted(s.elapsed_time(e) for s, e in ev)
        return ts[len(ts) // 2]

    with torch.no_grad():
        res = {"prec": prec, "env": {k: v for k, v in os.environ.items() if k.startswith("EMAP_")},
               "render_ms": timed(lambda: r.render(ro, rd, near, far, ds, cos_anneal_ratio=1.0, flip_saturation=0.9, t_rand=tr)),
               "fwd_grad_65536_ms": timed(lambda: net.hip_udf(x, with_grad=True)),
               "fwd_32768_ms": timed(lambda: net.hip_udf(x[:32768])),
               "fwd_8192_ms": timed(lambda: net.hip_udf(x[:8192]))}
    print(json.dumps(res))


main()
