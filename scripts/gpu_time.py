#!/usr/bin/env python3
"""Quick timing of the MLP kernels and the full render (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import emap_amd
from emap_amd import synthetic
from conftest import net_state
dev = torch.device("cuda:0")
precs = sys.argv[1].split(",") if len(sys.argv) > 1 else ["bf16", "bf16x3"]

def timeit(fn, n=20, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for prec in precs:
    kw, state = net_state("d8w256L10")
    net = emap_amd.UDFNetwork(scale=1.0, precision=prec, **kw); net.load_state_dict(state); net = net.to(dev)
    devn = emap_amd.SingleVarianceNetwork(0.3).to(dev); bet = emap_amd.BetaNetwork(0.5, 0.3, 0.3, 5e-5, True, True, False).to(dev)
    with torch.no_grad():
        for P in (8192, 32768, 65536, 262144):
            x = torch.rand(P, 3, device=dev) * 2 - 1
            for wg in (False, True):
                us = timeit(lambda: net.hip_udf(x, with_grad=wg))
                F = 918016 * (2 if wg else 1)
                print(f"mlp {prec:7s} grad={int(wg)} P={P:7d}: {us:8.1f} us  {P*F/us/1e6:7.1f} TF algorithmic", flush=True)
        for N in (512, 4096):
            r = emap_amd.UDFRendererBlending(None, net, devn, bet, 64, 64, 0, 4, 1.0, device=dev)
            ro, rd, near, far, ds = [v.to(dev) for v in synthetic.make_rays(N, seed=1)]
            tr = synthetic.make_t_rand(N).to(dev)
            us = timeit(lambda: r.render(ro, rd, near, far, ds, cos_anneal_ratio=1.0, flip_saturation=0.9, t_rand=tr))
            print(f"render {prec} N={N} S=128: {us:.1f} us -> {N*128/us*1e6:.3e} ray-samples/s ({N*128*2639296/us/1e6:.1f} TF alg)", flush=True)
