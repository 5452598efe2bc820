#!/usr/bin/env python3
"""Soak: the cross-ray reduction inside the value + grad_x launch (agent-scope partials, one launch-wide counter) over thousands of renders, eager and from a
replayed graph, with other work keeping the caches busy in between - every scalar of every render bit-equal to the unfused render's (a stale partial read
across XCDs would show as a different eikonal sum)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
import emap_amd
from emap_amd import _lib, synthetic
from test_gpu_parity import mk, mk_renderer
DEV = "cuda:0"
L = _lib.lib()
KEYS = ("gradient_error", "gradient_error_near_surface", "sparse_error", "variance", "beta", "gamma", "edge", "depth")
total = bad = 0
for (N, ns, ni, K, reps) in [(512, 64, 64, 4, 3000), (1024, 64, 50, 5, 1500), (4096, 64, 64, 4, 400), (100, 64, 64, 4, 2000)]:
    net, _, _ = mk("d8w256L10", "f16x3")
    r = mk_renderer(net, ns, ni, K)
    batches = []
    for sd in range(4):
        ro, rd, near, far, ds = [v.to(DEV) for v in synthetic.make_rays(N, seed=20 + sd)]
        batches.append(((ro, rd, near, far, ds), synthetic.make_t_rand(N, seed=70 + sd).to(DEV)))
    refs = []
    L.emap_set_fused_composite(0)
    with torch.no_grad():
        for args, tr in batches:
            o = r.render(*args, cos_anneal_ratio=1.0, flip_saturation=0.9, t_rand=tr)
            refs.append({k: o[k].clone() for k in KEYS if k in o and isinstance(o[k], torch.Tensor)})
    L.emap_set_fused_composite(1)
    junk = torch.empty(64 << 20, device=DEV)          # 256 MB of traffic between renders now and then: evicts L2 / MALL lines
    with torch.no_grad():
        for i in range(reps):
            args, tr = batches[i % 4]
            o = r.render(*args, cos_anneal_ratio=1.0, flip_saturation=0.9, t_rand=tr)
            if i % 7 == 0:
                junk.add_(1.0)
            ok = all(torch.equal(o[k], refs[i % 4][k]) for k in refs[i % 4])
            total += 1
            bad += 0 if ok else 1
    # from a replayed graph
    args, tr = batches[0]
    step = r.capture(*args, cos_anneal_ratio=1.0, flip_saturation=0.9, t_rand=tr)
    for i in range(reps):
        o = step()
        ok = all(torch.equal(o[k], refs[0][k]) for k in refs[0])
        total += 1
        bad += 0 if ok else 1
    torch.cuda.synchronize()
    r.check_errors()
    print(f"{N} rays x {ns}+{ni}/{K}: {2 * reps} renders, mismatching so far {bad}", flush=True)
print(f"soak: {total} renders, {bad} with a scalar or per-ray output different from the unfused render's")
sys.exit(1 if bad else 0)
