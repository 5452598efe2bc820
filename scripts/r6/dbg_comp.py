import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import emap_amd
from emap_amd import synthetic, _lib
from test_gpu_parity import mk, mk_renderer
DEV = "cuda:0"
L = _lib.lib()
net, _, _ = mk("d8w256L10", "f16x3")
for (N, ns, ni, K) in [(90, 64, 192, 4), (200, 64, 192, 4), (90, 64, 128, 4), (512, 64, 64, 4)]:
    r = mk_renderer(net, ns, ni, K)
    ro, rd, near, far, ds = [v.to(DEV) for v in synthetic.make_rays(N, seed=9)]
    tr = synthetic.make_t_rand(N).to(DEV)
    kw = dict(cos_anneal_ratio=0.7, flip_saturation=0.9, t_rand=tr)
    outs = {}
    for rep in range(3):
        for fused in (1, 0):
            L.emap_set_fused_composite(fused)
            with torch.no_grad():
                o = r.render(ro, rd, near, far, ds, **kw)
            torch.cuda.synchronize()
            outs[(fused, rep)] = {k: v.clone() for k, v in o.items() if isinstance(v, torch.Tensor)}
    L.emap_set_fused_composite(1)
    print("shape", N, ns + ni)
    for k in outs[(0, 0)]:
        a, b = outs[(1, 0)][k], outs[(0, 0)][k]
        d = (a != b)
        dd = (outs[(0, 0)][k] != outs[(0, 1)][k])
        d1 = (outs[(1, 0)][k] != outs[(1, 1)][k])
        if d.any() or dd.any() or d1.any():
            idx = d.nonzero()[:5].tolist()
            print("  ", k, "fused!=sep:", int(d.sum()), "of", d.numel(), "sep run-to-run:", int(dd.sum()), "fused run-to-run:", int(d1.sum()), idx,
                  [float(a[tuple(i)]) for i in idx], [float(b[tuple(i)]) for i in idx])
