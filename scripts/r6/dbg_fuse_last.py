#!/usr/bin/env python3
"""rev32 (LASTH) against the forward-mode tangent kernel at several launch sizes; where the bad points sit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
import emap_amd
from emap_amd import _lib
from conftest import net_state
dev = torch.device("cuda:0")
for name in ["d8w256L10"]:
    for prec in sys.argv[1:] or ["f16x3", "f16x3e", "bf16x3", "f16"]:
        kw, state = net_state(name)
        net = emap_amd.UDFNetwork(scale=1.0, precision=prec, **kw); net.load_state_dict(state); net = net.to(dev)
        g = torch.Generator().manual_seed(5)
        xall = (torch.rand(524288, 3, generator=g) * 2 - 1).to(dev)
        for P in [8193, 65536, 131072, 524288]:
            x = xall[:P].contiguous()
            with torch.no_grad():
                _lib.lib().emap_set_grad_mode(1)
                u, gr = net.hip_udf(x, with_grad=True)
                u2, gr2 = net.hip_udf(x, with_grad=True)
                Q = min(P, 65536)
                old = _lib.lib().emap_set_grad_mode(0)
                uf, gf = net.hip_udf(x[:Q].contiguous(), with_grad=True)
                _lib.lib().emap_set_grad_mode(-1)
            torch.cuda.synchronize()
            du = (u[:Q] - uf).abs().flatten(); dg = (gr[:Q] - gf).abs().amax(dim=1)
            print(prec, P, "udf rel %.2e grad rel %.2e  rerun equal %s %s" % (float(du.max() / uf.abs().max()), float(dg.max() / gf.abs().max()),
                  torch.equal(u, u2), torch.equal(gr, gr2)), flush=True)
            bad = (dg > 1e-3 * gf.abs().max()).nonzero().flatten().cpu()
            if len(bad):
                print("   bad points:", len(bad), "first", bad[:12].tolist(), "tiles", sorted(set((bad // 64).tolist()))[:12], "n_tiles_bad", len(set((bad // 64).tolist())),
                      "col-in-tile hist", torch.bincount(bad % 64, minlength=64).tolist())
