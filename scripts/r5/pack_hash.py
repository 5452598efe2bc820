"""sha256 of the packed weight buffer of seeded networks in every precision mode (to compare library variants: EMAP_HIP_LIB=... python scripts/r5/pack_hash.py)."""
import hashlib
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import emap_amd

dev = "cuda:0"
nets = {"d8w256L10": dict(d_in=3, d_out=1, d_hidden=256, n_layers=8, skip_in=(4,), multires=10),
        "d4w128L10": dict(d_in=3, d_out=1, d_hidden=128, n_layers=4, skip_in=(2,), multires=10),
        "d8w256L6": dict(d_in=3, d_out=1, d_hidden=256, n_layers=8, skip_in=(4,), multires=6),
        "d8w256L0": dict(d_in=3, d_out=1, d_hidden=256, n_layers=8, skip_in=(4,), multires=0)}
for name, kw in nets.items():
    for prec in ("f16x3", "f16x3m", "f16x3e", "bf16x3", "f16", "bf16"):
        torch.manual_seed(3)
        net = emap_amd.UDFNetwork(precision=prec, bias=0.5, scale=1.0, geometric_init=True, weight_norm=True, udf_type="abs", **kw).to(dev)
        with torch.no_grad():
            for p in net.parameters():
                p.add_(0.01 * torch.randn_like(p))
        try:
            buf = net.packed(prec)
        except RuntimeError as e:          # e.g. f16x3m needs d_hidden = 256
            print(name, prec, "unsupported:", str(e)[-60:])
            continue
        torch.cuda.synchronize()
        print(name, prec, buf.numel(), hashlib.sha256(buf.cpu().numpy().tobytes()).hexdigest()[:16])
