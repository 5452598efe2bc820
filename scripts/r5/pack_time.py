"""Time of the weight re-pack (rowscale_kernel + pack_all_kernel) by section: run under rocprofv3 --kernel-trace --stats with library variants
built with -DEMAP_PACK_SECTIONS=<mask> (EMAP_VARIANT_UNITS=udf_mlp scripts/build_variant.sh pk<mask> -DEMAP_PACK_SECTIONS=<mask>)."""
import os
import sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import emap_amd
from emap_amd import synthetic

dev = "cuda:0"
kw = dict(d_in=3, d_out=1, d_hidden=256, n_layers=8, skip_in=(4,), multires=10, bias=0.5, scale=1.0, geometric_init=True, weight_norm=True, udf_type="abs")
net = emap_amd.UDFNetwork(precision="f16x3", **kw).to(dev)
for _ in range(60):
    net.invalidate_packed()
    net.packed("f16x3")
torch.cuda.synchronize()
print("ok")
