#!/usr/bin/env python3
"""Dump the packed weight buffer of the synthetic d8 w256 network for a precision mode (A/B of packers: EMAP_HIP_LIB selects the library).
usage: python scripts/probes/dump_packed.py <precision> <out.npy>"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import emap_amd
from emap_amd import synthetic
kw = dict(d_in=3, d_out=1, d_hidden=256, n_layers=8, skip_in=(4,), multires=10, bias=0.5)
net = emap_amd.UDFNetwork(scale=1.0, precision=sys.argv[1], **kw)
net.load_state_dict(synthetic.make_udf_state(seed=42, pert=0.02, **kw))
net = net.to("cuda:0")
buf = net.packed(sys.argv[1])
torch.cuda.synchronize()
np.save(sys.argv[2], buf.cpu().numpy())
print(sys.argv[1], buf.numel(), "bytes")
