#!/usr/bin/env python3
"""Round 6: time of ONE value + grad_x launch per point count with forward-mode tangents (emap_set_grad_mode(0): udf_mlp_fs2_kernel<.., GRAD>) and with the
reverse sweep (1: udf_mlp_rev32_kernel) - where the launcher's crossover (udf_mlp.hip:mlp_variant, 10 240 points in the split modes) should sit."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import emap_amd
from emap_amd import synthetic, _lib
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
Ps = [int(v) for v in sys.argv[2:]] or [1024, 2048, 3072, 4096, 5120, 6144, 7168, 8192, 9216, 10240, 12288, 16384, 20480, 24576, 32768]
dev = torch.device("cuda:0")
kw = dict(d_in=3, d_out=1, d_hidden=256, n_layers=8, skip_in=(4,), multires=10, bias=0.5)
net = emap_amd.UDFNetwork(scale=1.0, precision=prec, **kw)
net.load_state_dict(synthetic.make_udf_state(seed=42, pert=0.02, **kw))
net = net.to(dev)
L = _lib.lib()
x = torch.rand(max(Ps), 3, device=dev) * 2 - 1
out = {}
with torch.no_grad():
    for P in Ps:
        xs = x[:P].contiguous()
        for rnd in range(2):
            for mode in (0, 1):
                L.emap_set_grad_mode(mode)
                for _ in range(5):
                    net.hip_udf(xs, with_grad=True)
                torch.cuda.synchronize()
                ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(40)]
                for s, e in ev:
                    s.record(); net.hip_udf(xs, with_grad=True); e.record()
                torch.cuda.synchronize()
                ts = sorted(s.elapsed_time(e) for s, e in ev)
                out.setdefault(P, {}).setdefault(("fwd", "rev")[mode], []).append(round(ts[len(ts) // 2] * 1e3, 1))
L.emap_set_grad_mode(-1)
print("# points    forward-mode us      reverse sweep us")
for P, v in out.items():
    print(f"{P:8d}    {'/'.join(str(t) for t in v['fwd']):>16s}    {'/'.join(str(t) for t in v['rev']):>16s}   {'rev' if min(v['rev']) < min(v['fwd']) else 'fwd'}")
