import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import load_golden
from test_gpu_backward import _render_bwd_on_reference_samples
from test_gpu_parity import t
for ci in (0, 3):
    g = load_golden(f"g6_training_{ci}")
    ref = {k[5:]: t(g[k]) for k in g if k.startswith("grad.lin")}
    gmax = max(float(v.abs().max()) for v in ref.values())
    res = {}
    for prec in ("f16x3", "f16x3e"):
        loss, got, extra = _render_bwd_on_reference_samples(g, prec)
        res[prec] = {k: float((got[k].double() - ref[k].double()).abs().max() / ref[k].abs().max()) for k in ref}
    print("case", ci)
    for k in ref:
        print(f"  {k:50s} max {float(ref[k].abs().max()):10.3e}  f16x3 {res['f16x3'][k]:.2e}  f16x3e {res['f16x3e'][k]:.2e}")
