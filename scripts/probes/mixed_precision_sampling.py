"""CPU emulation: how much do the FINAL render outputs move if only the sampling passes of importance_sample (coarse + 3 narrow value passes,
udf_renderer_blending.py:802-841, no_grad) run in cheaper MFMA arithmetic, the final value+gradient pass staying f16x3?  The sampling passes only
place samples; the question is the sensitivity of the quadrature to the sample positions.  Arithmetic emulated as in precision_emulation.py.

    python scripts/probes/mixed_precision_sampling.py [--rays 128]
"""
import argparse, json, os, sys
import numpy as np
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts", "probes"))
from oracle import emap_oracle as O          # noqa: E402  (a probe, not product code)
from emap_amd import synthetic                # noqa: E402
import precision_emulation as PE              # noqa: E402


def value_emulated(state, cfg, x, passes):
    Ws, bs = O._weights(state, cfg, torch.float32)
    xs = x * cfg.scale
    pe = O.positional_encoding(xs, cfg.multires)
    a = pe
    s2 = float(1.0 / np.sqrt(2))
    for l in range(cfg.n_lin):
        W = Ws[l]
        if l in cfg.skip_in:
            a = torch.cat([a, pe], 1)
            W = W * s2
        z = PE.gemm(W, a, passes) + bs[l]
        a = F.softplus(z, beta=100) if l < cfg.n_lin - 1 else z
    return a[:, :1].abs() / cfg.scale


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=128)
    args = ap.parse_args()
    kw = dict(d_in=3, d_out=1, d_hidden=256, n_layers=8, skip_in=(4,), multires=10, bias=0.5)
    state = synthetic.make_udf_state(seed=42, pert=0.02, **kw)
    cfg = O.UDFConfig(d_hidden=256, n_layers=8, multires=10)
    rcfg = O.RenderConfig(n_samples=64, n_importance=64, up_sample_steps=4)
    rays = synthetic.make_rays(args.rays, seed=1)
    ro, rd, near, far, ds = rays
    var, bp, gp = torch.tensor(0.3), torch.tensor(0.5), torch.tensor(0.3)
    kwargs = dict(cos_anneal_ratio=1.0, flip_saturation=0.9)
    orig = O.udf_value
    ref = O.render(state, cfg, rcfg, ro, rd, near, far, ds, var, bp, gp, **kwargs)
    rel = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max())
    for passes in ("hh+hl+lh", "hh+hl+l8h", "hh+lh", "hh"):
        calls = {"n": 0}

        def patched(st, c, x, _p=passes):
            calls["n"] += 1
            return value_emulated(st, c, x, _p)
        O.udf_value = patched       # importance_sample / cat_z_vals look the name up in the module at call time
        try:
            out = O.render(state, cfg, rcfg, ro, rd, near, far, ds, var, bp, gp, **kwargs)
        finally:
            O.udf_value = orig
        print(json.dumps({"sampling_passes": passes, "udf_value_calls": calls["n"],
                          **{k: rel(out[k], ref[k]) for k in ("z_vals", "edge", "depth", "normals", "weights", "gradient_error")}}))


if __name__ == "__main__":
    main()
