#!/usr/bin/env python3
"""Measurement of the full-image path (SURVEY par. 8 f4; emap_amd/validation.py): one H x W image (default 400 x 400 rays,
128 samples each) rendered (a) with the reference's schedule - batch_size=512 rays per render() and three device->host
copies per chunk (runner_udf.py:303-388) - and (b) with render_image (8192-ray launches, one copy at the end).  One JSON line.
    python scripts/bench_image.py [--H 400 --W 400]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import emap_amd  # noqa: E402
from emap_amd import synthetic  # noqa: E402
from emap_amd.validation import render_image  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--H", type=int, default=400)
    ap.add_argument("--W", type=int, default=400)
    ap.add_argument("--launch-rays", type=int, default=0, help="rays per emap_render_fwd call (0: the whole image in one call, the default of render_image)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    kw = dict(d_in=3, d_out=1, d_hidden=256, n_layers=8, skip_in=(4,), multires=10, bias=0.5)
    net = emap_amd.UDFNetwork(scale=1.0, precision="f16x3", **kw)
    net.load_state_dict(synthetic.make_udf_state(seed=42, pert=0.02, **kw))
    net = net.to(dev)
    r = emap_amd.UDFRendererBlending(None, net, emap_amd.SingleVarianceNetwork(0.3).to(dev),
                                     emap_amd.BetaNetwork(0.5, 0.3, 0.3, 5e-5, True, True, False).to(dev), 64, 64, 0, 4, 1.0, device=dev)
    n = a.H * a.W
    ro, rd, near, far, ds = synthetic.make_rays(n, seed=3)
    ro, rd, ds = ro.to(dev), rd.to(dev), ds.to(dev)
    nf, ff = float(near.reshape(-1)[0]), float(far.reshape(-1)[0])
    S = 128

    def reference_schedule():
        e, d, nm = [], [], []
        for h in range(0, n, 512):
            with torch.no_grad():
                o = r.render(ro[h:h + 512], rd[h:h + 512], nf, ff, depth_scale=ds[h:h + 512], cos_anneal_ratio=1.0)
            e.append(o["edge"].detach().cpu().numpy())
            d.append(o["depth"].detach().cpu().numpy())
            nm.append((o["gradients_flip"] * o["weights"][:, :S, None]).sum(dim=1).detach().cpu().numpy())
        return e, d, nm

    def fused():
        return render_image(r, ro, rd, nf, ff, ds, batch_size=512, cos_anneal_ratio=1.0, launch_rays=a.launch_rays or None)

    out = {}
    for name, fn in (("reference_schedule", reference_schedule), ("render_image", fused)):
        fn()
        torch.cuda.synchronize()
        t0 = time.time()
        fn()
        torch.cuda.synchronize()
        out[name] = time.time() - t0
    print(json.dumps({"metric": "ray-samples/sec (full image, edge+depth+normals on the host)", "value": n * S / out["render_image"],
                      "unit": "ray-samples/s", "config": {"workload": f"{a.H}x{a.W} rays x {S} samples, f16x3, " + (f"launches of {a.launch_rays} rays" if a.launch_rays else "ONE emap_render_fwd call for the image")},
                      "seconds": out["render_image"], "reference_schedule_seconds": out["reference_schedule"],
                      "speedup_vs_reference_schedule": out["reference_schedule"] / out["render_image"], "data": "synthetic"}))


if __name__ == "__main__":
    main()
