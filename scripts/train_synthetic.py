#!/usr/bin/env python3
"""BASELINE config C5 in miniature: the reference's training loop (src/runner/runner_udf.py:79-168 with the schedules of
runner_base.py:128-180, time axis compressed) on a multi-view consistent synthetic wire frame, entirely on the HIP path
(device ray sampler -> render forward -> HIP backward -> fused Adam).  Prints one JSON line: loss / PSNR trajectory, PSNR of a
held-out view, the learned UDF on and off the wire frame, wall time.  One GPU; `--steps 4000` takes ~10 s."""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import emap_amd  # noqa: E402
from emap_amd import synthetic  # noqa: E402
from emap_amd.parallel import Trainer  # noqa: E402
from emap_amd.validation import render_image  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--rays", type=int, default=512)
    ap.add_argument("--precision", default="f16x3")
    ap.add_argument("--views", type=int, default=16)
    ap.add_argument("--res", type=int, default=200)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    kw = dict(d_in=3, d_out=1, d_hidden=256, n_layers=8, skip_in=(4,), multires=10, bias=0.5)
    net = emap_amd.UDFNetwork(scale=1.0, precision=a.precision, **kw).to(dev)          # the reference's geometric initialisation
    devn = emap_amd.SingleVarianceNetwork(0.3).to(dev)
    bet = emap_amd.BetaNetwork(0.5, 0.3, 0.3, 5e-5, True, True, False).to(dev)
    r = emap_amd.UDFRendererBlending(None, net, devn, bet, n_samples=64, n_importance=64, n_outside=0, up_sample_steps=4,
                                     perturb=1.0, device=dev)
    meta, edges = synthetic.make_wireframe_scene(n_images=a.views, H=a.res, W=a.res)
    sampler = emap_amd.DeviceRaySampler.from_meta(meta, edges, device=dev, seed=5)
    held_out = 0
    sampler.set_image_perm([i for i in range(a.views) if i != held_out])
    near, far = float(meta["scene_box"]["near"]), float(meta["scene_box"]["far"])
    # ABC.conf: learning_rate 5e-4, learning_rate_geo 1e-4, alpha 0.05, end_iter 50000, warm_up_end 1000, anneal_end 10000,
    # igr_weight 0.1, igr_ns_weight 0, edge_weight 1 - the iteration axis scaled by steps / 50000
    lr, lr_geo, alpha, end_iter = 5e-4, 1e-4, 0.05, a.steps
    warm_up_end, anneal_end, flip_start = max(1, end_iter // 50), max(1, end_iter // 5), end_iter // 5
    t = Trainer(r, lr_geo=lr_geo, lr=lr, edge_weight=1.0, igr_weight=0.1, igr_ns_weight=0.0)

    def lr_factor(it):                                    # runner_base.py:128-141
        if it < warm_up_end:
            return it / warm_up_end
        prog = (it - warm_up_end) / (end_iter - warm_up_end)
        return (math.cos(math.pi * prog) + 1.0) * 0.5 * (1 - alpha) + alpha

    def lr_geo_factor(it):                                # :143-160 (fix_geo_end = 0)
        if it < warm_up_end * 2:
            return it / (warm_up_end * 2)
        if it < end_iter * 0.5:
            return 1.0
        prog = (it - end_iter * 0.5) / (end_iter * 0.5)
        return (math.cos(math.pi * prog) + 1.0) * 0.5 * (1 - alpha) + alpha

    def view_psnr(idx):
        s = sampler.gen_random_rays_patches_at(idx, a.res * a.res, pixels=torch.stack(torch.meshgrid(
            torch.arange(a.res, device=dev), torch.arange(a.res, device=dev), indexing="xy"), -1).reshape(-1, 2))
        res = render_image(r, s["rays"]["rays_o"], s["rays"]["rays_v"], near, far, s["depth_scale"], batch_size=8192,
                           cos_anneal_ratio=1.0, to_numpy=False)
        mse = float(((res["edge"].reshape(-1) - s["rays"]["edge"].reshape(-1)) ** 2).mean())
        return 10.0 * math.log10(1.0 / max(mse, 1e-12)), mse

    def udf_on_off():
        segs = torch.tensor(synthetic.wireframe_segments(), dtype=torch.float32, device=dev)
        tt = torch.linspace(0.05, 0.95, 64, device=dev).view(1, -1, 1)
        on = (segs[:, :1] * (1 - tt) + segs[:, 1:] * tt).reshape(-1, 3)
        g = torch.Generator(device="cpu").manual_seed(1)
        off = (torch.rand(4096, 3, generator=g) * 1.6 - 0.8).to(dev)
        d = torch.cdist(off, on).min(1).values
        off = off[d > 0.15]
        with torch.no_grad():
            return float(net.udf(on)[0].mean()), float(net.udf(off)[0].mean())

    log = []
    psnr0 = view_psnr(held_out)
    u0 = udf_on_off()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    acc = torch.zeros(2, device=dev)
    every = max(1, a.steps // 20)
    for it in range(a.steps):
        t.optimizer.param_groups[0]["lr"] = lr_geo * lr_geo_factor(it)
        for g_ in t.optimizer.param_groups[1:]:
            g_["lr"] = lr * lr_factor(it)
        car = min(1.0, it / anneal_end)                                        # runner_base.py:162-166
        fs = 0.0 if it < flip_start else (0.9 if it < end_iter * 0.5 else 1.0)   # :171-180
        smp = sampler.gen_random_rays_patches_at(None, a.rays, importance_sample=True)
        batch = {"rays_o": smp["rays"]["rays_o"], "rays_d": smp["rays"]["rays_v"], "near": near, "far": far,
                 "depth_scale": smp["depth_scale"], "cos_anneal_ratio": car, "flip_saturation": fs,
                 "t_rand": torch.rand(a.rays, 1, device=dev) - 0.5}
        acc += t.step(batch, smp["rays"]["edge"])
        if (it + 1) % every == 0:
            m = (acc / every).tolist()
            acc.zero_()
            log.append({"step": it + 1, "loss": m[0], "edge_loss": m[1], "psnr_batch": 10 * math.log10(1.0 / max(m[1], 1e-12))})
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    r.check_errors()
    psnr1 = view_psnr(held_out)
    psnr_train = view_psnr(1)
    u1 = udf_on_off()
    print(json.dumps({
        "what": "training loop of runner_udf.py on a synthetic wire frame (13 segments, %d views %dx%d, view %d held out), HIP path only"
                % (a.views, a.res, a.res, held_out),
        "steps": a.steps, "rays_per_step": a.rays, "precision": a.precision, "wall_s": wall, "ms_per_step_incl_python": wall / a.steps * 1e3,
        "held_out_view_psnr_db": {"before": psnr0[0], "after": psnr1[0]}, "train_view_psnr_db_after": psnr_train[0],
        "mean_udf_on_wireframe": {"before": u0[0], "after": u1[0]}, "mean_udf_away_from_it": {"before": u0[1], "after": u1[1]},
        "variance": float(devn.variance), "beta": float(bet.beta), "gamma": float(bet.gamma),
        "trajectory": log}))


main()
