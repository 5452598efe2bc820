#!/usr/bin/env python3
"""bench.py - ray-samples/s of the EMAP render hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode render|train] [--rays 512] [--precision f16x3|...]

--mode render (default, the headline): one "step" = one ``UDFRendererBlending.render()`` forward (coarse sampling -> 4
    occlusion-aware up-sampling steps -> UDF MLP value+gradient at 128 samples/ray -> compositing) over one batch of synthetic
    rays, entirely on the GPU (inputs resident in HBM before the timed region).
--mode train: one "step" = one optimizer step of ``emap_amd.parallel.Trainer``: that forward, the HIP backward
    (emap_render_bwd), the gradient all-reduce over RCCL when N > 1, and Adam.

Workload at N=1: the north-star batch, 512 rays x (64 coarse + 64 fine in 4 steps) = 128 samples, UDF MLP d=8 w=256
multires=10 (SURVEY.md par. 8d).  For N>1 every rank takes its own ``--rays`` shard of one global batch generated from a
shared seed (weak scaling; ``--global-rays R`` fixes the GLOBAL batch instead: strong scaling, e.g. 4096 for BASELINE config
C4).  ``python bench.py --gpus N`` without a launcher re-executes itself under ``torch.distributed.run`` with N ranks; under a
launcher it insists that WORLD_SIZE == N.  The forward path has no collective.

Timing: W untimed warm-up steps, then exactly K steps bracketed by barrier + torch.cuda.synchronize() on both sides (max
over ranks) -> ``value``; each step is additionally bracketed by HIP events (``ms_per_step_median``).  The roofline entry
comes from a SEPARATE short loop in which the library records HIP events around the dominant kernel on its launch stream.

Prints ONE JSON line on rank 0, including
  roofline     : the dominant kernel vs the dense fp16/bf16 MFMA peak
  parity       : measured in this run - the HIP kernels vs the committed reference goldens (tests/golden)
  cpu_baseline : the oracle (CPU restatement of the reference, pinned by the goldens) timed on this box's host cores
"""
import argparse
import ctypes as C
import gc
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
# the PMC passes of the dominant kernels recorded by scripts/profile_round.sh for this round's kernels (static: not measured in a bench run)
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "r06_traffic.json")
sys.path.insert(0, ROOT)

F_POINT = 918016          # FLOP per MLP point-forward, d8 w256 (SURVEY par. 7.0 / 8d)
A_FWD = 2639296           # algorithmic FLOP per ray-sample of a forward render (64+64/4)   (SURVEY par. 8d)
A_TRAIN = 6311360         # ... of a training step
MODE_DTYPE = {
    "f16x3": "f16x3 (split-fp16 MFMA, 3 passes, fp32 accumulate)",
    "f16x3m": "f16x3m (f16x3 with MX-fp6 cross terms in the forward sweep of the value+gradient pass too; gradient error 6.2e-5 of the 1e-4 gate)",
    "f16x3e": "f16x3e (f16x3 with f16 cross terms in both sweeps of the value+gradient pass - no MX fp6; gradient error 1.5e-5 of the 1e-4 gate)",
    "bf16x3": "bf16x3 (split-bf16 MFMA, 3 passes, fp32 accumulate)",
    "f16": "f16 (single-pass fp16 MFMA, fp32 accumulate)",
    "bf16": "bf16 (single-pass bf16 MFMA, fp32 accumulate)",
}
MFMA_PEAK_TFLOPS = 2500.0  # dense f16/bf16 MFMA peak, MI355X_MICROARCH.md
MFMA_MEASURED_TFLOPS = 2297.0   # scripts/probes/mfma_sustained.hip: v_mfma_f32_16x16x32_f16, two waves per SIMD, 140 ms sustained


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--mode", default="render", choices=["render", "train"])
    ap.add_argument("--path", default="native", choices=["native", "dropin", "dropin-fused", "dropin-patched"],
                    help="train mode, one GPU: native = emap_amd.parallel.Trainer (flat buffers, fused Adam); dropin = the reference runner's own "
                         "step on the drop-in classes (autograd, torch.optim.Adam, .item() reads); dropin-fused = the same with emap_amd's FusedAdam")
    ap.add_argument("--rays", type=int, default=512, help="rays per GPU (weak scaling)")
    ap.add_argument("--global-rays", type=int, default=0, help="rays of the GLOBAL batch, split over the ranks (strong scaling)")
    ap.add_argument("--precision", default=os.environ.get("EMAP_BENCH_PRECISION", "f16x3"), choices=list(MODE_DTYPE),
                    help="arithmetic of the MLP GEMMs for `value`; f16x3 is the mode that meets the 1e-4 parity gate")
    ap.add_argument("--eikonal-sync", default="exact_lagged", choices=["exact", "exact_lagged", "local"], help="train mode, N > 1 (emap_amd/parallel.py)")
    ap.add_argument("--allreduce", default="rccl", choices=["rccl", "oneshot"],
                    help="train mode, N > 1: the gradient bucket's collective - rccl = torch.distributed.all_reduce on --backend; oneshot = "
                         "emap_amd's peer-to-peer kernel over hipIpc mappings (csrc/allreduce.hip: every rank reads every peer's bucket directly "
                         "over xGMI, one launch)")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="replay the step from a captured hipGraph (auto: render mode and the one-rank training step yes - falling back to "
                         "eager launches if the capture fails -, multi-rank training no: RCCL inside a captured graph is untested here)")
    ap.add_argument("--settle-steps", type=int, default=150, help="untimed steps before the warm-up steps (clock settle)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N > 1: nccl (= RCCL, one rank per GPU) or gloo (debug: the ranks share the visible "
                         "GPUs round-robin and reduce through the host - runs every line of the N > 1 path on a one-GPU box)")
    ap.add_argument("--dry-run-nccl", action="store_true",
                    help="self-check of the RCCL branch: with >= 2 visible GPUs, run 2 ranks x 3 training steps over nccl (eager and from "
                         "per-phase graphs) and print one JSON line with the outcome; with one GPU it reports 'skipped'.  Meant to be the "
                         "first command on a multi-GPU lease.")
    ap.add_argument("--traffic", default=os.environ.get("EMAP_BENCH_TRAFFIC", "live"), choices=["live", "static", "off"],
                    help="roofline.traffic: live = measured after the timing by two rocprofv3 --pmc child passes of this command (one GPU; falls "
                         "back to static), static = the figure recorded under profiles/ for this round's kernels, off = null")
    ap.add_argument("--no-train-key", action="store_true", help="render mode, one GPU: skip the short training-step measurement")
    ap.add_argument("--no-other-modes", action="store_true", help="skip the short runs of the other precision modes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--cpu-rays", type=int, default=512)
    return ap.parse_args()


def relaunch_under_launcher(n, backend="nccl"):
    """`python bench.py --gpus N` without a launcher: spawn N ranks (one per GPU) and pass their output through."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def build_renderer(dev, precision):
    import emap_amd
    from emap_amd import synthetic
    kw = dict(d_in=3, d_out=1, d_hidden=256, n_layers=8, skip_in=(4,), multires=10, bias=0.5)
    state = synthetic.make_udf_state(seed=42, pert=0.02, **kw)
    net = emap_amd.UDFNetwork(scale=1.0, precision=precision, **kw)
    net.load_state_dict(state)
    net = net.to(dev)
    devn = emap_amd.SingleVarianceNetwork(0.3).to(dev)
    bet = emap_amd.BetaNetwork(0.5, 0.3, 0.3, 5e-5, True, True, False).to(dev)
    r = emap_amd.UDFRendererBlending(None, net, devn, bet, n_samples=64, n_importance=64, n_outside=0,
                                     up_sample_steps=4, perturb=1.0, device=dev)
    return r, state, kw


def measure_parity(dev, precision):
    """HIP kernels vs the committed reference goldens, in this run: MLP value / grad_x (g2), rendered edge (g5) and the
    training gradients on the reference's own samples are covered by tests/; here the two cheapest are re-measured."""
    import emap_amd
    gd = os.path.join(ROOT, "tests", "golden")
    g2 = np.load(os.path.join(gd, "g2_mlp.npz"))
    name = "d8w256L10"
    kw = dict(d_in=3, d_out=1, d_hidden=256, n_layers=8, skip_in=(4,), multires=10, bias=0.5)
    from emap_amd import synthetic
    net = emap_amd.UDFNetwork(scale=1.0, precision=precision, **kw)
    net.load_state_dict(synthetic.make_udf_state(seed=42, pert=0.02, **kw))
    net = net.to(dev)
    x = torch.from_numpy(g2["x"]).to(dev)
    rel = lambda a, b: float((a.double().cpu().reshape(-1) - b.double().reshape(-1)).abs().max() / b.double().abs().max())
    with torch.no_grad():
        u, g = net.hip_udf(x, with_grad=True)
        # the reverse-sweep kernel serves launches >= 10240 points: evaluate the same points inside a large launch too
        xb = torch.cat([x, torch.rand(20000, 3, device=dev) * 2 - 1])
        ub, gb = net.hip_udf(xb, with_grad=True)
    ref_u, ref_g = torch.from_numpy(g2[f"{name}.out"])[:, :1], torch.from_numpy(g2[f"{name}.grad"]).reshape(-1, 3)
    out = {"mode": precision, "udf_rel_err": rel(u, ref_u), "grad_rel_err": max(rel(g, ref_g), rel(gb[:x.shape[0]], ref_g)),
           "reference": "tests/golden/g2_mlp.npz, g5_render_c64_64_4.npz (recorded from the imported reference)"}
    g5 = np.load(os.path.join(gd, "g5_render_c64_64_4.npz"))
    devn = emap_amd.SingleVarianceNetwork(0.3).to(dev)
    bet = emap_amd.BetaNetwork(0.5, 0.3, 0.3, 5e-5, True, True, False).to(dev)
    r = emap_amd.UDFRendererBlending(None, net, devn, bet, 64, 64, 0, 4, 1.0, device=dev)
    a = [torch.from_numpy(g5[k]).to(dev) for k in ("rays_o", "rays_d", "near", "far", "depth_scale")]
    with torch.no_grad():
        o = r.render(*a, cos_anneal_ratio=1.0, perturb_overwrite=0, flip_saturation=0.9)
    out["edge_rel_err"] = rel(o["edge"], torch.from_numpy(g5["out.edge"]))
    out["meets_1e-4"] = bool(out["udf_rel_err"] <= 1e-4 and out["grad_rel_err"] <= 1e-4 and out["edge_rel_err"] <= 1e-4)
    return out


def train_key(dev, precision, rays, S, steps=40, warmup=10):
    """Short measurement of the OPTIMIZER STEP (forward + HIP backward + fused Adam, rays drawn on the device) appended to the
    default render line so that the driver's own run times it: ms per step, ray-samples/s, fraction of the MFMA peak for the
    algorithmic A_TRAIN FLOP per ray-sample, HBM traffic per step (static, from profiles/)."""
    import emap_amd
    from emap_amd import synthetic
    from emap_amd.parallel import Trainer
    r, _, _ = build_renderer(dev, precision)
    trainer = Trainer(r, lr_geo=1e-4, lr=5e-4, edge_weight=1.0, igr_weight=0.1, igr_ns_weight=0.0)
    meta, edges = synthetic.make_scene(n_images=8, H=400, W=400, seed=3)
    sampler = emap_amd.DeviceRaySampler.from_meta(meta, edges, device=dev, seed=1000)
    sampler.set_image_perm(list(range(8)))
    near_f, far_f = float(meta["scene_box"]["near"]), float(meta["scene_box"]["far"])

    def step():
        smp = sampler.gen_random_rays_patches_at(None, rays, importance_sample=True)
        batch = {"rays_o": smp["rays"]["rays_o"], "rays_d": smp["rays"]["rays_v"], "near": near_f, "far": far_f,
                 "depth_scale": smp["depth_scale"], "cos_anneal_ratio": 1.0, "flip_saturation": 0.9,
                 "t_rand": smp["t_rand"]}
        return trainer.step(batch, smp["rays"]["edge"], n_rays_global=rays)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for s_ev, e_ev in evs:
        s_ev.record()
        step()
        e_ev.record()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    r.check_errors()
    med = sorted(s_ev.elapsed_time(e_ev) for s_ev, e_ev in evs)[steps // 2]
    value = rays * S / dt
    out = {"metric": "ray-samples/sec (training step: render fwd + HIP bwd + Adam)", "steps": steps, "warmup": warmup,
           "ms_per_step": dt * 1e3, "ms_per_step_median": med, "value": value, "unit": "ray-samples/s",
           "whole_step_algorithmic_tflops": value * A_TRAIN / 1e12, "whole_step_frac": value * A_TRAIN / 1e12 / MFMA_PEAK_TFLOPS,
           "launch": "eager", "loss_after_run": trainer.last_stats.tolist(), "traffic": None}
    # the same step replayed from ONE captured hipGraph (sampler, jitter draw, pack, forward, backward, Adam, loss): the launch mode
    # `--mode train --graph on` times; `ms_per_step` above stays the eager figure of rounds 2-4
    try:
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(3):
                step()
        torch.cuda.current_stream(dev).wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            step()
        dt_g, med_g = _timed(graph.replay, steps, warmup)
        out["graph_replay"] = {"ms_per_step": dt_g * 1e3, "ms_per_step_median": med_g, "value": rays * S / dt_g}
        # round 6: the graph replay IS the native training step's launch mode (`--mode train` default) and the
        # figure of this key; the eager loop (host-sensitive: its mean was 15 % above its median on the driver's box) is kept beside it
        out["eager"] = {k: out[k] for k in ("ms_per_step", "ms_per_step_median", "value")}
        value = rays * S / dt_g
        out.update(ms_per_step=dt_g * 1e3, ms_per_step_median=med_g, value=value, launch="hipGraph replay",
                   whole_step_algorithmic_tflops=value * A_TRAIN / 1e12, whole_step_frac=value * A_TRAIN / 1e12 / MFMA_PEAK_TFLOPS)
    except Exception as e:   # pragma: no cover
        out["graph_replay"] = {"error": repr(e)}
        torch.cuda.synchronize()
    tpath = TRAFFIC_JSON
    if os.path.exists(tpath) and rays * S == 65536:
        try:
            ent = json.load(open(tpath)).get(f"train:{precision}")
            if ent:
                out["traffic"] = sum(ent.get("all", {}).values()) or ent.get("hbm_bytes_per_launch")
                out["traffic_source"] = f"STATIC: profiles/{os.path.basename(tpath)} (sum over the step's MLP / weight-gradient kernels, rocprofv3 --pmc), not measured in this run"
        except Exception:
            pass
    return out


def _timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for s_ev, e_ev in evs:
        s_ev.record()
        fn()
        e_ev.record()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return dt, sorted(s_ev.elapsed_time(e_ev) for s_ev, e_ev in evs)[steps // 2]


class SummaryWriter:
    """What tensorboard's SummaryWriter.add_scalar does with its value (torch/utils/tensorboard/_convert_np.py: make_np ->
    ``x.detach().cpu().numpy()``): a device tensor is READ at the call - one stream synchronisation each.  Nothing is written anywhere;
    the reads are counted.  (tensorboard itself is not in this image; the name is what runner_udf.py:47 instantiates, so
    emap_amd.dropin.train_wrapper finds and wraps it in this module exactly as it does in the runner's.)"""
    device_reads = 0

    def __init__(self, *a, **k):
        self.rows = 0

    def add_scalar(self, tag, scalar_value, global_step=None, *a, **k):
        if isinstance(scalar_value, torch.Tensor):
            if scalar_value.is_cuda:
                SummaryWriter.device_reads += 1
            scalar_value = float(scalar_value.detach().cpu().numpy())
        self.rows += 1

    def flush(self):
        pass

    def close(self):
        pass


class _RunnerLoop:
    """The optimizer step AS THE REFERENCE'S RUNNER TAKES IT: the body of ``Runner_UDF.train_udf``'s loop (src/runner/runner_udf.py:63-186,
    with runner_base.py:96-126's optimizer) restated statement by statement on the classes `emap_amd.dropin.install()` puts under
    `src.models.*` - the runner module itself needs pyhocon / cv2 / tensorboard (absent here).  ``train_udf`` has the runner's shape
    (creates ``self.writer = SummaryWriter(...)``, then loops), so ``emap_amd.dropin.train_wrapper`` applies to it unchanged."""

    def __init__(self, dev, precision, rays, fused_adam, n_samples=64, n_importance=64, up_sample_steps=4):
        import emap_amd
        from emap_amd import dropin, synthetic
        dropin.install()
        from src.models.udf_model import UDFNetwork, SingleVarianceNetwork, BetaNetwork      # = emap_amd's, through the aliases
        from src.models.udf_renderer_blending import UDFRendererBlending
        from src.models.loss import EdgeLoss
        kw = dict(d_in=3, d_out=1, d_hidden=256, n_layers=8, skip_in=(4,), multires=10, bias=0.5)
        self.udf_network = UDFNetwork(scale=1.0, **kw)
        self.udf_network.load_state_dict(synthetic.make_udf_state(seed=42, pert=0.02, **kw))
        self.udf_network = self.udf_network.to(dev)
        self.udf_network.precision = precision
        self.variance_network_fine = SingleVarianceNetwork(0.3).to(dev)
        self.beta_network = BetaNetwork(0.5, 0.3, 0.3, 5e-5, True, True, False).to(dev)
        self.learning_rate, self.learning_rate_geo = 5e-4, 1e-4
        groups = [{"params": list(self.udf_network.parameters()), "lr": self.learning_rate_geo},
                  {"params": list(self.variance_network_fine.parameters()) + list(self.beta_network.parameters())}, {"params": []}]
        if fused_adam:
            from emap_amd.parallel import FusedAdam
            self.optimizer = FusedAdam(groups[:2], lr=self.learning_rate)
        else:
            self.optimizer = torch.optim.Adam(groups, lr=self.learning_rate)
        self.renderer = UDFRendererBlending(None, self.udf_network, self.variance_network_fine, self.beta_network, n_samples=n_samples,
                                            n_importance=n_importance, n_outside=0, up_sample_steps=up_sample_steps, perturb=1.0, device=dev)
        self.edge_loss_func = EdgeLoss("mse")
        meta, edges = synthetic.make_scene(n_images=8, H=400, W=400, seed=3)
        self.sampler = emap_amd.DeviceRaySampler.from_meta(meta, edges, device=dev, seed=1000)
        self.near, self.far = float(meta["scene_box"]["near"]), float(meta["scene_box"]["far"])
        self.edge_weight, self.igr_weight, self.igr_ns_weight = 1.0, 0.1, 0.0
        self.batch_size, self.report_freq, self.iter_step = rays, 100, 0
        self.image_perm = list(range(8))
        self.beta_flag = True
        self.writer = None
        self.item_reads = 0
        self.last_loss = None

    def step(self):
        it = self.iter_step
        for g_, base in zip(self.optimizer.param_groups, (self.learning_rate_geo, self.learning_rate, self.learning_rate)):
            g_["lr"] = base * min(1.0, (it + 1) / 1000.0)                      # update_learning_rate (runner_base.py:128-161): warm-up ramp
        smp = self.sampler.gen_random_rays_patches_at(self.image_perm[it % len(self.image_perm)], self.batch_size, importance_sample=True)
        data = smp["rays"]
        rays_o, rays_d, true_edge = data["rays_o"], data["rays_v"], data["edge"]
        mask = torch.ones_like(true_edge).float()
        mask_sum = mask.sum() + 1e-5
        render_out = self.renderer.render(rays_o, rays_d, self.near, self.far, depth_scale=smp["depth_scale"], flip_saturation=0.9, pose=None,
                                          fx=None, fy=None, img_index=None, cos_anneal_ratio=1.0)
        udf, edge = render_out["udf"], render_out["edge"]
        variance, beta = render_out["variance"], render_out["beta"]
        gradient_error, gradient_error_near_surface = render_out["gradient_error"], render_out["gradient_error_near_surface"]
        udf_min = udf.min(dim=1)[0][mask[:, 0] > 0.5].mean()                   # noqa: F841  (the runner computes it too)
        edge_loss = self.edge_loss_func(edge, true_edge) * self.edge_weight
        psnr = 20.0 * torch.log10(1.0 / (((edge - true_edge) ** 2 * mask).sum() / mask_sum).sqrt())
        gradient_error_loss = gradient_error
        if (variance.mean() < 2 * beta.item() and variance.mean() < 0.01 and self.beta_flag
                and self.variance_network_fine.variance.requires_grad):
            self.beta_network.set_beta_trainable()
            self.beta_flag = False
        if self.variance_network_fine.variance.requires_grad is False and self.iter_step > 20000:
            self.variance_network_fine.set_trainable()
        igr_ns_weight = self.igr_ns_weight
        loss = edge_loss + gradient_error_near_surface * igr_ns_weight + gradient_error_loss * self.igr_weight
        self.last_loss = "PSNR: {:.2f}, Loss: {:.2f}".format(psnr, loss.item())  # par.set_description(...), runner_udf.py:164
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        self.iter_step += 1
        w = self.writer
        w.add_scalar("Loss/loss", loss, self.iter_step)
        w.add_scalar("Loss/edge_loss", edge_loss, self.iter_step)
        w.add_scalar("Loss/gradient_error_loss", gradient_error_loss * self.igr_weight, self.iter_step)
        w.add_scalar("Loss/gradient_error_near_surface", gradient_error_near_surface * igr_ns_weight, self.iter_step)
        w.add_scalar("Sta/variance", variance.mean(), self.iter_step)
        w.add_scalar("Sta/beta", beta.item(), self.iter_step)
        w.add_scalar("Sta/psnr", psnr, self.iter_step)

    def train_udf(self, timed):
        self.writer = SummaryWriter(log_dir=None)                              # runner_udf.py:47
        return timed(self.step)


def train_dropin_key(dev, precision, rays, steps=40, warmup=10, n_samples=64, n_importance=64, up_sample_steps=4, fused_adam=False,
                     patched=False):
    """See _RunnerLoop.  `patched`: ``emap_amd.dropin.train_wrapper`` around the SAME ``train_udf`` (what ``dropin.patch_runner(train=True)``
    does to ``Runner_UDF``): host-mirrored variance / beta / gamma, gradients installed by RenderFn, FusedAdam swapped in, deferred
    tensorboard scalars.  Host reads of device values per step are counted (writer) + known (.item() / format / bool of the loop body)."""
    from emap_amd import dropin
    loop = _RunnerLoop(dev, precision, rays, fused_adam, n_samples, n_importance, up_sample_steps)
    S = loop.renderer.samples_per_ray
    SummaryWriter.device_reads = 0
    run = lambda step: _timed(step, steps, warmup)
    train = _RunnerLoop.train_udf
    if patched:
        train = dropin.train_wrapper(train, sys.modules[__name__], single_thread_autograd=os.environ.get("EMAP_DROPIN_MT", "0") != "1")
    dt, med = train(loop, run)
    loop.renderer.check_errors()
    n_steps = steps + warmup
    writer_reads = SummaryWriter.device_reads / n_steps
    # loop body: unpatched - beta.item(), bool(variance.mean() < ...), format(psnr), loss.item(), beta.item() for the writer = 5 reads of
    # device tensors, each a stream synchronisation (the first waits for the forward, the fourth too after the loss kernels); patched -
    # format(psnr) + loss.item(): two reads behind ONE wait for the forward
    body_reads = 2 if patched else 5
    return {"metric": "ray-samples/sec (the reference runner's own step on the drop-in classes: autograd render + EdgeLoss + host reads + "
                      + (type(loop.optimizer).__name__) + (", emap_amd.dropin.train_wrapper" if patched else "") + ")",
            "rays": rays, "samples_per_ray": S, "steps": steps, "warmup": warmup, "ms_per_step": dt * 1e3, "ms_per_step_median": med,
            "value": rays * S / dt, "unit": "ray-samples/s", "optimizer": type(loop.optimizer).__name__, "patched": bool(patched),
            "host_syncs_per_step": body_reads + writer_reads, "host_waits_for_the_forward_per_step": 1 if patched else 2,
            "writer_device_reads_per_step": writer_reads, "last_progress_text": loop.last_loss,
            "reference": "src/runner/runner_udf.py:63-186, runner_base.py:96-126"}


def default_shape_key(dev, precision, steps=40, warmup=10):
    """The launch shape of the reference's own configuration (confs/ABC.conf:31,108-111): batch_size 1024, n_samples 64,
    n_importance 50, up_sample_steps 5 -> 64 + 5 x 10 = 114 samples per ray: forward render() and the native training step."""
    import emap_amd
    from emap_amd import synthetic
    from emap_amd.parallel import Trainer
    rays, S_c, S_f, K = 1024, 64, 50, 5
    r, _, _ = build_renderer(dev, precision)
    r = emap_amd.UDFRendererBlending(None, r.udf_network, r.deviation_network, r.beta_network, n_samples=S_c, n_importance=S_f, n_outside=0,
                                     up_sample_steps=K, perturb=1.0, device=dev)
    S = r.samples_per_ray
    ro, rd, near, far, ds = [t.contiguous().to(dev) for t in synthetic.make_rays(rays, seed=1)]
    tr = synthetic.make_t_rand(rays).to(dev)
    te = synthetic.make_true_edge(rays, seed=11).to(dev)

    def fwd():
        with torch.no_grad():
            return r.render(ro, rd, near, far, ds, cos_anneal_ratio=1.0, flip_saturation=0.9, t_rand=tr)
    dt_f, med_f = _timed(fwd, steps, warmup)
    trainer = Trainer(r, lr_geo=1e-4, lr=5e-4, edge_weight=1.0, igr_weight=0.1, igr_ns_weight=0.0)
    batch = {"rays_o": ro, "rays_d": rd, "near": near, "far": far, "depth_scale": ds, "cos_anneal_ratio": 1.0, "flip_saturation": 0.9, "t_rand": tr}
    dt_t, med_t = _timed(lambda: trainer.step(batch, te, n_rays_global=rays), steps, warmup)
    r.check_errors()
    e_ng = S_c + S_f * (K - 1) / K
    a_fwd, a_train = F_POINT * (e_ng / S + 2), F_POINT * (e_ng / S + 6)
    graphs = {}
    try:      # the same two steps replayed from a hipGraph each (round 6; the launch mode of the headline)
        replay_f = r.capture(ro, rd, near, far, ds, cos_anneal_ratio=1.0, flip_saturation=0.9, t_rand=tr)
        dt_g, med_g = _timed(replay_f, steps, warmup)
        graphs["render_graph"] = {"ms_per_step": dt_g * 1e3, "ms_per_step_median": med_g, "value": rays * S / dt_g, "unit": "ray-samples/s",
                                  "whole_step_frac_of_mfma_peak": rays * S / dt_g * a_fwd / 1e12 / MFMA_PEAK_TFLOPS, "launch": "hipGraph replay"}
        replay_t = trainer.capture(batch, te, n_rays_global=rays)
        dt_g, med_g = _timed(replay_t, steps, warmup)
        graphs["train_graph"] = {"ms_per_step": dt_g * 1e3, "ms_per_step_median": med_g, "value": rays * S / dt_g, "unit": "ray-samples/s",
                                 "whole_step_frac_of_mfma_peak": rays * S / dt_g * a_train / 1e12 / MFMA_PEAK_TFLOPS, "launch": "hipGraph replay"}
        r.check_errors()
    except Exception as e:   # pragma: no cover
        graphs["graph_error"] = repr(e)
        torch.cuda.synchronize()
    return {**graphs, "workload": f"{rays} rays x {S} samples ({S_c} coarse + {S_f} fine in {K} up-sampling steps), confs/ABC.conf:31,108-111",
            "rays": rays, "samples_per_ray": S, "steps": steps, "warmup": warmup,
            "render": {"ms_per_step": dt_f * 1e3, "ms_per_step_median": med_f, "value": rays * S / dt_f, "unit": "ray-samples/s",
                       "whole_step_frac_of_mfma_peak": rays * S / dt_f * a_fwd / 1e12 / MFMA_PEAK_TFLOPS, "launch": "eager"},
            "train": {"ms_per_step": dt_t * 1e3, "ms_per_step_median": med_t, "value": rays * S / dt_t, "unit": "ray-samples/s",
                      "whole_step_frac_of_mfma_peak": rays * S / dt_t * a_train / 1e12 / MFMA_PEAK_TFLOPS, "launch": "eager"}}


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_config(state, kw, n_rays, mode, threads, budget_s):
    """One thread-count configuration of the CPU baseline (module level: the all-cores configuration runs it in a child process)."""
    from oracle import emap_oracle as O
    from emap_amd import synthetic
    cfg = O.UDFConfig(d_in=3, d_out=1, d_hidden=kw["d_hidden"], n_layers=kw["n_layers"], skip_in=(4,), multires=kw["multires"])
    rcfg = O.RenderConfig(n_samples=64, n_importance=64, up_sample_steps=4)
    var, bp, gp = torch.tensor([0.3]), torch.tensor([0.5]), torch.tensor([0.3])

    def make(n):
        ro, rd, near, far, ds = synthetic.make_rays(n, seed=1)
        tr = synthetic.make_t_rand(n)
        te = synthetic.make_true_edge(n, seed=11)
        if mode == "train":
            return lambda: O.loss_and_param_grads(state, cfg, rcfg, ro, rd, near, far, ds, te, var, bp, gp, 1.0, 0.9, t_rand=tr,
                                                  edge_weight=1.0, igr_weight=0.1, igr_ns_weight=0.0)
        return lambda: O.render(state, cfg, rcfg, ro, rd, near, far, ds, var, bp, gp, cos_anneal_ratio=1.0, t_rand=tr, flip_saturation=0.9)

    torch.set_num_threads(threads)
    probe = make(32)
    t0 = time.perf_counter(); probe(); t_probe = time.perf_counter() - t0
    if t_probe > budget_s:                      # hopeless at this thread count: the probe is the sample
        return {"threads": threads, "rays": 32, "runs": 1, "ms": t_probe * 1e3, "value": 32 * 128 / t_probe}
    t0 = time.perf_counter(); probe(); t_probe = time.perf_counter() - t0
    n = 32
    while n * 2 <= n_rays and t_probe * (n * 2 / 32) <= budget_s / 3:
        n *= 2
    fn = make(n)
    ts, t_end = [], time.time() + budget_s
    while len(ts) < 1 or (time.time() < t_end and len(ts) < 5):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    return {"threads": threads, "rays": n, "runs": len(ts), "ms": med * 1e3, "value": n * 128 / med}


def cpu_baseline(state, kw, n_rays, mode):
    """The oracle (CPU restatement pinned to the reference by tests/golden) on the same synthetic workload: all host cores
    (BASELINE.md par. 3), 32 threads (eager torch on ~1e5-element ops stops scaling long before 256 threads and collapses when
    oversubscribed - measured 5e2 ray-samples/s on 256 threads vs 4.5e4 on 32) and one thread.  `value` is the best of them,
    `cores` says which.  Every configuration sizes its sample from a 32-ray probe so that the whole baseline stays near 30 s."""
    ncpu = os.cpu_count() or 1
    config = lambda t, b: _cpu_config(state, kw, n_rays, mode, t, b)
    res = [config(min(32, ncpu), 10.0), config(1, 6.0)]
    skipped = []
    if ncpu > 32:
        # all host cores (BASELINE.md par. 3): eager torch on these ~1e5-element ops collapses when oversubscribed (measured on the
        # 256-thread EPYC 9575F host: 50 s for 32 rays, 5e2 ray-samples/s), so that configuration runs in a child process with a
        # deadline instead of holding the bench for minutes
        code = ("import json,sys;sys.path.insert(0,%r);import bench,torch;from emap_amd import synthetic;"
                "kw=%r;st=synthetic.make_udf_state(seed=42,pert=0.02,**kw);"
                "print('CFG'+json.dumps(bench._cpu_config(st,kw,%d,%r,%d,8.0)))" % (ROOT, kw, n_rays, mode, ncpu))
        try:
            out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=25)
            ln = [x for x in out.stdout.splitlines() if x.startswith("CFG")]
            if ln:
                res.append(json.loads(ln[-1][3:]))
            else:
                skipped.append({"threads": ncpu, "note": "the child process produced no sample"})
        except subprocess.TimeoutExpired:
            # no sample = no entry: eager torch on 256 threads collapses on these ~1e5-element ops (round 3: 50 s for 32 rays)
            skipped.append({"threads": ncpu, "note": "no 32-ray run finished within 25 s (oversubscribed eager torch); not part of `best of`"})
    torch.set_num_threads(min(32, ncpu))
    best = max(res, key=lambda c: c["value"])
    one = [c for c in res if c["threads"] == 1][0]
    what = "forward render()" if mode == "render" else "forward + loss.backward() (autograd double backward)"
    return {"value": best["value"], "unit": "ray-samples/s", "cores": best["threads"], "kind": "port", "cpu_model": cpu_model(),
            "host_cores": ncpu, "single_thread_value": one["value"],
            "by_threads": {str(c["threads"]): {k: c[k] for k in c if k != "threads"} for c in res},
            "configurations_without_a_sample": skipped,
            "sample": f"{best['rays']} rays x 128 samples, {what}, median of {best['runs']} runs ({best['ms'] or 0:.0f} ms each) on "
                      f"{best['threads']} threads (best of {sorted(c['threads'] for c in res)} threads); oracle/emap_oracle.py on torch CPU fp32"}


def measure_traffic_live(mode, precision, rays, dominant_key):
    """HBM-side bytes per launch of the dominant kernel, MEASURED for this command on this box: two `rocprofv3 --pmc` child runs of this
    script (FETCH_SIZE, then WRITE_SIZE - separate passes, no tracing, as MI355X_MICROARCH.md prescribes), a few steps each, after
    all timing is done.  Bytes = 2 x FETCH_SIZE (the guide's gfx950 correction for wide coalesced reads) + WRITE_SIZE, both reported in
    KiB.  Returns (bytes, note) or (None, reason)."""
    import csv, glob, shutil, tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="emap_traffic_", dir=os.environ.get("TMPDIR", "/tmp"))
    vals = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, ctr)
            cmd = [exe, "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.abspath(__file__),
                   "--mode", mode, "--rays", str(rays), "--steps", "5", "--warmup", "2", "--settle-steps", "5", "--precision", precision,
                   "--no-cpu-baseline", "--no-other-modes", "--no-parity", "--no-train-key", "--traffic", "off", "--graph", "off"]
            env = dict(os.environ, TMPDIR=os.environ.get("TMPDIR", "/tmp"))
            p_ = subprocess.run(cmd, cwd=tmp, env=env, capture_output=True, text=True, timeout=240)
            f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if p_.returncode != 0 or not f:
                return None, f"rocprofv3 --pmc {ctr} failed (rc {p_.returncode})"
            per = {}
            for r_ in csv.DictReader(open(f[0])):
                if dominant_key in r_["Kernel_Name"] and r_["Counter_Name"] == ctr:
                    per[r_["Dispatch_Id"]] = per.get(r_["Dispatch_Id"], 0.0) + float(r_["Counter_Value"])
            if not per:
                return None, f"no dispatch of {dominant_key} in the {ctr} pass"
            vals[ctr] = (sum(per.values()) / len(per), len(per))
        b = vals["FETCH_SIZE"][0] * 1024 * 2 + vals["WRITE_SIZE"][0] * 1024
        return b, (f"MEASURED in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of this command (mean over {vals['FETCH_SIZE'][1]} / "
                   f"{vals['WRITE_SIZE'][1]} launches of {dominant_key}); 2 x FETCH_SIZE + WRITE_SIZE, KiB units (MI355X_MICROARCH.md)")
    except subprocess.TimeoutExpired:
        return None, "rocprofv3 pass timed out"
    except Exception as e:   # pragma: no cover
        return None, repr(e)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def dry_run_nccl():
    """`python bench.py --dry-run-nccl`: constructs the nccl (= RCCL) process group on 2 GPUs and takes three training steps eagerly and
    three from per-phase graphs, so that the first multi-GPU lease is not spent debugging the launch path."""
    n = torch.cuda.device_count()
    if n < 2:
        print(json.dumps({"dry_run_nccl": "skipped", "reason": f"{n} GPU(s) visible; the RCCL branch needs 2"}), flush=True)
        return 0
    outs = {}
    # eager and per-phase graphs over RCCL; then the same step with the gradient bucket through the peer-to-peer kernel (--allreduce oneshot)
    # in both eikonal modes a first scaling run would compare: local (ONE collective per step) and exact_lagged (two)
    for name, extra in (("eager", ["--graph", "off"]), ("per_phase_graphs", ["--graph", "on"]),
                        ("oneshot_local", ["--graph", "off", "--allreduce", "oneshot", "--eikonal-sync", "local"]),
                        ("oneshot_exact_lagged", ["--graph", "off", "--allreduce", "oneshot"])):
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "2", "--mode", "train", "--steps", "3", "--warmup", "1", "--settle-steps", "2",
               "--no-cpu-baseline", "--no-parity"] + extra
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            ln = [x for x in p.stdout.splitlines() if x.startswith("{")]
            outs[name] = {"rc": p.returncode, "ms_per_step": (json.loads(ln[-1])["ms_per_step"] if ln else None),
                          "stderr_tail": p.stderr[-400:] if p.returncode else ""}
        except subprocess.TimeoutExpired:
            outs[name] = {"rc": None, "error": "timeout after 600 s"}
    ok = all(o.get("rc") == 0 for o in outs.values())
    print(json.dumps({"dry_run_nccl": "ok" if ok else "FAILED", "ranks": 2, **outs}), flush=True)
    return 0 if ok else 1


def main():
    a = parse()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    if a.dry_run_nccl:
        sys.exit(dry_run_nccl())
    launched = "WORLD_SIZE" in os.environ
    if a.gpus > 1 and not launched:
        if torch.cuda.device_count() < a.gpus and a.backend == "nccl":
            raise SystemExit(f"bench.py --gpus {a.gpus}: only {torch.cuda.device_count()} GPU(s) visible")
        sys.exit(relaunch_under_launcher(a.gpus, a.backend))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"bench.py --gpus {a.gpus} was launched with WORLD_SIZE={world}: refusing to report n_gpus != ranks")
    device_index = local_rank if a.backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
        assert dist.get_world_size() == a.gpus

    from emap_amd import synthetic, _lib
    rays = a.rays
    scaling = "weak"
    if a.global_rays:
        assert a.global_rays % world == 0
        rays, scaling = a.global_rays // world, "strong"
    r, state, kw = build_renderer(dev, a.precision)
    S = r.samples_per_ray
    # one global batch from a shared seed, sliced by rank (SURVEY par. 8e)
    g = [t for t in synthetic.make_rays(rays * world, seed=1)]
    tr = synthetic.make_t_rand(rays * world)
    te = synthetic.make_true_edge(rays * world, seed=11)
    sl = slice(rank * rays, (rank + 1) * rays)
    ro, rd, near, far, ds = [t[sl].contiguous().to(dev) for t in g]
    tr, te = tr[sl].contiguous().to(dev), te[sl].contiguous().to(dev)

    if a.mode == "train" and a.path != "native":
        # the reference runner's own optimizer step on the drop-in classes (one GPU: EMAP itself is single-GPU)
        if world != 1:
            raise SystemExit("bench.py --path dropin runs on one GPU (the reference's loop has no data parallelism)")
        res = train_dropin_key(dev, a.precision, rays, steps=a.steps, warmup=a.warmup + a.settle_steps // 10, fused_adam=(a.path == "dropin-fused"),
                               patched=(a.path == "dropin-patched"))
        line = {"metric": res["metric"], "value": res["value"], "unit": "ray-samples/s", "n_gpus": 1, "steps": a.steps, "warmup": a.warmup,
                "ms_per_step": res["ms_per_step"], "ms_per_step_median": res["ms_per_step_median"], "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": MODE_DTYPE[a.precision], "data": "synthetic",
                "config": {"workload": f"{rays} rays x {res['samples_per_ray']} samples, UDF MLP d=8 w=256 multires=10, optimizer step as "
                                       f"src/runner/runner_udf.py:63-168 takes it, on the drop-in classes ({a.path})",
                           "mode": "train", "path": a.path, "precision": a.precision, "host_syncs_per_step": res["host_syncs_per_step"],
                           "host_waits_for_the_forward_per_step": res["host_waits_for_the_forward_per_step"]},
                "whole_step_frac_of_mfma_peak": res["value"] * A_TRAIN / 1e12 / MFMA_PEAK_TFLOPS, "last_progress_text": res["last_progress_text"]}
        print(json.dumps(line), flush=True)
        return

    trainer = None
    if a.mode == "train":
        # The training loop of runner_udf.py:79-168 on a synthetic wire-frame scene (no datasets travel): every step draws its
        # rays ON THE DEVICE (emap_amd.DeviceRaySampler = Dataset.gen_random_rays_patches_at, importance_sample=True),
        # jitters them with a device Philox draw, and takes one optimizer step.  No host->device copy per step.
        import emap_amd
        from emap_amd.parallel import Trainer
        trainer = Trainer(r, lr_geo=1e-4, lr=5e-4, edge_weight=1.0, igr_weight=0.1, igr_ns_weight=0.0, eikonal_sync=a.eikonal_sync,
                          allreduce=a.allreduce)
        meta, edges = synthetic.make_scene(n_images=8, H=400, W=400, seed=3)
        sampler = emap_amd.DeviceRaySampler.from_meta(meta, edges, device=dev, seed=1000 + rank)   # every rank draws its own shard
        sampler.set_image_perm(list(range(8)))
        near_f, far_f = float(meta["scene_box"]["near"]), float(meta["scene_box"]["far"])

        def eager_step():
            smp = sampler.gen_random_rays_patches_at(None, rays, importance_sample=True)
            batch = {"rays_o": smp["rays"]["rays_o"], "rays_d": smp["rays"]["rays_v"], "near": near_f, "far": far_f,
                     "depth_scale": smp["depth_scale"], "cos_anneal_ratio": 1.0, "flip_saturation": 0.9,
                     "t_rand": smp["t_rand"]}       # the jitter of render() :719 comes with the rays (one device draw, no torch.rand launch)
            return trainer.step(batch, smp["rays"]["edge"], n_rays_global=rays * world)
    else:
        def eager_step():
            with torch.no_grad():
                return r.render(ro, rd, near, far, ds, cos_anneal_ratio=1.0, flip_saturation=0.9, t_rand=tr)

    # hipGraph: the launch chain of a step (15 kernels forward, +8 backward, + torch's elementwise / Adam kernels) replays as one
    # graph launch.  Multi-rank training keeps eager launches unless --graph on (an RCCL collective inside a captured graph is
    # not something this build could test on its one-GPU boxes).
    # Measured on MI355X (round 2, same box): in a long loop replay and eager launches give the same step time (render 0.70 vs 0.70
    # ms, train 2.39 vs 2.39 ms at 200 steps): the stream is GPU-bound and the launches hide behind the kernels.  A timed region that
    # is bracketed by synchronisations starts with an empty queue, though, and with 20 steps the host-bound first step is 2 % of it
    # (0.789 vs 0.773 ms per step): the forward render replays from a graph by default; since round 6 the one-rank training step too
    # (the eager loop's mean sat 15 % above its median on the driver's box: host jitter, not GPU time).
    step, launch = eager_step, "eager"
    want_graph = a.graph == "on" or (a.graph == "auto" and (a.mode == "render" or world == 1))
    if want_graph:
        try:
            if a.mode == "train" and world > 1:
                # several ranks: one graph per device phase, the collectives launched between the replays (Trainer.capture); the
                # rays are still drawn per step, outside the graphs, and copied into the static batch
                smp0 = sampler.gen_random_rays_patches_at(None, rays, importance_sample=True)
                batch0 = {"rays_o": smp0["rays"]["rays_o"], "rays_d": smp0["rays"]["rays_v"], "near": near_f, "far": far_f,
                          "depth_scale": smp0["depth_scale"], "cos_anneal_ratio": 1.0, "flip_saturation": 0.9,
                          "t_rand": smp0["t_rand"]}
                replay = trainer.capture(batch0, smp0["rays"]["edge"], n_rays_global=rays * world, segmented=True)

                def step():
                    smp = sampler.gen_random_rays_patches_at(None, rays, importance_sample=True)
                    return replay({"rays_o": smp["rays"]["rays_o"], "rays_d": smp["rays"]["rays_v"], "depth_scale": smp["depth_scale"],
                                   "t_rand": smp["t_rand"]}, smp["rays"]["edge"])
            elif a.mode == "train":
                side = torch.cuda.Stream(device=dev)
                side.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(side):
                    for _ in range(3):
                        eager_step()
                torch.cuda.current_stream(dev).wait_stream(side)
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    graph_out = eager_step()

                def step():
                    graph.replay()
                    return graph_out
            else:
                step = r.capture(ro, rd, near, far, ds, cos_anneal_ratio=1.0, flip_saturation=0.9, t_rand=tr)
            launch = "hipGraph replay" if not (a.mode == "train" and world > 1) else "hipGraph replay per phase, eager collectives"
        except Exception as e:   # pragma: no cover
            if a.graph == "on":
                raise
            step, launch = eager_step, f"eager (graph capture failed: {e!r})"
            torch.cuda.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    parity = None
    if rank == 0 and not a.no_parity:
        try:
            parity = measure_parity(dev, a.precision)
        except Exception as e:   # pragma: no cover
            parity = {"error": repr(e)}

    L = _lib.lib()
    # Power-state settle: the clocks of a fresh MI355X box take tens of milliseconds of load to reach their steady level, more than
    # the driver's `--warmup 5` (3-10 ms) provides; a fixed number of untimed steps (the same on every rank: they contain the
    # collectives in train mode) runs before the W warm-up steps the contract asks for.  Nothing of it is inside the timed region.
    for _ in range(a.settle_steps):
        step()
    for _ in range(a.warmup):
        step()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.steps)]
    # no cyclic-GC pause inside the timed region: the host runs only a few ms ahead of the GPU at 0.65 ms per step, and a generation-2
    # collection (tens of ms with torch's object graph) would starve the queue
    gc.collect()
    gc.disable()
    barrier()
    t0 = time.perf_counter()
    for s_ev, e_ev in evs:
        s_ev.record()
        step()
        e_ev.record()
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    r.check_errors()
    per_step = sorted(s_ev.elapsed_time(e_ev) for s_ev, e_ev in evs)
    med_ms = per_step[len(per_step) // 2]
    if dist is not None:
        tt = torch.tensor([dt, med_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, med_ms = float(tt[0]), float(tt[1])

    # roofline: a second, short loop (eager launches: events cannot be recorded inside a captured graph) with the library's HIP
    # events around the dominant kernel - not in the headline loop
    which = 1 if a.mode == "train" else 0
    _lib.check(L.emap_profile_enable(1))
    n_prof = min(a.steps, 20)
    for _ in range(n_prof):
        eager_step()
    torch.cuda.synchronize()
    _lib.check(L.emap_profile_enable(0))
    kms, kn = C.c_float(), C.c_int()
    _lib.check(L.emap_profile_read_kernel(which, C.byref(kms), C.byref(kn)))
    k_avg_s = (kms.value / max(kn.value, 1)) * 1e-3
    kclk = C.c_float()   # shader clock during the last profiled launch of the dominant kernel (s_memtime / s_memrealtime inside the kernel)
    _lib.check(L.emap_profile_read_clock(which, C.byref(kclk)))
    bwd_us = None
    if a.mode == "train":   # per-step sums of the two big backward kernels (all chunks of a step)
        wms, wn = C.c_float(), C.c_int()
        _lib.check(L.emap_profile_read_kernel(2, C.byref(wms), C.byref(wn)))
        bwd_us = {"udf_mlp_vjp_us_per_step": kms.value * 1e3 / n_prof, "wgrad_us_per_step": wms.value * 1e3 / n_prof,
                  "launches_per_step": [max(kn.value, 1) / n_prof, max(wn.value, 1) / n_prof]}
    launches_per_step = max(kn.value, 1) / n_prof      # > 1 when the backward sweep runs in chunks (> 524 288 points per GPU, or a bounded workspace)
    loss_now = trainer.last_stats.tolist() if trainer is not None else None

    if rank == 0:
        ms_per_step = dt / a.steps * 1e3
        value = rays * world * S / (dt / a.steps)
        if a.mode == "train":
            # dominant kernel: the per-point sweep of the MLP double backward (value + one tangent column forward, two adjoint
            # columns backward) = 2F + 2F of the 6F the backward has to do per point; the weight-gradient GEMMs are a second kernel
            flops_launch = rays * S * 4 * F_POINT / launches_per_step
            dominant = f"udf_mlp_vjp_kernel<256,{a.precision},8> (forward recompute + reverse sweep of the double backward)"
            alg = A_TRAIN
            metric = "ray-samples/sec (training step: render fwd + HIP bwd + all-reduce + Adam)"
            workload = ("optimizer step (emap_amd.parallel.Trainer) on a synthetic 8-view 400x400 wire-frame scene, rays drawn on the "
                        "device per step (DeviceRaySampler, importance_sample=True)")
        else:
            # dominant kernel: the value+grad MLP launch over rays*S points; algorithmic work = value + reverse-mode input
            # gradient = 2F per point (SURVEY par. 8d)
            flops_launch = rays * S * 2 * F_POINT / launches_per_step
            rev = rays * S >= (10240 if a.precision in ("f16x3", "f16x3m", "f16x3e", "bf16x3") else 16384)
            dominant = (f"udf_mlp_rev32_kernel<256,{a.precision}>" if rev else f"udf_mlp_fs2_kernel<256,{a.precision},4,grad>") + " (final value+grad pass)"
            alg = A_FWD
            metric = "ray-samples/sec (UDF MLP + composite)"
            workload = "forward render()"
        ach = flops_launch / k_avg_s / 1e12 if k_avg_s > 0 else 0.0
        line = {
            "metric": metric, "value": value, "unit": "ray-samples/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step, "ms_per_step_median": med_ms,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": MODE_DTYPE[a.precision], "data": "synthetic",
            "config": {"workload": f"{rays} rays/GPU x {S} samples (64 coarse + 64 fine in 4 up-sampling steps), "
                                   f"UDF MLP d=8 w=256 multires=10, {workload}",
                       "rays_per_gpu": rays, "rays_global": rays * world, "samples_per_ray": S, "precision": a.precision,
                       "mode": a.mode, "launch": launch,
                       "parallelism": f"dp{world} over rays" + (", no collective in forward" if a.mode == "render" else
                                                               f", {trainer.collectives_per_step} collective(s) per step "
                                                               f"(eikonal_sync={a.eikonal_sync}: "
                                                               + ("20 B statistics (SUM) + 8 B range maxima (MAX) + " if a.eikonal_sync == "exact" and world > 1 else "")
                                                               + ("20 B statistics (SUM) + " if a.eikonal_sync == "exact_lagged" and world > 1 else "")
                                                               + f"one flat {4 * trainer.flat.grad.numel()} B gradient all-reduce"
                                                               + (" whose tail carries every rank's range maxima for the NEXT step" if a.eikonal_sync == "exact_lagged" and world > 1 else "")
                                                               + ")"),
                       "collectives_per_step": (trainer.collectives_per_step if a.mode == "train" else 0),
                       "eikonal_sync": (a.eikonal_sync if a.mode == "train" else None),
                       "gradient_allreduce": ((a.allreduce if world > 1 else None) if a.mode == "train" else None),
                       "ranks": world, "backend": (a.backend if world > 1 else None), "rccl_ranks": (world if (world > 1 and a.backend == "nccl") else 0),
                       "devices": ([(i if a.backend == "nccl" else i % torch.cuda.device_count()) for i in range(world)]),
                       "settle_steps": a.settle_steps},
            "roofline": {"bound": "mfma", "kernel": dominant, "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": ach / MFMA_PEAK_TFLOPS, "avg_launch_us": k_avg_s * 1e6, "launches": kn.value,
                         "algorithmic_flops_per_launch": flops_launch, "traffic": None,
                         # the sustained dense f16 rate measured on this hardware (profiles/r02_probe_mfma_sustained.txt), SURVEY par. 8d
                         "peak_measured": MFMA_MEASURED_TFLOPS, "frac_of_measured_peak": ach / MFMA_MEASURED_TFLOPS,
                         # the kernel runs at the 1400 W package power cap on real data (profiles/r03_probe_power.txt): the shader clock it
                         # actually got, and the nominal peak scaled to it (2.5 PF/s is quoted at 2.4 GHz)
                         "shader_clock_mhz": (kclk.value or None),
                         "frac_of_peak_at_measured_clock": (ach / (MFMA_PEAK_TFLOPS * kclk.value / 2400.0) if kclk.value else None)},
            "whole_step_algorithmic_tflops": value * alg / 1e12,
            "whole_step_frac_of_mfma_peak": value * alg / 1e12 / MFMA_PEAK_TFLOPS / world,
        }
        if loss_now is not None:
            line["loss_after_run"] = loss_now
        if bwd_us is not None:
            line["backward_kernels"] = bwd_us
        if parity is not None:
            line["parity"] = parity
        tpath = TRAFFIC_JSON
        live_note = None
        if a.traffic == "live" and world == 1:
            tb, live_note = measure_traffic_live(a.mode, a.precision, rays, "udf_mlp_vjp_kernel" if a.mode == "train" else "udf_mlp_rev")
            if tb is not None:
                line["roofline"]["traffic"] = tb
                line["roofline"]["traffic_source"] = live_note
        if line["roofline"]["traffic"] is None and a.traffic != "off" and os.path.exists(tpath):
            try:
                ent = json.load(open(tpath)).get(f"{a.mode}:{a.precision}")
                # the counters were collected on launches of 65 536 points (512 rays x 128 samples): no figure is quoted for a launch of
                # another size (the backward sweep runs in one launch up to 524 288 points)
                same_launch = rays * S == 65536
                if ent and same_launch:
                    line["roofline"]["traffic"] = ent["hbm_bytes_per_launch"]
                    line["roofline"]["traffic_source"] = (f"STATIC: profiles/{os.path.basename(tpath)}, recorded by rocprofv3 --pmc passes of this kernel at this launch "
                                                          "size (scripts/profile_round.sh; MI355X_MICROARCH.md corrections), not measured in this run"
                                                          + (f" (live measurement unavailable: {live_note})" if live_note else ""))
            except Exception:
                pass
        if world == 1 and a.mode == "render" and launch.startswith("hipGraph") and not a.no_other_modes:
            # the headline replays the SAME rays from one graph (rays, weights and the sigma' slabs stay cache-resident).  Beside
            # it: the same graph fed a DIFFERENT ray batch at every replay (8 batches resident in HBM, copied into the graph's static inputs:
            # four small device-to-device copies per step, inside the timed region)
            try:
                from emap_amd import synthetic as _syn
                batches = []
                for sd in range(101, 109):
                    rb = [t.contiguous().to(dev) for t in _syn.make_rays(rays, seed=sd)]
                    batches.append((rb, _syn.make_t_rand(rays, seed=sd + 50).to(dev)))
                k = [0]

                def rot():
                    rb, trb = batches[k[0] % len(batches)]
                    k[0] += 1
                    return step(rb[0], rb[1], rb[2], rb[3], rb[4], t_rand=trb)
                dt_r, med_r = _timed(rot, max(a.steps, 40), 10)
                r.check_errors()
                line["rotating_rays"] = {"ms_per_step": dt_r * 1e3, "ms_per_step_median": med_r, "value": rays * S / dt_r, "unit": "ray-samples/s",
                                         "batches": len(batches), "note": "same hipGraph, a different 512-ray batch copied in before every replay"}
                step(ro, rd, near, far, ds, t_rand=tr)      # back to the headline batch
            except Exception as e:
                line["rotating_rays"] = {"error": repr(e)}
        if not a.no_other_modes and world == 1 and a.mode == "render":
            other = {}
            for mode in MODE_DTYPE:
                if mode == a.precision:
                    continue
                try:
                    r.precision = mode
                    for _ in range(10):
                        eager_step()
                    torch.cuda.synchronize()
                    # median of per-step event times: these short secondary runs otherwise pick up one-off stalls (a first-use code
                    # object load or an allocator trim of ~70 ms landed in one of them in about every second run)
                    n_o = max(20, a.steps // 4)
                    evo = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n_o)]
                    for s_ev, e_ev in evo:
                        s_ev.record()
                        eager_step()
                        e_ev.record()
                    torch.cuda.synchronize()
                    dto = sorted(s_ev.elapsed_time(e_ev) for s_ev, e_ev in evo)[n_o // 2] * 1e-3
                    pm = measure_parity(dev, mode) if not a.no_parity else {}
                    other[mode] = {"value": rays * S / dto, "ms_per_step": dto * 1e3, "udf_rel_err": pm.get("udf_rel_err"),
                                   "grad_rel_err": pm.get("grad_rel_err"), "edge_rel_err": pm.get("edge_rel_err"),
                                   "meets_1e-4": pm.get("meets_1e-4")}
                except Exception as e:
                    other[mode] = {"error": repr(e)}
            r.precision = a.precision
            line["other_precision_modes"] = other
        if a.mode == "render" and world == 1 and not a.no_train_key:
            try:
                line["train"] = train_key(dev, a.precision, rays, S)
            except Exception as e:   # the secondary measurement must never take the headline down
                line["train"] = {"error": repr(e)}
            try:    # the reference runner's own step through the drop-in classes
                line["train_dropin"] = train_dropin_key(dev, a.precision, rays)
                line["train_dropin_fused_adam"] = train_dropin_key(dev, a.precision, rays, fused_adam=True)
                # round 5: the same loop under emap_amd.dropin.train_wrapper (= dropin.patch_runner(train=True) on Runner_UDF)
                line["train_dropin_patched"] = train_dropin_key(dev, a.precision, rays, patched=True)
                if isinstance(line.get("train"), dict) and line["train"].get("ms_per_step"):
                    for k in ("train_dropin", "train_dropin_fused_adam", "train_dropin_patched"):
                        line[k]["vs_native_trainer"] = line[k]["ms_per_step"] / line["train"]["ms_per_step"]      # the graph replay
                        if isinstance(line["train"].get("eager"), dict):
                            line[k]["vs_native_trainer_eager"] = line[k]["ms_per_step"] / line["train"]["eager"]["ms_per_step"]
            except Exception as e:
                line["train_dropin"] = {"error": repr(e)}
            try:
                line["reference_default_shape"] = default_shape_key(dev, a.precision)
            except Exception as e:
                line["reference_default_shape"] = {"error": repr(e)}
        if not a.no_cpu_baseline and world == 1:
            try:
                line["cpu_baseline"] = cpu_baseline(state, kw, a.cpu_rays, a.mode)
            except Exception as e:  # the baseline must never take the GPU line down
                line["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(line), flush=True)
    if trainer is not None:
        trainer.close()                 # a launch that gave up on a peer must fail the bench, not report a number
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
