#!/usr/bin/env python3
"""bench.py - ray-samples/s of the EMAP render hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--rays 512] [--precision bf16x3|bf16]

One "step" = one ``UDFRendererBlending.render()`` forward (coarse sampling -> 4 occlusion-aware
up-sampling steps -> UDF MLP value+gradient at 128 samples/ray -> compositing) over one batch of
synthetic rays, entirely on the GPU (inputs resident in HBM before the timed region).
Workload at N=1: the north-star batch, 512 rays x (64 coarse + 64 fine in 4 steps) = 128 samples,
UDF MLP d=8 w=256 multires=10 (SURVEY.md par. 8d).  For N>1 every rank renders its own 512-ray shard of
one global batch (weak scaling); the forward path has no collective.

Prints ONE JSON line on rank 0 (see the driver contract in the task statement), including
  roofline     : the dominant kernel (final value+grad MLP pass) vs the dense bf16 MFMA peak
  cpu_baseline : the oracle (CPU restatement of the reference) timed on this box's host cores
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

F_POINT = 918016          # FLOP per MLP point-forward, d8 w256 (SURVEY par. 7.0 / 8d)
# measured max error vs the reference goldens (tests/test_gpu_parity.py asserts these bounds): (udf, grad_x udf),
# relative to the tensor's max magnitude
MODE_INFO = {
    "f16x3": {"dtype": "f16x3 (split-fp16 MFMA, 3 passes, fp32 accumulate)", "passes": 3, "udf_err": 8e-7, "grad_err": 2e-5,
              "meets_1e-4": True},
    "bf16x3": {"dtype": "bf16x3 (split-bf16 MFMA, 3 passes, fp32 accumulate)", "passes": 3, "udf_err": 7e-6, "grad_err": 3e-5,
               "meets_1e-4": False},
    "f16": {"dtype": "f16 (single-pass fp16 MFMA, fp32 accumulate)", "passes": 1, "udf_err": 6e-4, "grad_err": 1.6e-3,
            "meets_1e-4": False},
    "bf16": {"dtype": "bf16 (single-pass bf16 MFMA, fp32 accumulate)", "passes": 1, "udf_err": 5e-3, "grad_err": 1.5e-2,
             "meets_1e-4": False},
}
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak, MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--rays", type=int, default=512, help="rays per GPU")
    ap.add_argument("--precision", default=os.environ.get("EMAP_BENCH_PRECISION", "f16x3"),
                    choices=["f16x3", "bf16x3", "f16", "bf16"],
                    help="arithmetic of the MLP GEMMs for `value`; f16x3 is the mode that meets the 1e-4 parity gate")
    ap.add_argument("--no-other-modes", action="store_true", help="skip the short runs of the other precision modes")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rays", type=int, default=256)
    return ap.parse_args()


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


def cpu_baseline(state, kw, n_rays, threads):
    """The oracle (CPU restatement pinned to the reference by tests/golden) on the same synthetic workload."""
    from oracle import emap_oracle as O
    from emap_amd import synthetic
    torch.set_num_threads(threads)
    cfg = O.UDFConfig(d_in=3, d_out=1, d_hidden=kw["d_hidden"], n_layers=kw["n_layers"], skip_in=(4,), multires=kw["multires"])
    rcfg = O.RenderConfig(n_samples=64, n_importance=64, up_sample_steps=4)
    ro, rd, near, far, ds = synthetic.make_rays(n_rays, seed=1)
    tr = synthetic.make_t_rand(n_rays)
    var, bp, gp = torch.tensor([0.3]), torch.tensor([0.5]), torch.tensor([0.3])
    run = lambda: O.render(state, cfg, rcfg, ro, rd, near, far, ds, var, bp, gp, cos_anneal_ratio=1.0, t_rand=tr,
                           flip_saturation=0.9)
    t0 = time.perf_counter(); run(); first = time.perf_counter() - t0  # warm-up, also sizes the sample
    ts = []
    t_end = time.time() + 12.0
    while len(ts) < (1 if first > 8.0 else 3) or (time.time() < t_end and len(ts) < 20):
        t0 = time.perf_counter(); run(); ts.append(time.perf_counter() - t0)
    ts.sort()
    med = ts[len(ts) // 2]
    return {"value": n_rays * 128 / med, "unit": "ray-samples/s", "cores": threads, "kind": "port",
            "sample": f"{n_rays} rays x 128 samples, forward render(), median of {len(ts)} runs ({med*1e3:.0f} ms each), "
                      f"oracle/emap_oracle.py on torch CPU fp32"}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    n_gpus = max(a.gpus, world)

    from emap_amd import synthetic, _lib
    r, state, kw = build_renderer(dev, a.precision)
    S = r.samples_per_ray
    # one global batch from a shared seed, sliced by rank (SURVEY par. 8e)
    g = [t for t in synthetic.make_rays(a.rays * world, seed=1)]
    tr = synthetic.make_t_rand(a.rays * world)
    sl = slice(rank * a.rays, (rank + 1) * a.rays)
    ro, rd, near, far, ds = [t[sl].contiguous().to(dev) for t in g]
    tr = tr[sl].contiguous().to(dev)

    def step():
        return r.render(ro, rd, near, far, ds, cos_anneal_ratio=1.0, flip_saturation=0.9, t_rand=tr)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    L = _lib.lib()
    with torch.no_grad():
        for _ in range(a.warmup):
            out = step()
        barrier()
        _lib.check(L.emap_profile_enable(1))
        t0 = time.perf_counter()
        for _ in range(a.steps):
            out = step()
        barrier()
        dt = time.perf_counter() - t0
        _lib.check(L.emap_profile_enable(0))
    r.check_errors()
    if dist is not None:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    import ctypes as C
    kms, kn = C.c_float(), C.c_int()
    _lib.check(L.emap_profile_read(C.byref(kms), C.byref(kn)))
    k_avg_s = (kms.value / max(kn.value, 1)) * 1e-3

    if rank == 0:
        ms_per_step = dt / a.steps * 1e3
        value = a.rays * world * S / (dt / a.steps)
        # dominant kernel: the value+grad MLP launch over rays*S points; algorithmic work = value +
        # reverse-mode input gradient = 2F per point (SURVEY par. 8d)
        flops_launch = a.rays * S * 2 * F_POINT
        # udf_mlp.hip:mlp_variant: reverse-sweep kernel for grad launches of >= 10240 (f16x3) / 16384 (single-pass) points, not bf16x3
        dominant_kernel = (f"udf_mlp_rev_kernel<256,{a.precision}>" if (a.rays * S >= (10240 if a.precision == "f16x3" else 16384) and a.precision != "bf16x3")
                           else f"udf_mlp_fs2_kernel<256,{a.precision},4,grad>")
        ach = flops_launch / k_avg_s / 1e12 if k_avg_s > 0 else 0.0
        line = {
            "metric": "ray-samples/sec (UDF MLP + composite)", "value": value, "unit": "ray-samples/s",
            "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": MODE_INFO[a.precision]["dtype"],
            "data": "synthetic",
            "config": {"workload": f"{a.rays} rays/GPU x {S} samples (64 coarse + 64 fine in 4 up-sampling steps), "
                                   f"UDF MLP d=8 w=256 multires=10, forward render()",
                       "rays_per_gpu": a.rays, "samples_per_ray": S, "precision": a.precision,
                       "parallelism": f"dp{world} over rays, no collective in forward"},
            "roofline": {"bound": "mfma", "kernel": dominant_kernel + " (final value+grad pass)",
                         "achieved": ach, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / MFMA_PEAK_TFLOPS,
                         "avg_launch_us": k_avg_s * 1e6, "launches": kn.value,
                         "algorithmic_flops_per_launch": flops_launch, "traffic": None},
            "whole_render_algorithmic_tflops": value * 2639296 / 1e12,
            "parity": {"mode": a.precision, "udf_rel_err": MODE_INFO[a.precision]["udf_err"],
                       "grad_rel_err": MODE_INFO[a.precision]["grad_err"], "meets_1e-4": MODE_INFO[a.precision]["meets_1e-4"],
                       "checked_by": "tests/test_gpu_parity.py vs tests/golden (reference outputs)"},
        }
        tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                ent = tj.get(a.precision)
                if ent:
                    line["roofline"]["traffic"] = ent["hbm_bytes_per_launch"]
                    line["roofline"]["traffic_source"] = "profiles/r01_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this kernel)"
            except Exception:
                pass
        if not a.no_other_modes and world == 1:
            other = {}
            for mode in MODE_INFO:
                if mode == a.precision:
                    continue
                try:
                    r.precision = mode
                    with torch.no_grad():
                        for _ in range(5):
                            step()
                        torch.cuda.synchronize()
                        t1 = time.perf_counter()
                        n_o = max(20, a.steps // 4)
                        for _ in range(n_o):
                            step()
                        torch.cuda.synchronize()
                        dto = (time.perf_counter() - t1) / n_o
                    other[mode] = {"value": a.rays * S / dto, "ms_per_step": dto * 1e3, "udf_rel_err": MODE_INFO[mode]["udf_err"],
                                   "grad_rel_err": MODE_INFO[mode]["grad_err"], "meets_1e-4": MODE_INFO[mode]["meets_1e-4"]}
                except Exception as e:
                    other[mode] = {"error": repr(e)}
            r.precision = a.precision
            line["other_precision_modes"] = other
        if not a.no_cpu_baseline and world == 1:
            try:
                line["cpu_baseline"] = cpu_baseline(state, kw, a.cpu_rays, min(32, os.cpu_count() or 1))
            except Exception as e:  # the baseline must never take the GPU line down
                line["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
