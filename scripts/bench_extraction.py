#!/usr/bin/env python3
"""Measurement of the dense-grid extraction queries (SURVEY par. 8 f2; emap_amd/extraction.py).

One JSON line: wall time of get_udf_normals_grid on an N^3 grid with line directions (value pass over the grid, gradient
on the thresholded subset, 50 jittered gradients per thresholded point, null-direction kernel), the rates of its parts, the
HBM roofline of the null-direction kernel (12*k + 12 algorithmic bytes per point) and the CPU oracle (the reference
algorithm in torch, bounded sample) beside it.  Not the headline bench (bench.py); run on the GPU box:
    python scripts/bench_extraction.py [--N 128] [--frac 0.05]
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
from emap_amd import synthetic, extraction  # noqa: E402

HBM_PEAK_GBS = 8000.0


def ev_time(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=128)
    ap.add_argument("--frac", type=float, default=0.05, help="fraction of grid points below the threshold")
    ap.add_argument("--cpu-N", type=int, default=48)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    kw = dict(d_in=3, d_out=1, d_hidden=256, n_layers=8, skip_in=(4,), multires=10, bias=0.5)
    state = synthetic.make_udf_state(seed=42, pert=0.02, **kw)
    net = emap_amd.UDFNetwork(scale=1.0, precision="f16x3", **kw)
    net.load_state_dict(state)
    net = net.to(dev)
    N = a.N
    with torch.no_grad():
        df = extraction.get_udf_normals_grid(net.udf, net.gradient, N, -1.0, False, device=dev)[0]
        thr = float(df.reshape(-1)[torch.randperm(N ** 3, device=dev)[:200000]].quantile(a.frac))
        t_all, out = ev_time(lambda: extraction.get_udf_normals_grid(net.udf, net.gradient, N, thr, True, device=dev), reps=2)
        n_thr = int((out[0].reshape(-1) < thr).sum())

        def func_grad(xyz):                                   # the closure of Runner_UDF.extract_edge (runner_udf.py:522-526)
            gradients = net.gradient(xyz)
            gradients_mag = torch.linalg.norm(gradients, ord=2, dim=-1, keepdim=True)
            return gradients / (gradients_mag + 1e-5)
        t_closure, _ = ev_time(lambda: extraction.get_udf_normals_grid(net.udf, func_grad, N, thr, True, device=dev), reps=2)
        pts = out[3][:, :3].contiguous()
        t_val, _ = ev_time(lambda: net.hip_udf(pts, with_grad=False))
        big = torch.rand(1 << 20, 3, device=dev) * 2 - 1
        t_grad, _ = ev_time(lambda: net.hip_udf(big, with_grad=True))
        G = torch.randn(max(n_thr, 1), 50, 3, device=dev)
        t_nd, _ = ev_time(lambda: extraction.null_direction(G), reps=10)
    nd_bytes = n_thr * (12 * 50 + 12)
    line = {
        "metric": "grid points/sec (get_udf_normals_grid with line directions)", "value": N ** 3 / t_all, "unit": "points/s",
        "config": {"workload": f"{N}^3 grid, threshold at the {a.frac:.0%} quantile ({n_thr} points), sampling_N=50, f16x3",
                   "mlp_evaluations": N ** 3 + n_thr * 51},
        "seconds": t_all, "mlp_evals_per_s": (N ** 3 + n_thr * 51) / t_all,
        "seconds_through_the_runners_closure": t_closure, "points_per_s_through_the_runners_closure": N ** 3 / t_closure,
        "parts": {"value_pass_points_per_s": N ** 3 / t_val, "grad_points_per_s": (1 << 20) / t_grad,
                  "null_direction_points_per_s": n_thr / t_nd},
        "roofline_null_direction": {"bound": "hbm", "achieved": nd_bytes / t_nd / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": nd_bytes / t_nd / 1e9 / HBM_PEAK_GBS, "avg_launch_us": t_nd * 1e6,
                                    "algorithmic_bytes_per_point": 612},
        "data": "synthetic", "dtype": "f16x3",
    }
    if not a.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import emap_oracle as O
        torch.set_num_threads(min(32, os.cpu_count() or 1))
        cfg = O.UDFConfig()
        n = a.cpu_N
        df_c = O.udf_normals_grid(state, cfg, n, -1.0)[0]
        thr_c = float(df_c.reshape(-1).quantile(a.frac))
        n_c = int((df_c.reshape(-1) < thr_c).sum())
        noise = torch.randn(n_c, 50, 3)
        t0 = time.time()
        O.udf_normals_grid(state, cfg, n, thr_c, True, 50, 0.005, noise=noise)
        dt = time.time() - t0
        line["cpu_baseline"] = {"value": (n ** 3 + n_c * 51) / dt, "unit": "MLP evaluations/s", "cores": torch.get_num_threads(),
                                "kind": "port", "sample": f"{n}^3 grid, {n_c} thresholded points, oracle/emap_oracle.py udf_normals_grid, {dt:.1f} s"}
    print(json.dumps(line))


if __name__ == "__main__":
    main()
