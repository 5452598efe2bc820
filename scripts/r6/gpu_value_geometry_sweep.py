#!/usr/bin/env python3
"""Round 6: time of ONE value launch (udf only, d8 w256, f16x3) per point count for every tile geometry of udf_mlp_fs2_kernel, to check the
launcher's size rule (udf_mlp_kernel.inc:launch_mlp_fs2_mode).  Needs the probe build (scripts/build_variant.sh geom -DEMAP_FS2_GEOM_ENV);
one subprocess per geometry (the override is read once per process), two interleaved rounds.
usage: python scripts/r6/gpu_value_geometry_sweep.py [P ...]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
Ps = sys.argv[1:] or [str(v) for v in (2048, 3072, 4096, 5120, 6144, 7168, 8192, 10240, 12288, 16384, 18432, 20480, 24576, 28672, 32704, 32768, 40960, 49152, 65536)]
geoms = {"rule": "0", "1x8": "18", "2x8": "28", "3x8": "38", "4x8": "48", "5x8": "58", "2x4": "24", "3x4": "34", "4x4": "44"}
if os.environ.get("GEOMS"):
    geoms = {k: v for k, v in geoms.items() if k in os.environ["GEOMS"].split(",")}
res = {}
for rnd in range(2):
    for name, code in geoms.items():
        env = dict(os.environ, EMAP_FS2_GEOM=code, EMAP_HIP_LIB=os.path.join(ROOT, "emap_amd/lib/geom/libemap_hip.so"))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts/gpu_time_value_sweep.py"), os.environ.get("PREC", "f16x3")] + Ps, env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(name, "FAILED", out.stderr[-400:], file=sys.stderr)
            continue
        for P, us in json.loads(line[-1])["value_launch_us_by_points"].items():
            res.setdefault(int(P), {}).setdefault(name, []).append(us)
print("# points   " + "  ".join(f"{n:>13s}" for n in geoms))
for P in sorted(res):
    best = min(geoms, key=lambda n: min(res[P].get(n, [1e9])))
    print(f"{P:8d}   " + "  ".join(f"{'/'.join(str(v) for v in res[P].get(n, [])):>13s}" for n in geoms) + f"   best: {best}")
