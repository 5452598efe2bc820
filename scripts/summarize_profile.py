#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (rocprofv3 csv output of scripts/profile_round.sh) into the small committed summaries under
profiles/.  usage: summarize_profile.py <tag> <precision> <mode>"""
import collections
import csv
import glob
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
prec = sys.argv[2] if len(sys.argv) > 2 else "f16x3"
mode = sys.argv[3] if len(sys.argv) > 3 else "render"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "gpurun_out", f"profiles_{tag}")      # copied into profiles/ by hand after inspection
os.makedirs(dst, exist_ok=True)
# the kernels whose counters are reported, by mode
PER_RAY = {"sampler_step_kernel": "one fused importance-sampling step per launch (udf_renderer_blending.py:228-377)",
           "composite_kernel": "render_core tail: alpha, visibility, weights, per-ray sums (udf_renderer_blending.py:418-677)",
           "composite_reduce_kernel": "cross-ray reduction of the eikonal terms"}
KERNELS = {"render": {"udf_mlp_rev": "final value+grad MLP pass (reverse sweep)", "udf_mlp_fs2_kernel": "value passes of the sampler", **PER_RAY},
           "train": {"udf_mlp_vjp_kernel": "MLP double-backward sweep", "wgrad_kernel": "weight-gradient GEMMs",
                     "udf_mlp_rev": "final value+grad MLP pass (reverse sweep)", "composite_bwd_kernel": "adjoint of the render_core tail",
                     **PER_RAY}}[mode]


def find(pat):
    f = glob.glob(os.path.join(src, pat), recursive=True)
    return f[0] if f else None


ks = find(f"trace_{mode}_{prec}/**/*kernel_stats.csv")
if ks:
    rows = list(csv.DictReader(open(ks)))
    lines = [f"# rocprofv3 --kernel-trace --stats -- python bench.py --mode {mode} --steps 50 --warmup 10 --precision {prec} --no-cpu-baseline --no-other-modes --no-parity",
             f"# source: gpurun_out/prof_{tag}/trace_{mode}_{prec} (MI355X, 1 GPU); durations in ns", ",".join(rows[0].keys())]
    lines += [",".join(str(r[k]) for k in rows[0].keys()) for r in rows[:30]]
    open(os.path.join(dst, f"{tag}_kernel_stats_{mode}_{prec}.csv"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:14]))
bj = find(f"bench_under_rocprof_{mode}_{prec}.json")
if bj:
    open(os.path.join(dst, f"{tag}_bench_under_rocprof_{mode}_{prec}.json"), "w").write(open(bj).read())

out = {}
for d in glob.glob(os.path.join(src, f"pmc_{mode}_{prec}_*")):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        continue
    by = collections.defaultdict(lambda: collections.defaultdict(float))
    names = {}
    for r in csv.DictReader(open(f[0])):
        by[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        names[r["Dispatch_Id"]] = r["Kernel_Name"]
    for key, label in KERNELS.items():
        agg = collections.defaultdict(list)
        for did, cs in by.items():
            if key in names[did]:
                for c, v in cs.items():
                    agg[c].append(v)
        ent = out.setdefault(key, {"kernel": label, "counters": {}})
        for c, v in agg.items():
            ent["counters"][c] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
traffic = {}
for key, ent in out.items():
    c = ent["counters"]
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of wide coalesced
        # reads (MI355X_MICROARCH.md, HBM): doubled.  WRITE_SIZE is taken as reported (uncalibrated, says the guide).
        fetch = c["FETCH_SIZE"]["mean_per_launch"] * 1024 * 2
        write = c["WRITE_SIZE"]["mean_per_launch"] * 1024
        ent["hbm_bytes_per_launch"] = fetch + write
        ent["fetch_bytes_corrected_x2"] = fetch
        ent["write_bytes"] = write
        traffic[key] = fetch + write
if out:
    json.dump(out, open(os.path.join(dst, f"{tag}_pmc_{mode}_{prec}.json"), "w"), indent=1)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "counters"} for k, v in out.items()}, indent=1))
    tpath = os.path.join(dst, f"{tag}_traffic.json")
    tj = json.load(open(tpath)) if os.path.exists(tpath) else {}
    dom = "udf_mlp_vjp_kernel" if mode == "train" else "udf_mlp_rev"
    if dom in traffic:
        tj[f"{mode}:{prec}"] = {"hbm_bytes_per_launch": traffic[dom], "kernel": dom, "all": traffic}
        json.dump(tj, open(tpath, "w"), indent=1)
