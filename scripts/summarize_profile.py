#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (rocprofv3 csv output) into the small committed summaries under profiles/."""
import csv, collections, glob, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
prec = sys.argv[2] if len(sys.argv) > 2 else "f16x3"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)

def find(pat):
    f = glob.glob(os.path.join(src, pat), recursive=True)
    return f[0] if f else None

# 1. kernel stats
ks = find(f"trace_{prec}/**/*kernel_stats.csv")
lines = []
if ks:
    rows = list(csv.DictReader(open(ks)))
    lines.append(f"# rocprofv3 --kernel-trace --stats -- python bench.py --steps 50 --warmup 10 --precision {prec} --no-cpu-baseline --no-other-modes")
    lines.append(f"# source: gpurun_out/prof_{tag}/trace_{prec} (MI355X, 1 GPU); durations in ns")
    lines.append(",".join(rows[0].keys()))
    for r in rows[:25]:
        lines.append(",".join(str(r[k]) for k in rows[0].keys()))
    open(os.path.join(dst, f"{tag}_kernel_stats_{prec}.csv"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:12]))
bj = find(f"bench_under_rocprof_{prec}.json")
if bj:
    open(os.path.join(dst, f"{tag}_bench_under_rocprof_{prec}.json"), "w").write(open(bj).read())

# 2. PMC passes -> per-launch numbers for the dominant kernel (the grad MLP kernel)
summary = {}
for d in glob.glob(os.path.join(src, f"pmc_{prec}_*")):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        continue
    by = collections.defaultdict(lambda: collections.defaultdict(float))
    names = {}
    for r in csv.DictReader(open(f[0])):
        by[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        names[r["Dispatch_Id"]] = r["Kernel_Name"]
    agg = collections.defaultdict(list)
    for did, cs in by.items():
        kn = names[did]
        if "udf_mlp_rev_kernel" in kn or ("udf_mlp" in kn and "true" in kn.split("udf_mlp")[1][:48]):
            for c, v in cs.items():
                agg[c].append(v)
    for c, v in agg.items():
        summary[c] = {"mean_per_launch": sum(v) / len(v), "launches": len(v)}
if summary:
    out = {"kernel": f"final value+grad MLP pass ({prec}), 512 rays x 128 samples", "counters": summary}
    if "FETCH_SIZE" in summary and "WRITE_SIZE" in summary:
        # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of wide
        # coalesced reads (MI355X_MICROARCH.md, HBM): double it.  WRITE_SIZE is taken as reported.
        fetch = summary["FETCH_SIZE"]["mean_per_launch"] * 1024 * 2
        write = summary["WRITE_SIZE"]["mean_per_launch"] * 1024
        out["hbm_bytes_per_launch"] = fetch + write
        out["fetch_bytes_corrected_x2"] = fetch
        out["write_bytes"] = write
    json.dump(out, open(os.path.join(dst, f"{tag}_pmc_{prec}.json"), "w"), indent=1)
    print(json.dumps(out, indent=1)[:1500])
    tpath = os.path.join(dst, "r01_traffic.json") if tag == "r01" else os.path.join(dst, f"{tag}_traffic.json")
    tj = json.load(open(tpath)) if os.path.exists(tpath) else {}
    if "hbm_bytes_per_launch" in out:
        tj[prec] = {"hbm_bytes_per_launch": out["hbm_bytes_per_launch"]}
        json.dump(tj, open(tpath, "w"), indent=1)
