#!/bin/bash
# Run ON THE GPU BOX: L1 <-> L2 request counters of the forward step's kernels (what the value passes move between L2 and the CUs).
# Usage: scripts/profile_l2.sh <tag>
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o -E "\b(TCP|TCC|TA|TD)_[A-Z0-9_a-z]+" | sort -u > $OUT/avail_tcp_tcc.txt
PARGS="--mode render --steps 5 --warmup 2 --settle-steps 5 --no-cpu-baseline --no-other-modes --no-parity --no-train-key --traffic off"
for set in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_READ_sum" "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"; do
  n=$(echo $set | cut -d" " -f1)
  rocprofv3 --pmc $set --output-format csv -d $OUT/l2_$n -o p -- python $R/bench.py $PARGS > /dev/null 2> $OUT/l2_$n.err
done
python - "$OUT" <<'PY'
import collections, csv, glob, json, os, sys
out = collections.defaultdict(lambda: collections.defaultdict(list))
for d in glob.glob(os.path.join(sys.argv[1], "l2_*")):
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        by = collections.defaultdict(lambda: collections.defaultdict(float)); names = {}
        for r in csv.DictReader(open(f)):
            by[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"]); names[r["Dispatch_Id"]] = r["Kernel_Name"]
        for did, cs in by.items():
            k = names[did].split("(")[0].replace("void emap::", "").replace("emap::", "")
            for c, v in cs.items():
                out[k][c].append(v)
res = {k: {c: {"mean_per_launch": sum(v) / len(v), "launches": len(v)} for c, v in cs.items()} for k, cs in out.items() if "udf_mlp" in k or "sampler" in k or "composite" in k}
json.dump(res, open(os.path.join(sys.argv[1], "l2_summary.json"), "w"), indent=1)
for k, cs in res.items():
    print(k, {c: round(v["mean_per_launch"]) for c, v in cs.items()})
PY
