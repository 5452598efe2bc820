#!/usr/bin/env python3
"""profiles/r03_pmc_sampler.json: HBM-side bytes, GB/s and fraction of the 8 TB/s peak of the per-ray kernels (sampler_step_kernel,
composite_kernel, composite_bwd_kernel) at 512 and 4096 rays, from the rocprofv3 passes of scripts/profile_round.sh
(gpurun_out/profiles_<tag>/<tag>_pmc_<mode>_f16x3.json + <tag>_kernel_stats_<mode>_f16x3.csv).
usage: summarize_sampler_pmc.py <out.json> <label>=<tag>:<mode> ...   e.g. 512=r03:train 4096=r03x4096:train"""
import csv, json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {"_about": "per-ray kernels: algorithmic bytes per ray-sample (DESIGN 3.2) vs HBM-side traffic from rocprofv3 --pmc (FETCH_SIZE x2 for the "
                 "gfx950 under-count + WRITE_SIZE as reported, MI355X_MICROARCH.md), durations from rocprofv3 --kernel-trace --stats of the same "
                 "command; peak = 8 TB/s.  These kernels are launch/latency-bound (512 - 4096 waves, one per ray): the fraction of the HBM "
                 "peak is reported because SURVEY 8(d) asks for it, not because they are bandwidth-bound."}
for spec in sys.argv[2:]:
    label, rest = spec.split("=")
    tag, mode = rest.split(":")
    d = os.path.join(root, "gpurun_out", f"profiles_{tag}")
    pmc = json.load(open(os.path.join(d, f"{tag}_pmc_{mode}_f16x3.json")))
    dur = {}
    for l in open(os.path.join(d, f"{tag}_kernel_stats_{mode}_f16x3.csv")):
        if l.startswith("#") or l.startswith("Name,"):
            continue
        f = l.rstrip("\n").rsplit(",", 7)          # kernel names contain commas: Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs,StdDev
        dur[f[0]] = (float(f[3]), int(f[1]))
    ent = {}
    for key in ("sampler_step_kernel", "composite_kernel", "composite_bwd_kernel", "composite_reduce_kernel"):
        if key not in pmc or "hbm_bytes_per_launch" not in pmc[key]:
            continue
        names = [n for n in dur if key in n and (key != "composite_kernel" or "reduce" not in n and "bwd" not in n)]
        tot = sum(dur[n][0] * dur[n][1] for n in names)
        calls = sum(dur[n][1] for n in names)
        avg_ns = tot / max(calls, 1)
        b = pmc[key]["hbm_bytes_per_launch"]
        c = pmc[key]["counters"]
        ent[key] = {"what": pmc[key]["kernel"], "avg_duration_us": avg_ns / 1e3, "hbm_bytes_per_launch": b, "fetch_bytes_x2": pmc[key]["fetch_bytes_corrected_x2"],
                    "write_bytes": pmc[key]["write_bytes"], "GB_per_s": b / avg_ns, "frac_of_8TBps": b / avg_ns / 8000.0,
                    "SQ_WAIT_ANY_frac_of_wave_cycles": (c["SQ_WAIT_ANY"]["mean_per_launch"] / c["SQ_WAVE_CYCLES"]["mean_per_launch"]) if "SQ_WAIT_ANY" in c else None,
                    "waves": c.get("SQ_WAVES", {}).get("mean_per_launch")}
    out[f"{label} rays ({mode} step)"] = ent
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
