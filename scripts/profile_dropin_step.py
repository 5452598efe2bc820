#!/usr/bin/env python3
"""Host-side profile of the drop-in training step (bench.train_dropin_key's step): where the time between the GPU kernels goes.
usage: python scripts/profile_dropin_step.py [--fused | --patched] > out.txt      (needs a GPU)"""
import cProfile
import io
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    fused = "--fused" in sys.argv
    patched = "--patched" in sys.argv
    dev = torch.device("cuda", 0)
    grabbed = {}
    orig = bench._timed

    def grab(step, steps, warmup):
        grabbed["step"] = step
        return orig(step, steps, warmup)

    bench._timed = grab
    r = bench.train_dropin_key(dev, "f16x3", 512, steps=20, warmup=10, fused_adam=fused, patched=patched)
    print("ms_per_step", r["ms_per_step"], "median", r["ms_per_step_median"])
    step = grabbed["step"]
    # (1) the GPU work of one step if the host never waited: enqueue 20 steps' worth is impossible (the step syncs), so time the
    #     kernels with a profiler-free estimate: total step time - time the host spends blocked in synchronisations
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pr.enable()
    for _ in range(50):
        step()
    pr.disable()
    torch.cuda.synchronize()
    print("profiled ms/step", (time.perf_counter() - t0) / 50 * 1e3)
    for key in ("cumulative", "tottime"):
        s = io.StringIO()
        pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
        print(s.getvalue()[:9000])


if __name__ == "__main__":
    main()
