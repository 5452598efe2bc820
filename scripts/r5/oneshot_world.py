#!/usr/bin/env python3
"""N processes sharing GPU 0 run the one-shot all-reduce (tests/test_gpu_multi.py:_rank_oneshot) - does R = 4 / 8 work on one device?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch.multiprocessing as mp
import test_gpu_multi as T
if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ctx = mp.get_context("spawn"); q = ctx.Queue(); port = T._free_port()
    t0 = time.time()
    ps = [ctx.Process(target=T._rank_oneshot, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = sorted(q.get(timeout=400) for _ in range(world))
    [p.join(timeout=120) for p in ps]
    print("world", world, "seconds", round(time.time() - t0, 1), "exit codes", [p.exitcode for p in ps]); print(res)
