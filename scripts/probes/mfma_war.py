#!/usr/bin/env python3
"""Per kernel: asm VMEM loads whose destination overlaps an operand of an MFMA issued within the last W instructions, split by
operand kind (A/B inputs vs C input / D output).  usage: mfma_war.py file.s [W]"""
import re, sys, collections
RE_RANGE = re.compile(r"v\[(\d+):(\d+)\]")
def rr(tok):
    m = RE_RANGE.search(tok)
    return set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else set()
path = sys.argv[1]; W = int(sys.argv[2]) if len(sys.argv) > 2 else 12
func = None; recent = []; idx = 0; in_asm = False
stats = collections.defaultdict(lambda: collections.Counter())
for line in open(path):
    if line.lstrip().startswith(";;#ASMSTART"): in_asm = True; continue
    if line.lstrip().startswith(";;#ASMEND"): in_asm = False; continue
    s = line.split(";", 1)[0].strip()
    if not s or s.startswith("."): continue
    if s.endswith(":"):
        func = s[:-1]; recent = []; continue
    idx += 1
    if s.startswith("v_mfma"):
        ops = [o.strip() for o in s.split(None, 1)[1].split(",")]
        recent.append((idx, rr(ops[0]), rr(ops[1]) | rr(ops[2]), rr(ops[3]) if len(ops) > 3 else set()))
        recent = recent[-64:]
        continue
    if in_asm and s.startswith("global_load"):
        dst = rr(s.split(None, 1)[1].split(",")[0])
        for (i, d, ab, c) in recent:
            dist = idx - i
            if dist <= W:
                if dst & ab: stats[func]["AB<=%d" % W] += 1
                if dst & (c | d): stats[func]["CD<=%d" % W] += 1
        stats[func]["loads"] += 1
for f, c in stats.items():
    if "rev_kernel" in f or "fs2_kernel" in f:
        print(f[:64].ljust(64), dict(c))
