#!/usr/bin/env python3
"""Scan gfx950 assembly for LDS/VMEM loads (or VALU writes) whose destination overlaps the destination of an MFMA issued at most
WINDOW instructions earlier with no read of that register in between (a write-after-write against a possibly still executing
MFMA), and for loads whose destination overlaps a SOURCE of a recent MFMA (write-after-read).  usage: mfma_waw.py file.s [window]"""
import re, sys
RE_RANGE = re.compile(r"\bv\[(\d+):(\d+)\]")
RE_SINGLE = re.compile(r"\bv(\d+)\b")
def regs(text):
    out = set()
    for a, b in RE_RANGE.findall(text):
        out.update(range(int(a), int(b) + 1))
    for a in RE_SINGLE.findall(RE_RANGE.sub("", text)):
        out.add(int(a))
    return out
def operands(s):
    body = s.split(None, 1)[1] if " " in s.strip() or "\t" in s.strip() else ""
    parts = [p.strip() for p in body.split(",")]
    return parts
path = sys.argv[1]; W = int(sys.argv[2]) if len(sys.argv) > 2 else 24
func = None; recent = []  # (idx, dst regs, src regs, line)
idx = 0; waw = war = 0; shown = 0
for no, line in enumerate(open(path), 1):
    s = line.split(";", 1)[0].strip()
    if not s or s.startswith("."): continue
    if s.endswith(":"):
        func = s[:-1]; recent = []; continue
    idx += 1
    ops = operands(s)
    if s.startswith("v_mfma"):
        recent.append((idx, regs(ops[0]), regs(",".join(ops[1:3])), no))
        recent = [r for r in recent if idx - r[0] <= W]
        continue
    is_load = s.startswith("ds_read") or s.startswith("global_load") or s.startswith("scratch_load") or s.startswith("buffer_load")
    if is_load and ops:
        dst = regs(ops[0])
        for (i, d, srcs, ln) in recent:
            if idx - i <= W:
                if dst & d:
                    waw += 1
                    if shown < 12 and "rev_kernel" in (func or ""): print(f"WAW {func[:40]} line {no}: {s[:60]}  vs MFMA at line {ln} (distance {idx - i})"); shown += 1
                if dst & srcs:
                    war += 1
                    if shown < 12 and "rev_kernel" in (func or "") and idx - i <= 4: print(f"WAR {func[:40]} line {no}: {s[:60]}  vs MFMA at line {ln} (distance {idx - i})"); shown += 1
    # a read of an MFMA destination retires it from the WAW watch
    rd = regs(",".join(ops[1:])) if len(ops) > 1 else set()
    recent = [(i, d - rd, sr, ln) for (i, d, sr, ln) in recent if idx - i <= W]
print(f"{path}: loads overlapping a recent MFMA destination (WAW, window {W}): {waw}; overlapping a recent MFMA source (WAR): {war}")
