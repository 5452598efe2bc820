#!/usr/bin/env python3
"""ISA lint for the software-pipelined inline-asm loads of the MLP kernels.

The kernels issue `global_load_dwordx4` from inline asm and retire them with counted `s_waitcnt vmcnt(N)` statements that
name the destination registers.  The compiler believes an asm output is valid as soon as the load statement has executed,
so nothing but register coalescing keeps it from copying (or spilling) such a register before the matching wait - which
reads data that has not arrived (observed once: a v_mov of in-flight bias registers above the wait; wrong results only
under load).  This script re-derives the in-flight set from the generated gfx950 assembly and fails the build if any
instruction outside the asm statements reads or writes a VGPR that an asm load still has in flight.

Model (straight-line, conservative): every asm `global_load` pushes its destination range on a FIFO; every
`s_waitcnt vmcnt(N)` (asm or compiler) pops until N entries are left (loads return in order; compiler-issued memory
operations are ignored, which only makes the hardware retire *more* than the model assumes).  The K-loops are straight-line
code that ends with vmcnt(0), so the FIFO is empty at every loop back-edge and join.

Second rule (write-after-read against the matrix pipe): an asm load may reuse a register that an MFMA issued just before it
still reads as an operand (the compiler frees a fragment register after its last MFMA and the next prefetch lands in it).  The
load's data cannot come back before the MFMA has read its sources (a vector-memory round trip is >= 100 cycles, the operand
reads happen in the MFMA's first passes), but the rule is made explicit instead of being left to timing: such a load must be
separated from that MFMA by >= 5 wait states.  Every asm load statement of the kernels opens with `s_nop 4` (needed anyway
for the VALU-write-SGPR -> VMEM-address hazard), which satisfies it; the lint checks that no overwriting load comes closer.

usage: isa_lint.py file.s [file.s ...]      exit status 1 on a violation
"""
import re
import sys

RE_RANGE = re.compile(r"\bv\[(\d+):(\d+)\]")
RE_SINGLE = re.compile(r"\bv(\d+)\b")
RE_VMCNT = re.compile(r"s_waitcnt\b.*vmcnt\((\d+)\)")
RE_LOAD = re.compile(r"^\s*global_load_dwordx[234]\s+v\[(\d+):(\d+)\]")
RE_LOAD1 = re.compile(r"^\s*global_load_dword\s+v(\d+)\b")      # one register (round 5: the E8M0 scale words of the MX sweep)
RE_NOP = re.compile(r"^\s*s_nop\s+(\d+)")
WAR_WINDOW = 2        # MFMAs this many instructions (or fewer) before an asm load are checked
WAR_WAIT_STATES = 5   # required between such an MFMA and the load that overwrites one of its source registers


def regs_of(text):
    out = set()
    for a, b in RE_RANGE.findall(text):
        out.update(range(int(a), int(b) + 1))
    for a in RE_SINGLE.findall(RE_RANGE.sub("", text)):
        out.add(int(a))
    return out


def lint(path):
    violations = []
    fifo = []          # list of (set(regs), line_no)
    recent = []        # the last WAR_WINDOW non-asm instructions: (is_mfma, source registers, wait states it contributes)
    nops = 0           # wait states accumulated inside the current asm statement before its load
    in_asm = False
    func = None
    n_loads = 0
    with open(path) as f:
        for no, line in enumerate(f, 1):
            code = line.split(";", 1)[0] if not line.lstrip().startswith(";;#") else line
            s = code.strip()
            if not s:
                continue
            if s.startswith(";;#ASMSTART"):
                in_asm = True
                nops = 0
                continue
            if s.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if s.endswith(":") and not s.startswith("."):
                func = s[:-1]
                fifo = []
                continue
            if s.startswith("."):          # labels / directives
                continue
            if s.startswith("s_endpgm"):
                fifo = []
                continue
            m = RE_VMCNT.search(s)
            if m:
                n = int(m.group(1))
                while len(fifo) > n:
                    fifo.pop(0)
                continue
            if s.startswith("s_waitcnt") and "vmcnt" not in s and "lgkmcnt" not in s and "expcnt" not in s:
                fifo = []                  # bare "s_waitcnt 0"-style encodings: everything retired
                continue
            if in_asm:
                mn = RE_NOP.match(s)
                if mn:
                    nops += int(mn.group(1)) + 1
                m = RE_LOAD.match(s)
                m1 = RE_LOAD1.match(s) if not m else None
                if m or m1:
                    dst = set(range(int(m.group(1)), int(m.group(2)) + 1)) if m else {int(m1.group(1))}
                    m = m or m1
                    between = nops
                    for is_mfma, src, ws in reversed(recent):
                        if is_mfma and (src & dst) and between < WAR_WAIT_STATES:
                            violations.append((func, no, s + f"   <- overwrites an operand of an MFMA {between} wait states earlier", sorted(src & dst)))
                        between += ws
                    nops += 1
                    rest = s[m.end():]
                    busy = set().union(*[r for r, _ in fifo]) if fifo else set()
                    bad = (regs_of(rest) | dst) & busy
                    if bad:
                        violations.append((func, no, s, sorted(bad)))
                    fifo.append((dst, no))
                    n_loads += 1
                continue
            is_mfma = s.startswith("v_mfma")
            mn = RE_NOP.match(s)
            src = set()
            if is_mfma:
                ops = s.split(None, 1)[1].split(",") if len(s.split(None, 1)) > 1 else []
                src = regs_of(",".join(ops[1:]))          # everything but the destination
            recent.append((is_mfma, src, int(mn.group(1)) + 1 if mn else 1))
            recent = recent[-WAR_WINDOW:]
            if not fifo:
                continue
            busy = set().union(*[r for r, _ in fifo])
            bad = regs_of(s) & busy
            if bad:
                violations.append((func, no, s, sorted(bad)))
    return violations, n_loads


def main(paths):
    total = 0
    rc = 0
    for p in paths:
        v, n = lint(p)
        total += n
        for func, no, s, bad in v[:20]:
            print(f"{p}:{no}: [{func}] touches in-flight v{bad}: {s}")
        if len(v) > 20:
            print(f"{p}: ... {len(v) - 20} more")
        if v:
            rc = 1
    print(f"isa_lint: {total} asm loads checked in {len(paths)} file(s): {'VIOLATIONS' if rc else 'clean'}")
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
