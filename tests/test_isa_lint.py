"""scripts/isa_lint.py: the build-time check of the inline-asm load pipelines (DESIGN.md par. 3.1).  CPU-only."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("isa_lint", os.path.join(ROOT, "scripts", "isa_lint.py"))
isa_lint = importlib.util.module_from_spec(spec)
spec.loader.exec_module(isa_lint)

HEAD = "_Zkernel:\n"
LOAD = "\t;;#ASMSTART\n\ts_nop 4\n\tglobal_load_dwordx4 v[{a}:{b}], v1, s[2:3]\n\t;;#ASMEND\n"
WAIT = "\t;;#ASMSTART\n\ts_waitcnt vmcnt({n})\n\t;;#ASMEND\n"


def run(tmp_path, body):
    p = tmp_path / "k.s"
    p.write_text(HEAD + body + "\ts_endpgm\n")
    return isa_lint.lint(str(p))


def test_clean_pipeline(tmp_path):
    body = (LOAD.format(a=10, b=13) + LOAD.format(a=14, b=17) + "\tds_read_b128 v[20:23], v5\n" + WAIT.format(n=1) +
            "\tv_mfma_f32_16x16x32_f16 v[30:33], v[10:13], v[20:23], v[30:33]\n" + WAIT.format(n=0) +
            "\tv_mfma_f32_16x16x32_f16 v[30:33], v[14:17], v[20:23], v[30:33]\n")
    v, n = run(tmp_path, body)
    assert n == 2 and v == []


def test_copy_above_the_wait_is_flagged(tmp_path):
    # the bug this lint exists for: a register copy of an asm-loaded value scheduled above its s_waitcnt
    body = (LOAD.format(a=34, b=37) + LOAD.format(a=10, b=13) + "\tv_mov_b64_e32 v[108:109], v[36:37]\n" + WAIT.format(n=0) +
            "\tv_mfma_f32_16x16x32_f16 v[30:33], v[10:13], v[20:23], v[106:109]\n")
    v, n = run(tmp_path, body)
    assert n == 2 and len(v) == 1 and v[0][3] == [36, 37]


def test_counted_wait_retires_in_order_only(tmp_path):
    # vmcnt(1) retires the older load only: using the younger one is a violation, using the older is fine
    body = (LOAD.format(a=10, b=13) + LOAD.format(a=14, b=17) + WAIT.format(n=1) +
            "\tv_add_f32_e32 v40, v10, v11\n" + "\tv_add_f32_e32 v41, v14, v15\n" + WAIT.format(n=0))
    v, _ = run(tmp_path, body)
    assert len(v) == 1 and v[0][3] == [14, 15]


def test_overwriting_an_in_flight_register_is_flagged(tmp_path):
    body = LOAD.format(a=10, b=13) + "\tds_read_b128 v[12:15], v5\n" + WAIT.format(n=0)
    v, _ = run(tmp_path, body)
    assert len(v) == 1 and v[0][3] == [12, 13]


def test_compiler_waitcnt_also_retires(tmp_path):
    body = LOAD.format(a=10, b=13) + "\ts_waitcnt vmcnt(0) lgkmcnt(0)\n" + "\tv_add_f32_e32 v40, v10, v11\n"
    v, _ = run(tmp_path, body)
    assert v == []


def test_load_that_overwrites_an_mfma_operand_needs_wait_states(tmp_path):
    # write-after-read against the matrix pipe: a prefetch reusing the fragment register of the MFMA just issued must be
    # >= 5 wait states behind it (the `s_nop 4` every asm load statement opens with); a bare load right behind it is flagged
    mfma = "\tv_mfma_f32_16x16x32_f16 v[30:33], v[10:13], v[20:23], v[30:33]\n"
    bare = "\t;;#ASMSTART\n\tglobal_load_dwordx4 v[10:13], v1, s[2:3]\n\t;;#ASMEND\n"
    v, _ = run(tmp_path, mfma + bare + WAIT.format(n=0))
    assert len(v) == 1 and v[0][3] == [10, 11, 12, 13]
    v, _ = run(tmp_path, mfma + LOAD.format(a=10, b=13) + WAIT.format(n=0))          # with the s_nop 4: fine
    assert v == []
    v, _ = run(tmp_path, mfma + "\ts_nop 5\n" + bare + WAIT.format(n=0))              # wait states outside the statement count too
    assert v == []
    v, _ = run(tmp_path, mfma + bare.replace("v[10:13]", "v[40:43]") + WAIT.format(n=0))   # unrelated destination: fine
    assert v == []
