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


def test_hot_kernels_of_the_built_library_do_not_spill_and_the_compositing_adjoint_has_no_atomics():
    """Structural properties of the ISA that build.sh leaves next to the library (emap_amd/lib/isa/*.s): no kernel of the hot path spills a
    vector register (metadata .vgpr_spill_count 0 and not one scratch_ instruction in the unit: a spill in one of them once cost 30 %; a few
    kernels keep a dead 36-byte frame, which is not a spill), and composite_bwd_kernel carries no global atomic - two atomicMax per ray on
    one cache line cost 10.5 ns EACH, serialised (round 5: 85 of the kernel's 100 us at 4096 rays)."""
    import glob
    import re
    import pytest
    files = glob.glob(os.path.join(ROOT, "emap_amd", "lib", "isa", "*gfx950*.s"))
    if not files:
        pytest.skip("library not built here (emap_amd/csrc/build.sh writes emap_amd/lib/isa)")
    hot = ("udf_mlp_rev32_kernel", "udf_mlp_fs2_kernel", "udf_mlp_vjp_kernel", "wgrad_kernel", "composite_kernel", "composite_bwd_kernel",
           "sampler_step_kernel", "adam_kernel", "pack_all_kernel")
    seen = set()
    for f in files:
        text = open(f).read()
        split_fp16_unit = "udf_mlp_f16x3" in os.path.basename(f)
        md = text[text.index("amdhsa.kernels:"):] if "amdhsa.kernels:" in text else ""
        for blk in md.split("  - .agpr_count")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            key = next((h for h in hot if h in name), None)
            if key is None or (key.startswith("udf_mlp") and not split_fp16_unit):
                continue          # MLP kernels: the split-fp16 unit (the default mode and its variants)
            seen.add(key)
            assert int(re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1)) == 0, name
        if split_fp16_unit or "sampler" in os.path.basename(f) or "wgrad" in os.path.basename(f):
            assert not re.search(r"^\s*scratch_(load|store)", text, flags=re.M), os.path.basename(f)
        for m in re.finditer(r"^(_ZN4emap20composite_bwd_kernel\w+):[^\n]*\n(.*?)\n\s*s_endpgm", text, flags=re.S | re.M):
            assert not re.search(r"\b(global|flat|buffer)_atomic", m.group(2)), m.group(1)
            seen.add("composite_bwd_body")
    assert seen >= set(hot) | {"composite_bwd_body"}, sorted(set(hot) - seen)
