"""The fast loop is generated code: the committed .inc must be what the generator produces, and
it must respect the gfx950 wait states the assembler does not insert inside inline asm
(tools/hazards.py) -- a missing one once made every stream decode wrong."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    spec = importlib.util.spec_from_file_location("gen_fastpath_under_test", os.path.join(ROOT, "tools", "gen_fastpath.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_committed_inc_is_generated_and_hazard_free():
    g = _load()
    text, final, n_nops = g.render()
    with open(os.path.join(ROOT, "lzma_amd", "csrc", "xlz_fastpath.inc")) as f:
        assert f.read() == text, "run python3 tools/gen_fastpath.py"
    import hazards
    assert hazards.analyse(final) == {}
    assert n_nops < 20  # instruction order, not s_nop, is what satisfies the wait states


def test_the_compact_loop_is_the_same_loop_over_other_table_bases():
    """xlz_fastpath_pb2.inc (round 5: the model layout with room for 4 posStates instead of 16, xlz_format.h:
    ModelLayout<true>) is `--variant compact` of the same generator: committed text in sync, hazard-free, and instruction
    for instruction the full loop with other numbers -- nothing but table bases may differ."""
    import re
    g = _load()
    g.VARIANT.add("compact")
    text, final, n_nops = g.render()
    with open(os.path.join(ROOT, "lzma_amd", "csrc", "xlz_fastpath_pb2.inc")) as f:
        assert f.read() == text, "run python3 tools/gen_fastpath.py --variant compact --out lzma_amd/csrc/xlz_fastpath_pb2.inc"
    import hazards
    assert hazards.analyse(final) == {}
    full = open(os.path.join(ROOT, "lzma_amd", "csrc", "xlz_fastpath.inc")).read()
    strip = lambda t: re.sub(r"\b(0x[0-9a-f]+|\d+)\b", "N", t)
    assert strip(full) == strip(text) and full != text
    # the third committed loop: the compact layout with BRANCHY decisions (launches of 24 workgroups per CU)
    g2 = _load()
    g2.VARIANT.update(("compact", "dbr", "dbrs"))
    text_br, final_br, _ = g2.render()
    with open(os.path.join(ROOT, "lzma_amd", "csrc", "xlz_fastpath_pb2_br.inc")) as f:
        assert f.read() == text_br, "run python3 tools/gen_fastpath.py --variant compact,dbr,dbrs --out lzma_amd/csrc/xlz_fastpath_pb2_br.inc"
    assert hazards.analyse(final_br) == {}
    lay = g.model_layout(True)
    assert (lay["P_IS_REP"], lay["P_IS_REP0_LONG"], lay["P_POS_SLOT"], lay["P_LEN"], lay["P_REP_LEN"], lay["P_LIT"]) == (48, 96, 144, 532, 856, 924)
    assert 2 * (lay["P_LIT"] + (0x100 << 3)) + 128 <= 5 * 1280     # lc+lp = 3: five LDS granules


def test_committed_layout_keeps_conditional_branches_in_lower_halves():
    """tools/layout.py: behind the alignment directive every conditional branch of the committed
    loop sits at address mod 16 = 0 or 4 -- checked with the assembler's own instruction sizes -- and the pass
    got there almost without executed padding."""
    g = _load()
    text, final, _ = g.render()
    import layout
    n, bad = layout.report(final)
    assert n > 250 and bad == 0
    start = next(i for i, l in enumerate(final) if l.startswith(".p2align"))
    assert sum(1 for l in final[start:] if l.endswith("_e64") or "_e64 " in l) > 20   # re-encoded, not padded
    # the pass itself, on a toy stream: one promotion moves the branch from offset 12 to 16
    toy = [".p2align 4", "s_mov_b32 s80, 1", "v_add_u32 v1, v2, v3", "s_mov_b32 s81, 2", "s_cbranch_scc1 .Lx_%=", ".Lx_%=:"]
    out, promos, nops, dead = layout.align_branches(toy)
    assert promos == 1 and nops == 0 and "v_add_u32_e64 v1, v2, v3" in out
    # nothing to re-encode: padding, behind an unconditional branch if there is one (never executed)
    toy = [".p2align 4", "s_mov_b32 s80, 1", "s_branch .Ly_%=", ".Ly_%=:", "s_mov_b32 s81, 2", "s_cbranch_scc1 .Lx_%=", ".Lx_%=:"]
    out, promos, nops, dead = layout.align_branches(toy)
    assert (promos, nops, dead) == (0, 0, 1) and out[3] == "s_nop 0"


def test_hazard_checker_sees_the_known_cases():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import hazards
    # R1: VALU write of a VGPR, v_readlane of it right after
    assert hazards.analyse(["v_mul_u32_u24 v55, v55, v40", "v_readlane_b32 s80, v55, 3"])
    assert not hazards.analyse(["v_mul_u32_u24 v55, v55, v40", "s_lshr_b32 s81, s80, 24", "v_readlane_b32 s80, v55, 3"])
    # R2: SGPR written by v_readlane, read by a VALU instruction one slot later
    assert hazards.analyse(["v_readlane_b32 s80, v55, 3", "s_sub_u32 s81, s82, s80", "v_subrev_co_u32 v28, vcc, s80, v29"])
    assert not hazards.analyse(["v_readlane_b32 s80, v55, 3", "s_sub_u32 s81, s82, s80", "s_nop 0",
                                "v_subrev_co_u32 v28, vcc, s80, v29"])
    # R3: VCC written by v_cmp, read by v_cndmask
    assert hazards.analyse(["v_cmp_eq_u32 vcc, 0, v1", "v_cndmask_b32 v2, v3, v4, vcc"])
    assert not hazards.analyse(["v_cmp_eq_u32 vcc, 0, v1", "s_nop 1", "v_cndmask_b32 v2, v3, v4, vcc"])
    # R7: a DPP instruction reads (as src0) a VGPR the VALU wrote one slot earlier
    dpp = "v_sub_u32_dpp v40, v40, v63 quad_perm:[0,1,2,3] row_mask:0x1 bank_mask:0x2"
    assert hazards.analyse(["v_add_u32 v40, v1, v2", "s_mov_b32 s80, 1", dpp])
    assert not hazards.analyse(["v_add_u32 v40, v1, v2", "s_mov_b32 s80, 1", "s_mov_b32 s81, 1", dpp])
    assert not hazards.analyse(["v_add_u32 v63, v1, v2", dpp])   # src1 is read without DPP: no extra wait state
    # an assembler directive between producer and consumer is not a wait state
    assert hazards.analyse(["v_mul_u32_u24 v55, v55, v40", ".p2align 4", "v_readlane_b32 s80, v55, 3"])
    # across a branch: the consumer is the first instruction at the target
    bad = ["v_readlane_b32 s80, v55, 3", "s_cbranch_scc0 .Lx", "s_nop 3", ".Lx:", "v_add_u32 v1, s80, v1"]
    assert hazards.analyse(bad)
    fixed, n = hazards.fix(bad)
    assert n == 1 and not hazards.analyse(fixed)
