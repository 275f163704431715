#!/usr/bin/env python3
"""Generates lzma_amd/csrc/xlz_fastpath.inc: the hand-scheduled gfx950 fast loop of the
LZMA packet decoder as ONE inline-asm statement (run from the repo root).

Why generated: ~60 decision sites each need the same scalar core, an out-of-line
normalisation stub with its own labels, and the bit trees are unrolled.  Writing that by
hand invites typos; this script is the source, the .inc file is committed next to it.

The loop decodes whole packets (decompress.go:13 ff.) while
    arel <= arel_lim   (>= 32 readable bytes left in the 256-byte input window)
    pos  <  pos_lim    (>= 336 bytes of output room and of bytesLeft; tested where window.pos is
                        advanced, together with the dictionary wrap: reaching it makes the NEXT packet
                        head leave through the input test, see wpos_advance / packet_limits)
and leaves with an exit code:
    0  a limit was reached at a packet boundary (caller refills the window / re-checks)
    1  ErrResultError condition (bad distance, rep match on an empty window)
    2  distance 0xFFFFFFFF decoded (end marker): caller applies decompress.go:633-645
    3  a match copy the loop does not do itself (len >= 64, overlapping, or reaching in
       front of the dictionary epoch): caller copies `lenout` bytes and carries on
Every probability is read once and written once per packet, in the reference's order of
decisions; normalisation order is irrelevant here because no input exhaustion is possible
inside the loop.

How the 64 lanes are used (the wave is the register file of ONE decoder):
  * head gather: one ds_read with per-lane addresses fetches the ten context-selected
    probabilities a packet can start with (isMatch, isRep, isRepG0-2, isRep0Long, the four
    length `choice` bits) into v40, each alone in a DPP cell (lane 16 (j / 4) + 4 (j % 4)); a
    decision takes its bound with v_readlane from the VALU product of all of them, the update
    is written with row_mask / bank_mask.
  * tree blocks: one ds_read fetches 64 consecutive probabilities of a bit tree (lane j =
    slot 64b + j).  A tree level is then a handful of scalar instructions plus one
    `v_readlane`: the lane select IS the node index, so neither the decoded bit nor an LDS
    address is ever formed.  Blocks are requested well before they are walked.
  * model updates need no record of the walk: single-block trees find the visited slots again in
    the block register (tree_update, every lane looks at its own slot); the 8-level trees read the
    eight probabilities back from LDS with one gather (rec_gather_issue / tree_update_rec).
  * match copy: one byte per lane; its completion (store + prevByte/matchByte) is deferred
    behind the next packet's decode.
After generation tools/hazards.py inserts the wait states the assembler does not, and tools/layout.py
places every conditional branch in the lower half of a 16-byte block (measured: DESIGN.md 3.2).

Register conventions (fixed temporaries, declared as clobbers in xlz_kernel.hip):
  s80 bound, s81 core temps   s82..s86 temps   s88 tree slot (1, then !bits)
  s89 LEN   s90 posState   s91 state2   s92 table base (bytes)   s93 dist
  s95 length of the pending copy (0: none)   s96 next event of window.pos (wrap or output limit)
  s98 posSlot / nbits   s99 input limit (-1 once the output limit is reached)
  v13 v_perm selector   v14..v16 tree_update shift counts   v18,v19 tree_update_rec shift counts
  v20..v25 lane tables   v28 code - bound   v29 code   v30,v31 tree_update lane constants
  v35 align block   v36 posSlot block   v37 posDecoders block   v40 head probabilities
  v41,v42 len low/mid blocks   v43..v46 len high blocks   v47 head addresses
  v48,v49 pending copy (destination, bytes)   v50..v53 literal blocks   v54 gathered walk
  v55 temp   v56 2*lane   v57 address temp   v58 walk base   v59 gather address
  v60..v63 temps
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

# probability-table layout: must match xlz_format.h (ModelLayout<COMPACT>; checked by static_asserts in the .hip).
# Two layouts (round 5): the FULL one has room for 16 posStates in every table that is indexed by one (pb <= 4,
# types.go:15); the COMPACT one for 4 (pb <= 2 -- liblzma's and 7-Zip's default is pb = 2): the 7416-byte model of
# lc+lp = 3 becomes 5944 bytes = five of gfx950's LDS granules instead of six, and 24 workgroups fit a CU instead of 21.
# Only the table bases differ: a posState still selects eight consecutive probabilities of a length tree, and the
# (state, posState) pairs of isMatch / isRep0Long are addressed through per-lane constants that xlz_kernel.hip builds
# (head_vectors).  `--variant compact` renders the compact loop (xlz_fastpath_pb2.inc).
REP_HIGH_BYTES = 512  # the rep-length coder's high tree: the first 256 entries of the model's HBM part (xlz_format.h: kRepHigh)
LEN_CHOICE, LEN_CHOICE2, LEN_LOW = 0, 1, 4


def model_layout(compact):
    """-> dict of the table bases (units: 16-bit probabilities), xlz_format.h: ModelLayout<compact>"""
    ps = 4 if compact else 16                     # posStates a table has room for
    d = {"P_IS_MATCH": 0, "P_IS_REP": 12 * ps}
    d["P_IS_REP_G0"], d["P_IS_REP_G1"], d["P_IS_REP_G2"] = d["P_IS_REP"] + 12, d["P_IS_REP"] + 24, d["P_IS_REP"] + 36
    d["P_IS_REP0_LONG"] = d["P_IS_REP"] + 48
    d["P_POS_SLOT"] = d["P_IS_REP0_LONG"] + 12 * ps
    d["P_POS_DEC"] = d["P_POS_SLOT"] + 256
    d["P_ALIGN"] = d["P_POS_DEC"] + 116
    d["P_LEN"] = d["P_ALIGN"] + 16
    d["LEN_MID"] = LEN_LOW + 8 * ps
    d["LEN_HIGH"] = LEN_LOW + 16 * ps
    d["P_REP_LEN"] = d["P_LEN"] + d["LEN_HIGH"] + 256
    d["P_LIT"] = d["P_REP_LEN"] + d["LEN_HIGH"]   # (the rep-length coder's high tree is not in LDS)
    d["POS_STATES"] = ps
    return d


def set_layout(compact):
    globals().update(model_layout(compact))


set_layout(False)
assert (P_IS_REP, P_IS_REP0_LONG, P_POS_SLOT, P_POS_DEC, P_ALIGN, P_LEN, LEN_MID, LEN_HIGH, P_REP_LEN, P_LIT) == \
    (192, 240, 432, 688, 804, 820, 132, 260, 1336, 1596)

# head gather lanes (the per-lane address constants are built in xlz_kernel.hip: head_vectors)
H_IS_MATCH, H_IS_REP, H_G0, H_G1, H_G2, H_REP0_LONG, H_LEN_C, H_LEN_C2, H_REP_C, H_REP_C2 = range(10)

# Code paths that can be switched from the command line for A/B builds (--variant a,b adds, --without a,b removes).
# The defaults are what round 2 measured as best:
#   lgather  literal walks record nothing; the eight probabilities are read back with one LDS gather
#   hdpp     head probabilities sit alone in DPP cells: their update needs no lane compare / select
#   flim     the output limit is folded into the window-wrap test: one limit test per packet head
#   cflag    one SGPR says "copy pending" and "literal blocks not requested yet"
#   tuc      tree_update's per-lane shift counts are loop constants
#   rlhoist  literal walk: the next level's probability is read in front of the normalisation branch
#   vperm    normalisation: code = code << 8 | byte as one v_perm_b32 (no scalar mask of the byte)
#   bralign  tools/layout.py: conditional branches in the lower half of a 16-byte block; stub32 / head32 /
#            pktl64: normalisation stubs, out-of-line blocks and the literal loop on 32 / 32 / 64-byte boundaries
# count-only changes prepared in round 2, checked on the emulator (tests/test_fastpath_emulated.py, tools/fuzz_emulated.py);
# the first five measured once on the hardware (+6.6 % on incompressible data, +0.9 % on text, profiles/r02/layout_scan.md):
# `--variant next` switches all of them on
NEXT_VARIANT = {"slot0", "vprev", "rmov", "nopos", "l7blk", "warel", "vreps"}
# round 3 (profiles/r03/ab_wsb.txt, ab_latency_variants.txt): wsb adopted (+1.0 % text, +0.4 % cfg3 shape); measured and NOT
# adopted: dbr / dbrw (branchy decisions: -25 % scalar instructions, -7 % on incompressible data: a taken branch costs more
# than the two scalar instructions it saves), scode (the literal's decisions entirely scalar: +1 scalar instruction per level,
# -6 % on incompressible data -- that data IS bound by the scalar port), hsb (scalar bounds for the head decisions: -0.6 %)
# lwait: the next literal's block reads stay in flight across the isMatch decision (+0.3 .. +0.6 %, profiles/r03/ab_lwait.txt)
ROUND3_VARIANT = {"wsb", "lwait"}
# round 4 (profiles/r04/ab_round4_variants.txt): instruction-count changes, each checked on the emulator first (tools/fuzz_emulated.py)
# and then measured (tools/ab_bench.py; T = 4096 x 1 MiB text, R = incompressible, S = the cfg3 shape, M = mixed):
#   db6    direct bits in six instructions per bit instead of eight (the subtract's borrow is the complemented bit)   T +1.6 %  S +0.7 %
#   tu8    tree_update in eight instructions instead of eleven, no base register to prepare; the 8-level trees' update
#          with one multiply-add                                                                                      T +2.4 %  R +0.9 %  S +0.8 %
#   cchk   ONE window test for a new distance (validity and "source inside the window" are the same comparison)        T +1.3 %
#   lctx   the literal state as ONE bit field of (window.pos << 8 | prevByte); the literal table addressed relative to
#          its start (LDS offsets): three vector instructions less per literal
#   hiss   the head gather's address in three instructions instead of four                                  lctx + hiss: T +1.0 %  R +2.8 %  S +2.1 %
# all five: T +6.5 %  R +3.3 %  S +5.5 %  M +4.2 %
#   g8     the 8-level trees' update addresses slot 0 in the lanes that are no level by a shift count of 31: no select
#   hd2    head update as ONE multiply-add and a DPP shift: (31 p + 2048) >> 5 = p - ((p - 2017) >> 5)
#   rmov2  isRep = 0 and the length choice = 0 leave their bound where it was read; no s_mov into the range per decision
#          g8 + hd2 + rmov2: T +0.7 %  R +0.9 %  M +0.7 %  (S within the noise of +-0.4 %)
# measured and NOT adopted (profiles/r04/ab_round4_variants.txt): rot (the literal loop rotated so that the isMatch branch is the
# back edge: S +0.9 %, R -0.3 %), vcur (the input word on the VALU: one scalar instruction less per input byte, +-0.2 % --
# incompressible data is NOT simply bound by the scalar port), hoist0 (the packet head's hazard s_nop replaced by useful work: -0.4 % T)
#   ml4    the matched literal's gather addresses from the loop's per-lane constants: four instructions instead of nine
#   pkm    a pending copy means the packet before was a match: the literal behind it IS a matched literal, no test of the state
#          ml4 + pkm: +0.3 % on every family; with the fast loop running to 128 bytes in front of a unit's end (kFastOutput): S +0.5 .. 0.8 %
ROUND4_VARIANT = {"db6", "tu8", "cchk", "lctx", "hiss", "g8", "hd2", "rmov2", "ml4", "pkm"}
DEFAULT_VARIANT = {"lgather", "hdpp", "flim", "cflag", "tuc", "vperm", "rlhoist", "bralign", "stub32", "head32", "pktl64"}
VARIANT = set(DEFAULT_VARIANT) | NEXT_VARIANT | ROUND3_VARIANT | ROUND4_VARIANT   # (xlz_kernel.hip passes what lctx / hiss expect: XLZ_NO_LCTX / XLZ_NO_HISS for A/B builds without them)


def ml4():
    return "ml4" in VARIANT and g8()


def mlv():
    return "mlv" in VARIANT and ml4() and "lctx" in VARIANT


def rot():
    """rot (round 4): the literal loop is rotated -- the literal's body lies in FRONT of the packet head that follows it, and
    the isMatch decision's branch is the loop's back edge (taken for a literal); a match falls through into the match path.
    One instruction less per literal (the unconditional branch back to the head) and no taken branch into the match path."""
    return "rot" in VARIANT and "rmov" in VARIANT and "litrun" not in VARIANT


def hdpp_lane(j):
    return 16 * (j // 4) + 4 * (j % 4)


def pend():
    """the SGPR that says a match copy is still in flight.  cflag: its length s95 itself (>= 1)"""
    return "s95" if "cflag" in VARIANT else "s94"


def lgather():
    return "lgather" in VARIANT or "lit8g" in VARIANT

lines = []
stubs = []
finish_sites = []
uid = [0]


def emit(s):
    for l in s.strip("\n").split("\n"):
        l = l.strip()
        if l:
            lines.append(l)


ALIGNED = ("pktl", "match", "mlit", "rep")  # reached by branches only (the code in front ends in s_branch)


def label(name):
    if "bralign" in VARIANT and name == ("litb" if rot() else "pktl"):  # tools/layout.py lays the stream out from here
        lines.append(".p2align %d" % (6 if "pktl64" in VARIANT else 5 if "pktl32" in VARIANT else 4))
    for v in VARIANT:
        if v.startswith("shift") and name == "pktl":  # A/B: everything behind the loop's entry code moves by N dwords
            lines.extend(["s_nop 0"] * int(v[5:]))
    for v in VARIANT:  # A/B: alignN pads the hot loop heads to 2^N bytes, salignN the normalisation stubs
        if v.startswith("align") and name in ALIGNED:
            lines.append(".p2align %s" % v[5:])
            for w in VARIANT:  # offN: the literal loop's head N dwords behind the boundary
                if w.startswith("off") and name == "pktl":
                    lines.extend(["s_nop 0"] * int(w[3:]))
        if v.startswith("salign") and name in stubs:
            lines.append(".p2align %s" % v[6:])
    lines.append(".L%s_%%=:" % name)


def L(name):
    return ".L%s_%%=" % name


def hsb():
    """hsb: the head decisions (isMatch, isRep, ..., the length coders' choice bits) form their bound on the scalar side
    too (the lane read fetches the probability, s_lshr + s_mul in front of the compare): the range's chain no longer
    goes through the vector side to become a bound (cf. wsb)"""
    return "hsb" in VARIANT and "litrun" not in VARIANT


def head_src():
    return "v40" if hsb() else "v55"


def bounds(src, dst="v55", rin="%[range]"):
    """dst = (range >> 11) * p for all 64 probabilities of VGPR `src` -- entirely on the VALU (the
    scalar port is the bottleneck).  The lane select of a later v_readlane then picks the BOUND.
    gfx940-family hazard: that v_readlane must not be the very next instruction (one wait state
    between a VALU write of a VGPR and a v_readlane of it; the assembler does not insert it
    inside inline asm) -- callers put independent work in between."""
    if src == "v40" and hsb():
        return  # (the head probabilities are read as they are; hbit multiplies on the scalar side)
    emit("v_lshrrev_b32 %s, 11, %s\nv_mul_u32_u24 %s, %s, %s" % (dst, rin, dst, dst, src))


def decide(scalar_bound=False, rin="%[range]"):
    """One decision against bound s80.  The CODE lives in v29 (wave-uniform) for the whole loop:
    one subtract with borrow-out gives code - bound and VCC = (code < bound), an unsigned min
    selects the new code; the scalar side only keeps what steers control: the range and
    SCC = (code < bound) = !bit, which the caller uses next (s_addc that advances a tree slot)."""
    if "order3" in VARIANT and scalar_bound:
        # the bound came from s_mul (no lane read two slots ahead that the VALU compare would have to keep its
        # distance from): the scalar subtract moves behind the compare, the VCC reader one slot further away
        emit("""
        v_subrev_co_u32 v28, vcc, s80, v29
        s_sub_u32 s81, RIN, s80
        v_min_u32 v29, v29, v28
        s_cmp_lg_u32 vcc_lo, 0
        s_cselect_b32 %[range], s80, s81
        """.replace("RIN", rin))
        return
    if "order1" in VARIANT:  # round 1's order: the VCC reader right behind its writer (measured 1.7-2.9 % slower)
        emit("""
        s_sub_u32 s81, RIN, s80
        v_subrev_co_u32 v28, vcc, s80, v29
        s_cmp_lg_u32 vcc_lo, 0
        v_min_u32 v29, v29, v28
        s_cselect_b32 %[range], s80, s81
        """.replace("RIN", rin))
        return
    emit("""
    s_sub_u32 s81, RIN, s80
    v_subrev_co_u32 v28, vcc, s80, v29
    v_min_u32 v29, v29, v28
    s_cmp_lg_u32 vcc_lo, 0
    s_cselect_b32 %[range], s80, s81
    """.replace("RIN", rin))


def decide_branchy(k, rin="%[range]"):
    """dbr: one decision against bound s80 that BRANCHES on the outcome (VCC is wave-uniform: s_cbranch_vccz) instead of
    selecting, tree slot included: each outcome writes the range and the slot with two scalar instructions of its own --
    no s_cmp_lg vcc, no s_cselect, no s_sub in front of the outcome: two scalar-port instructions less per decision; the
    bit-1 outcome lies out of line.  Branches do not issue on the scalar port, which is what binds literal-heavy
    data.  VCC = (code < bound) stays valid behind it (l7blk)."""
    uid[0] += 1
    one, join = "d%d" % uid[0], "d%dj" % uid[0]
    first = "slot0" in VARIANT and k == 0
    emit("""
    v_cmp_gt_u32 vcc, s80, v29
    s_cbranch_vccz ONE
    s_mov_b32 %[range], s80
    """.replace("ONE", L(one)))
    emit("s_mov_b32 s88, 3" if first else "s_lshl1_add_u32 s88, s88, 1")
    if "dbrs" in VARIANT:
        # dbrs: the bit-1 outcome lies right behind the bit-0 one: every outcome takes ONE short forward branch (12
        # bytes) instead of none / two far ones
        emit("s_branch %s" % L(join))
        label(one)
        emit("v_subrev_u32 v29, s80, v29\ns_sub_u32 %%[range], %s, s80" % rin)
        emit("s_mov_b32 s88, 2" if first else "s_lshl_b32 s88, s88, 1")
        label(join)
        return
    label(join)

    def out_of_line(one=one, join=join, first=first, rin=rin):
        label(one)
        emit("v_subrev_u32 v29, s80, v29\ns_sub_u32 %%[range], %s, s80" % rin)
        emit("s_mov_b32 s88, 2" if first else "s_lshl_b32 s88, s88, 1")
        emit("s_branch %s" % L(join))
    deferred.append(out_of_line)


def nchk(prefix=None, pick=None, mid=None, late_test=False, rreg=None):
    """normalisation test; the stub is emitted out of line at the end of the block.
    prefix / pick: functions that emit the first instructions of the NEXT decision -- the
    range-only VALU product (bounds) and the lane read of its bound.  They are hoisted in front of
    the branch: the product, the test's s_lshr, the lane read, the branch -- which gives every
    gfx950 wait state (tools/hazards.py) for free.  After a normalisation the stub returns in
    front of them."""
    uid[0] += 1
    k = "n%d" % uid[0]
    if "vnorm" in VARIANT and not mid:
        # the test on the VALU (v17 = 1 << 24) and the branch on VCC: one scalar-port instruction less per decision
        if prefix:
            label(k + "b")
            prefix()
            emit("v_cmp_lt_u32 vcc, %[range], v17")
            if pick:
                pick()
            emit("s_cbranch_vccnz %s" % L(k))
        else:
            emit("v_cmp_lt_u32 vcc, %%[range], v17\ns_cbranch_vccnz %s" % L(k))
            label(k + "b")
        stubs.append(k)
        return
    if prefix:
        label(k + "b")
        prefix()
        if late_test:  # the pick itself uses SCC (level 7 of an 8-level tree: block 2 or 3): test after it
            pick()
            emit("s_lshr_b32 s81, %[range], 24")
        else:
            emit("s_lshr_b32 s81, %[range], 24")
            if pick:
                pick()
        emit("s_cbranch_scc0 %s" % L(k))
    else:
        emit("s_lshr_b32 s80, %s, 24" % (rreg or "%[range]"))
        if mid:  # VALU instructions of the caller between test and branch (SCC is kept)
            emit(mid)
        emit("s_cbranch_scc0 %s" % L(k))
        label(k + "b")
        if rreg:
            stub_reg[k] = rreg
    stubs.append(k)
    if SCODE[0]:
        stub_scode.add(k)


stub_reg = {}  # stub -> the SGPR that holds the range at its site (rmov), default %[range]
stub_scode = set()   # stubs of a walk whose CODE lives in s87 (scode)
SCODE = [False]      # a scode walk is being generated


def vcur():
    return "vcur" in VARIANT and "vperm" in VARIANT and "warel" in VARIANT and "tu8" in VARIANT and "scode" not in VARIANT


def emit_stubs():
    for k in stubs:
        label(k)
        rr = stub_reg.get(k, "%[range]")
        if k in stub_scode:  # the code is in s87 during this walk (s81: the normalisation test's dead result)
            emit("s_lshl_b32 %s, %s, 8\ns_lshl_b32 s87, s87, 8\ns_and_b32 s81, %%[cur], 0xff\ns_or_b32 s87, s87, s81" % (rr, rr))
        elif vcur():
            # vcur (round 4): the current input word lives in v30 (wave-uniform; free since tu8): its shift is a vector
            # instruction -- one scalar-port instruction less per input byte (the port is what binds literal-heavy data)
            emit("""
            s_lshl_b32 %s, %s, 8
            v_perm_b32 v29, v29, v30, v13
            v_lshrrev_b32 v30, 8, v30
            s_sub_u32 s91, s91, 1
            s_cbranch_scc0 %s
            s_add_u32 s90, s90, 1
            s_mov_b32 s91, 3
            v_readlane_b32 %%[cur], %%[vin], s90
            v_mov_b32 v30, %%[cur]
            s_branch %s
            """ % (rr, rr, L(k + "b"), L(k + "b")))
            continue
        elif "vperm" in VARIANT:  # code = code << 8 | next byte in ONE byte permute (v13 = the selector), no scalar mask
            emit("s_lshl_b32 %s, %s, 8\nv_perm_b32 v29, v29, %%[cur], v13" % (rr, rr))
        else:
            emit("s_lshl_b32 %s, %s, 8\ns_and_b32 s80, %%[cur], 0xff\nv_lshl_or_b32 v29, v29, 8, s80" % (rr, rr))
        if "warel" in VARIANT:
            # the input position as (word s90, bytes left in it after this one s91): the countdown's borrow IS the
            # "word used up" test -- one scalar instruction less per input byte; %[arel] is rebuilt where the loop is left
            emit("""
            s_lshr_b32 %%[cur], %%[cur], 8
            s_sub_u32 s91, s91, 1
            s_cbranch_scc0 %s
            s_add_u32 s90, s90, 1
            s_mov_b32 s91, 3
            v_readlane_b32 %%[cur], %%[vin], s90
            s_branch %s
            """ % (L(k + "b"), L(k + "b")))
            continue
        emit("""
        s_lshr_b32 %%[cur], %%[cur], 8
        s_add_u32 %%[arel], %%[arel], 1
        s_and_b32 s80, %%[arel], 3
        s_cbranch_scc1 %s
        s_lshr_b32 s80, %%[arel], 2
        v_readlane_b32 %%[cur], %%[vin], s80
        s_branch %s
        """ % (L(k + "b"), L(k + "b")))


def need_copy_done(inline=False):
    """The previous match copy's load is left in flight while the next packet decodes; whoever
    needs its bytes (prevByte / matchByte for a literal, the next copy, any exit) comes here
    first.  s94 = copy pending, v48 = its destination offsets, v49 = loaded bytes, s95 = len."""
    uid[0] += 1
    k = "f%d" % uid[0]
    if inline:  # a copy is nearly always pending here: no branch out and back
        emit("s_cmp_eq_u32 %s, 0\ns_cbranch_scc1 %s" % (pend(), L(k + "b")))
        finish_body()
        label(k + "b")
        return
    emit("s_cmp_eq_u32 %s, 0\ns_cbranch_scc0 %s" % (pend(), L(k)))
    label(k + "b")
    finish_sites.append(k)


def finish_body():
    if "smask" in VARIANT:  # experiment: store only the len bytes of the copy (not the scratch lanes behind them)
        emit("""
        s_sub_u32 s80, 64, s95
        s_lshr_b64 exec, -1, s80
        s_waitcnt vmcnt(0)
        global_store_byte v48, v49, %[outp]
        s_mov_b64 exec, -1
        s_sub_u32 s80, s95, 1
        v_readlane_b32 %[prev], v49, s80
        v_readlane_b32 %[mb], v49, s95
        s_mov_b32 PEND, 0
        """.replace("PEND", pend()))
        return
    emit("""
    s_waitcnt vmcnt(0)
    global_store_byte v48, v49, %[outp]
    s_sub_u32 s80, s95, 1
    v_readlane_b32 %[prev], v49, s80
    v_readlane_b32 %[mb], v49, s95
    s_mov_b32 PEND, 0
    """.replace("PEND", pend()))
    if "vprev" in VARIANT:  # prevByte lives in v32 (literal_tail): keep it current behind a copy too
        emit("v_mov_b32 v32, %[prev]")


def emit_finish_blocks():
    for k in finish_sites:
        label(k)
        finish_body()
        emit("s_branch %s" % L(k + "b"))


def hoist0():
    """hoist0 (round 4): the first instruction of the isMatch = 0 update (v63 = p - 2017) does not depend on the decision: it
    sits in the wait state between the lane read of the bound and the compare that uses it (gfx950: two wait states
    between a VALU write of an SGPR and its VALU read), where the hazard pass had to put an s_nop -- one instruction less
    per packet head.  (A match recomputes v63 in its own update.)"""
    return "hoist0" in VARIANT and "hdpp" in VARIANT and "flim" in VARIANT and not hsb()


def hd2():
    """hd2 (round 4): p - ((p - 2017) >> 5) = (31 p + 2048) >> 5 and p - (p >> 5) = (31 p + 31) >> 5: the head update is one
    multiply-add (s87 = 2048) and one DPP shift (v30 = 5 in every lane; free since tu8) -- two vector instructions instead of
    three for a decision that comes out 0 (every literal's isMatch, a simple match's isRep and length choice)"""
    return "hd2" in VARIANT and "hdpp" in VARIANT and "tu8" in VARIANT and "scode" not in VARIANT and not vcur() and not hoist0()


def head_update(lane, bit, hoisted=False):
    """new value of head probability `lane` (decompress.go:30 / :177), VALU only: the lanes of v40
    all compute it from their own value, lane `lane` keeps it.  v40 goes back to LDS in one
    store when the packet is over (head_issue / exit)."""
    if "hdpp" in VARIANT:
        # head probability j sits alone in DPP cell (row j / 4, bank j % 4), lane 16 (j / 4) + 4 (j % 4): the
        # last subtract writes only that cell (row_mask / bank_mask) -- no lane compare, no select
        j = (lane // 16) * 4 + (lane % 16) // 4
        assert lane == hdpp_lane(j)
        if hd2():
            emit("v_mad_u32_u24 v63, v40, 31, %s" % ("s87" if bit == 0 else "31"))
            emit("v_lshrrev_b32_dpp v40, v30, v63 quad_perm:[0,1,2,3] row_mask:0x%x bank_mask:0x%x" % (1 << (j // 4), 1 << (j % 4)))
            return
        if bit == 0:
            emit(("" if hoisted else "v_add_u32 v63, 0xfffff81f, v40\n") + "v_ashrrev_i32 v63, 5, v63")
        else:
            emit("v_ashrrev_i32 v63, 5, v40")
        emit("v_sub_u32_dpp v40, v40, v63 quad_perm:[0,1,2,3] row_mask:0x%x bank_mask:0x%x" % (1 << (j // 4), 1 << (j % 4)))
        return
    assert not hoisted
    emit("v_cmp_eq_u32 vcc, %d, %%[vlane]" % lane)
    if bit == 0:
        emit("v_add_u32 v63, 0xfffff81f, v40\nv_ashrrev_i32 v63, 5, v63")
    else:
        emit("v_ashrrev_i32 v63, 5, v40")
    emit("v_sub_u32 v63, v40, v63\nv_cndmask_b32 v40, v40, v63, vcc")


def head_pick(lane, src=None, dst="s80"):
    return lambda: emit("v_readlane_b32 %s, %s, %d" % (dst, src or head_src(), lane))


def hbit(lane, one, stage=0, next_head=None, breg="s80", keep=False, hoisted=False):
    """One decision on head probability `lane` (already in v40), both outcomes specialised: VCC of
    the compare is branched on directly.  Bit 0 falls through; bit 1 goes to label `one`, where
    the caller emits hbit_one(lane) first.
    stage: what the code in front already did for this decision -- 0 nothing, 1 bounds("v40"),
    2 also the lane read of the bound (both hoisted in front of the previous normalisation test).
    next_head: lane of the head decision the bit-0 path continues with (hoisted likewise)."""
    if stage < 1:
        bounds("v40")
    if stage < 2:
        emit("v_readlane_b32 %s, %s, %d" % (breg, head_src(), lane))
    if hsb():
        emit("s_lshr_b32 s81, %%[range], 11\ns_mul_i32 %s, s81, %s" % (breg, breg))
    emit("v_cmp_gt_u32 vcc, %s, v29\ns_cbranch_vccz %s" % (breg, one))
    if not keep:  # keep (rmov): the new range stays in `breg`; the literal's first level reads it there
        emit("s_mov_b32 %%[range], %s" % breg)
    head_update(lane, 0, hoisted=hoisted)
    if next_head is not None:
        assert not keep
        nchk(prefix=lambda: bounds("v40"), pick=head_pick(next_head))
    else:
        nchk(rreg=breg if keep else None)


def hbit_one(lane, next_head=None, breg="s80", rin="%[range]", pick_dst="s80"):
    emit("v_subrev_u32 v29, %s, v29\ns_sub_u32 %%[range], %s, %s" % (breg, rin, breg))
    head_update(lane, 1)
    if next_head is not None:
        nchk(prefix=lambda: bounds("v40"), pick=head_pick(next_head, dst=pick_dst))
    else:
        nchk()


def level_prefix(k, blocks):
    """range-only part of level k (0-based) of a tree walk: bounds of all slots of its block(s)"""
    if k <= 5:
        bounds(blocks[0])
    elif k == 6:
        bounds(blocks[1])
    else:  # slot 128..255: block 2 or 3 by bit 6
        bounds(blocks[2])
        bounds(blocks[3], dst="v62")


def level_pick(k):
    """bound of slot s88 -> s80 (one instruction must separate this from level_prefix)"""
    emit("v_readlane_b32 s80, v55, s88")
    if k == 7:
        emit("v_readlane_b32 s84, v62, s88\ns_bitcmp1_b32 s88, 6\ns_cselect_b32 s80, s84, s80")


def walk(nbits, blocks, early_exit=None, filler="s_nop 0", range0=None):
    """Walk nbits <= 6 levels of a bit tree whose 64-prob block is already in blocks[0]
    (bit_tree_decoder.go:18-40).  s88 ends as 1 followed by the COMPLEMENTED decided bits (the tree
    slot order of xlz_kernel.hip: tree_slot).  Nothing is recorded: tree_update finds the
    probabilities again in the block register.  early_exit = (sgpr, label): leave after as many
    levels as the SGPR says (reverse tree over posDecoders, 1..5 levels).
    filler: independent instruction(s) of the caller, placed in the wait state between the first
    VALU product and its lane read."""
    assert nbits <= 6
    if "wsb" in VARIANT and lgather():
        # wsb: scalar bound as in the 8-level walks (the probability comes with the lane read, s_lshr + s_mul form the
        # bound): two scalar instructions more per level than the VALU product, but the range's dependent chain crosses
        # between the scalar and the vector side twice per level instead of four times -- the waves are bound by that
        # chain's latency, not by the issue ports (bench.py roofline.issue: 78 % of a lone wave's speed at 16 per CU)
        emit(filler)
        walk_rec(nbits, blocks, early_exit=early_exit, range0=range0)
        return
    assert range0 is None
    level_prefix(0, blocks)
    emit(filler)
    emit("v_readlane_b32 s80, v55, 1\n" + slot_init())
    for k in range(nbits):
        if "dbrw" in VARIANT:
            decide_branchy(k)
        else:
            decide()
            slot_step(k)  # J = 2J + SCC = 2J + !bit
        if k + 1 < nbits:
            nchk(prefix=lambda: level_prefix(k + 1, blocks), pick=lambda: level_pick(k + 1))
            if early_exit:
                emit("s_cmp_eq_u32 %s, %d\ns_cbranch_scc1 %s" % (early_exit[0], k + 1, early_exit[1]))
        else:
            nchk()


def tree_update(nb, blocks, base="v58", addr=None, off=0, dump="v38"):
    """Apply the model updates of a finished walk.  s88 = final slot (1, then the complemented
    bits), `blocks` = the tree's block registers (lane j of block b = slot 64b + j), `base` = VGPR
    with the byte address of the tree.  nb: int, or the name of an SGPR (single-block trees).
    Every lane looks at its OWN slot: slot j at level l (v31 = l + 1) was visited iff
    s88 >> (nb - l) == j, the decision taken there was !((s88 >> (nb - l - 1)) & 1), and the new
    value is p - ((p - (bit ? 0 : 2017)) >>a 5) (decompress.go:30 / :177).  Pure VALU work on
    registers that are already there -- nothing is recorded during the walk.  Only visited
    slots are stored (a block register may overlap other tables and be stale there); the other
    lanes store to the unused slot whose address is in v38.  Single-block trees."""
    if "tu8" in VARIANT:
        # tu8 (round 4): eight instructions instead of eleven, and no base register to prepare.  v61 = 2 * slot + !bit of
        # the lane's level; XOR with 2 * lane (v56) is 0 / 1 exactly in the visited lanes and then IS !bit; one multiply-add
        # forms p - 2017 * !bit (s85 = -2017); the store goes to `addr` (+ offset: the VGPR the block was requested with), the
        # other lanes to `dump` (a VGPR or an inline constant: an unused slot, relative to the same offset)
        assert addr
        if "tuc" in VARIANT and nb in (3, 4, 6):
            emit("v_lshrrev_b32 v61, %s, s88" % {3: "v14", 4: "v15", 6: "v16"}[nb])
        else:
            emit("v_sub_u32 v55, %s, v31\nv_lshrrev_b32 v61, v55, s88" % nb)
        emit("""
        v_xor_b32 v60, v61, v56
        v_cmp_gt_u32 vcc, 2, v60
        v_mad_i32_i24 v61, v60, s85, %s
        v_ashrrev_i32 v61, 5, v61
        v_sub_u32 v61, %s, v61
        v_cndmask_b32 v60, %s, %s, vcc
        ds_write_b16 v60, v61%s
        """ % (blocks[0], blocks[0], dump, addr, (" offset:%d" % off) if off else ""))
        return
    own = "v30"
    if "tuc" in VARIANT and nb in (3, 4, 6):  # nb - level(lane) is a loop constant: v14 / v15 / v16
        emit("v_lshrrev_b32 v61, %s, s88" % {3: "v14", 4: "v15", 6: "v16"}[nb])
    else:
        emit("v_sub_u32 v55, %s, v31\nv_lshrrev_b32 v61, v55, s88" % nb)
    emit("""
    v_lshrrev_b32 v60, 1, v61
    v_cmp_eq_u32 vcc, v60, %s
    v_and_b32 v61, 1, v61
    v_mul_u32_u24 v61, 0x7e1, v61
    v_sub_u32 v61, %s, v61
    v_ashrrev_i32 v61, 5, v61
    v_sub_u32 v61, %s, v61
    v_add_u32 v60, %s, v56
    v_cndmask_b32 v60, v38, v60, vcc
    ds_write_b16 v60, v61
    """ % (own, blocks[0], blocks[0], base))
    assert len(blocks) == 1


# ---- 8-level trees (literal, high length): four blocks.  Here the per-slot update above would
# be four vector passes and the VALU pipe becomes the bottleneck (measured: -15 % on literal-heavy
# data), so these trees keep a record: the probability of level k is parked in lane k of v54
# (v_writelane) and ONE vector operation, lanes = levels, updates them.

def fetch_level(k, blocks, merged=False):
    """probability of tree slot s88 at level k (0-based) of an 8-level tree -> s86.  merged (l7blk): v62 already
    holds the one of the two last-level blocks this walk can reach (its first decision chose it)"""
    if k == 7 and merged:
        emit("v_readlane_b32 s86, v62, s88")
    elif k <= 5:
        emit("v_readlane_b32 s86, %s, s88" % blocks[0])
    elif k == 6:
        emit("v_readlane_b32 s86, %s, s88" % blocks[1])
    else:  # slot 128..255: block 2 or 3 by bit 6
        emit("v_readlane_b32 s86, %s, s88\nv_readlane_b32 s84, %s, s88\ns_bitcmp1_b32 s88, 6\n"
             "s_cselect_b32 s86, s84, s86" % (blocks[2], blocks[3]))


def slot_step(k):
    """tree slot J = 2J + SCC (SCC = !bit).  slot0: the first level of a walk knows J = 1, so it writes 3 or 2
    directly and nobody has to set s88 = 1 first (one scalar instruction less per tree walk)"""
    if "slot0" in VARIANT and k == 0:
        emit("s_cselect_b32 s88, 3, 2")
    else:
        emit("s_addc_u32 s88, s88, s88")


def ismatch_reg():
    """the SGPR the isMatch bound is read into (rmov keeps it there as the literal's range)"""
    return "s82" if "rmov" in VARIANT else "s80"


def slot_init():
    return "" if "slot0" in VARIANT else "s_mov_b32 s88, 1"


def level_rec(k=None, rin="%[range]"):
    """decision of a recorded level on the probability in s86 (parked in lane k of v54)"""
    if "dbr" in VARIANT and lgather():
        emit("s_lshr_b32 s80, %s, 11\ns_mul_i32 s80, s80, s86" % rin)
        decide_branchy(k, rin)
        return
    if SCODE[0]:
        # scode: the decision entirely on the scalar side -- code - bound with its borrow in SCC, two selects -- one
        # scalar instruction more and two vector ones less than the VALU compare, and the range's dependent chain no
        # longer crosses to the vector side and back through VCC
        emit("""
        s_lshr_b32 s80, RIN, 11
        s_mul_i32 s80, s80, s86
        s_sub_u32 s81, RIN, s80
        s_sub_u32 s83, s87, s80
        s_cselect_b32 %[range], s80, s81
        s_cselect_b32 s87, s87, s83
        """.replace("RIN", rin))
        slot_step(k)
        return
    emit("s_lshr_b32 s80, %s, 11\ns_mul_i32 s80, s80, s86" % rin)
    if k is not None and not lgather():
        emit("v_writelane_b32 v54, s86, %d" % k)
    decide(scalar_bound=True, rin=rin)
    slot_step(k)


def walk_rec(nbits, blocks, entries=None, range0=None, early_exit=None):
    """walk() for the 8-level trees (scalar bound: the probability is read with v_readlane, the bound formed
    with s_lshr / s_mul).  entries = label prefix: entered at level k >= 1 through <prefix>k with s88 set."""
    hoist = "rlhoist" in VARIANT and (not entries or "pwhoist" in VARIANT)
    scode = "scode" in VARIANT and lgather() and not entries and nbits == 8 and blocks is LIT_BLOCKS and not early_exit
    if scode:
        SCODE[0] = True
        emit("v_readfirstlane_b32 s87, v29")
    if not entries:
        if "flim" in VARIANT and blocks is LIT_BLOCKS:  # (the packet head set s88 = 1 in a wait state)
            emit("v_readlane_b32 s86, %s, 1" % blocks[0])
        else:
            emit(slot_init() + "\nv_readlane_b32 s86, %s, 1" % blocks[0])
    for k in range(nbits):
        if entries:
            if k == 0:
                continue
            if hoist:
                label("%sd%d" % (entries, k))  # level k with its probability already in s86
            else:
                label("%s%d" % (entries, k))
                fetch_level(k, blocks)
        level_rec(k, rin=range0) if (k == 0 and range0) else level_rec(k)
        merged = "l7blk" in VARIANT and not entries and nbits == 8
        if merged and k == 0:
            # slots 128..255 of the last level lie in blocks[2] (first decided bit 1) or blocks[3] (bit 0: the slot's
            # complemented bit 6 is set), and VCC still holds level 0's borrow = that complemented bit in every lane:
            # one select now replaces two lane reads + a bit test + a scalar select at level 7
            if scode:  # (the decision left its outcome in SCC only; slot_step(0) keeps SCC)
                emit("s_cselect_b64 vcc, -1, 0")
            emit("v_cndmask_b32 v62, %s, %s, vcc" % (blocks[2], blocks[3]))
        if hoist and k + 1 < nbits:
            # the next level's probability is read in front of the normalisation branch: the lane read's result
            # has the branch between it and its first use (s_mul)
            uid[0] += 1
            kk = "n%d" % uid[0]
            if k + 1 == 7 and not merged:  # (its block select uses SCC)
                fetch_level(k + 1, blocks)
                emit("s_lshr_b32 s81, %[range], 24")
            else:
                emit("s_lshr_b32 s81, %[range], 24")
                fetch_level(k + 1, blocks, merged)
            emit("s_cbranch_scc0 %s" % L(kk))
            label(kk + "b")  # (the stub changes neither the slot nor the probability read for it)
            stubs.append(kk)
            if scode:
                stub_scode.add(kk)
            if early_exit:
                emit("s_cmp_eq_u32 %s, %d\ns_cbranch_scc1 %s" % (early_exit[0], k + 1, early_exit[1]))
            continue
        nchk()
        if k + 1 < nbits and not entries:
            if early_exit:
                emit("s_cmp_eq_u32 %s, %d\ns_cbranch_scc1 %s" % (early_exit[0], k + 1, early_exit[1]))
            fetch_level(k + 1, blocks, merged)
    if scode:
        SCODE[0] = False
        emit("v_mov_b32 v29, s87")


def walk8(blocks, entries=None):
    """8-level walk with the bounds from a VALU product (as walk()): nothing is recorded, a level is
    6 scalar instructions + one lane read instead of 8 + 2.  gather8() finds the eight probabilities
    again for the update.  entries = label prefix: <prefix>k enters at level k >= 1 with s88 set."""
    if not entries:
        level_prefix(0, blocks)
        emit("s_mov_b32 s88, 1")
        emit("v_readlane_b32 s80, v55, 1")
    for k in range(8):
        if entries:
            if k == 0:
                continue
            label("%sd%d" % (entries, k))
        decide()
        emit("s_addc_u32 s88, s88, s88")
        if k + 1 < 8:
            nchk(prefix=lambda: level_prefix(k + 1, blocks), pick=lambda: level_pick(k + 1), late_test=(k + 1 == 7))
        else:
            nchk()
    if entries:  # out of line: the bound of the level entered at, then into the chain
        deferred.append(lambda: walk8_entries(blocks, entries))


def walk8_entries(blocks, entries):
    for k in range(1, 8):
        label("%s%d" % (entries, k))
        level_prefix(k, blocks)
        emit("s_nop 0")
        level_pick(k)
        emit("s_branch %s" % L("%sd%d" % (entries, k)))


def gather8(blocks, dst="v54"):
    """the eight probabilities a finished 8-level walk met -> lanes 0..7 of dst (lane k = level k):
    slot_k = s88 >> (8 - k) lives in block slot_k >> 6 at lane slot_k & 63; one ds_bpermute per block
    (its lane select wraps mod 64), then lane 6 takes block 1 and lane 7 block 2 or 3."""
    emit("""
    v_lshrrev_b32 v60, v19, s88
    v_lshlrev_b32 v61, 2, v60
    ds_bpermute_b32 v62, v61, %s
    ds_bpermute_b32 v63, v61, %s
    ds_bpermute_b32 v28, v61, %s
    ds_bpermute_b32 v34, v61, %s
    s_bitcmp1_b32 s88, 7
    s_cselect_b64 vcc, -1, 0
    s_waitcnt lgkmcnt(0)
    v_cndmask_b32 %s, v62, v63, s[78:79]
    v_cndmask_b32 v28, v28, v34, vcc
    v_cmp_eq_u32 vcc, 7, %%[vlane]
    v_cndmask_b32 %s, %s, v28, vcc
    """ % (blocks[0], blocks[1], blocks[2], blocks[3], dst, dst, dst))


def g8():
    """g8 (round 4): the per-lane shift count v19 (8 - level) is 31 in the lanes that are no level (>= 8): their slot is 0, the
    tree's unused entry, without a select -- one vector instruction less per 8-level tree update"""
    return "g8" in VARIANT


def lit_off(base):
    """lctx: the literal table is addressed relative to its start (v39 = litState << 9); the LDS instructions carry P_LIT"""
    return P_LIT * 2 if ("lctx" in VARIANT and base == "v39") else 0


def rec_gather_issue(base, masked):
    """lgather: nothing was recorded during the walk: the eight probabilities it met are read back from LDS
    with ONE 8-lane gather at the addresses the update stores to anyway (instead of eight v_writelane),
    issued as soon as the walk is over so that the round trip hides behind the literal's tail.
    masked: VCC = lanes whose level was decided in the matched table (their v54 is the HBM gather).
    v60 = the addresses (kept for the store), data -> v54 (masked: v33)."""
    emit("""
    v_lshrrev_b32 v60, v19, s88
    v_lshl_add_u32 v60, v60, 1, %s
    """ % base)
    off = lit_off(base)
    dump = "0" if off else "v38"   # (relative addressing: slot 0 of the first literal state's tree, which no walk visits)
    if masked:
        emit("v_cndmask_b32 v60, v60, %s, vcc" % dump)
    if not g8():
        emit("v_cndmask_b32 v60, %s, v60, s[76:77]" % dump)
    emit("ds_read_u16 %s, v60%s" % ("v33" if masked else "v54", (" offset:%d" % off) if off else ""))


def tree_update_rec(nb, base, store=True, issued=False, pending=0, filler=None):
    """model updates of a walk_rec in ONE vector operation, lane k = level k: slot = s88 >> (8-k),
    !bit = (s88 >> (7-k)) & 1, p = v54; lanes >= 8 store to the unused slot (v38).  The per-lane
    shift counts 8-k and 7-k are v19 / v18, the lanes-below-8 mask is s[76:77] (loop constants)."""
    assert nb == 8
    if lgather():
        if not issued:
            rec_gather_issue(base, masked=not store)
            pending = 0
        tu8 = "tu8" in VARIANT
        emit("v_bfe_u32 v61, s88, v18, 1" + ("" if tu8 else "\nv_mul_u32_u24 v61, 0x7e1, v61"))
        if filler:
            filler()  # independent work of the caller in front of the wait
        emit("s_waitcnt lgkmcnt(%d)" % pending)
        if not store:
            emit("v_cndmask_b32 v54, v33, v54, vcc")
        emit("v_mad_i32_i24 v61, v61, s85, v54" if tu8 else "v_sub_u32 v61, v54, v61")   # p - 2017 * !bit (s85 = -2017)
        emit("""
        v_ashrrev_i32 v61, 5, v61
        v_sub_u32 v61, v54, v61
        """)
        if store:
            emit("ds_write_b16 v60, v61" + ((" offset:%d" % lit_off(base)) if lit_off(base) else ""))
        return
    emit("""
    v_lshrrev_b32 v60, v19, s88
    v_bfe_u32 v61, s88, v18, 1
    v_lshl_add_u32 v60, v60, 1, %s
    v_mul_u32_u24 v61, 0x7e1, v61
    v_sub_u32 v61, v54, v61
    v_ashrrev_i32 v61, 5, v61
    v_sub_u32 v61, v54, v61
    """ % base)
    if store:
        off = lit_off(base)
        if not g8():
            emit("v_cndmask_b32 v60, %s, v60, s[76:77]" % ("0" if off else "v38"))
        emit("ds_write_b16 v60, v61%s" % ((" offset:%d" % off) if off else ""))


def len_request(base):
    """request the low and mid length trees of this posState (blocks v41, v42; v22 = posState, on
    the VALU).  The high tree (four blocks) is rare and is requested where it is needed."""
    av = "v58" if "tu8" in VARIANT else "v59"   # tu8: the update stores through the request's own address vector
    emit("""
    v_and_b32 v22, %%[wpos], %%[vpm]
    v_lshl_add_u32 AV, v22, 4, v56
    ds_read_u16 v41, AV offset:%d
    ds_read_u16 v42, AV offset:%d
    """.replace("AV", av) % ((base + LEN_LOW) * 2, (base + LEN_MID) * 2))


def len_pick(lane_c):
    """bounds("v40") was emitted at least one instruction ago: read the bound of the length
    coder's first decision (head lane lane_c)"""
    emit("v_readlane_b32 s80, %s, %d" % (head_src(), lane_c))


def posslot_request(static):
    """posSlot block of lenState = min(len, 3) (decompress.go:433-441) requested into v36; s92 = its
    tree base.  static: the length is >= 8, lenState is 3."""
    if static:
        emit("s_movk_i32 s92, %d" % ((P_POS_SLOT + 3 * 64) * 2))
    else:
        emit("v_readlane_b32 s92, v21, s89")  # table: lane = length 0..7 -> base of posSlot[min(len, 3)]
    emit("v_add_u32 v57, s92, v56\nds_read_u16 v36, v57" if "tu8" in VARIANT else "v_add_u32 v59, s92, v56\nds_read_u16 v36, v59")


def rmov2():
    """rmov2 (round 4): what rmov does for isMatch -> literal, for isRep -> length choice -> length tree: a head decision
    that comes out 0 leaves its bound, which is the new range, where it was read (s83, then s84) and the next decision takes
    it from there -- no s_mov into %[range] per decision (two scalar instructions less per simple match)"""
    return "rmov2" in VARIANT and "wsb" in VARIANT and lgather() and not hsb() and "litrun" not in VARIANT


def len_decode(tag, base, lane_c, lane_c2, posslot, chain=None):
    """lenDecoder.Decode (len_decoder.go:34-60): raw length -> s89, the walked tree updated.
    posslot: request the distance-slot block as soon as the length is known (simple match).
    chain = (rin, breg): rmov2 -- the range comes in `rin`, the choice's bound was read into `breg`"""
    h0 = hoist0()
    if h0:  # (the wait state between the lane read of the choice's bound and the compare; see hoist0)
        emit("v_add_u32 v63, 0xfffff81f, v40")
    if chain:
        hbit(lane_c, L(tag + "c2"), stage=2, breg=chain[1], keep=True, hoisted=h0)
    else:
        hbit(lane_c, L(tag + "c2"), stage=2, hoisted=h0)  # bounds("v40") and the lane read: len_prefetch
    emit("s_waitcnt lgkmcnt(0)")
    tu8 = "tu8" in VARIANT
    walk(3, ["v41"], filler="" if tu8 else "v_lshlrev_b32 v58, 4, v22\nv_add_u32 v58, %d, v58" % ((base + LEN_LOW) * 2),
         range0=chain[1] if chain else None)
    emit("s_andn2_b32 s89, 7, s88")
    if posslot:
        posslot_request(False)
    tree_update(3, ["v41"], addr="v58", off=(base + LEN_LOW) * 2, dump="0")
    label(tag + "end")  # the common (low) path runs straight on; mid and high trees are out of line
    deferred.append(lambda: len_decode_rest(tag, base, lane_c, lane_c2, posslot, chain))


def len_decode_rest(tag, base, lane_c, lane_c2, posslot, chain=None):
    label(tag + "c2")
    if chain:
        hbit_one(lane_c, next_head=lane_c2, breg=chain[1], rin=chain[0])
    else:
        hbit_one(lane_c, next_head=lane_c2)
    hbit(lane_c2, L(tag + "hi"), stage=2)
    emit("s_waitcnt lgkmcnt(0)")
    walk(3, ["v42"], filler="" if "tu8" in VARIANT else "v_lshlrev_b32 v58, 4, v22\nv_add_u32 v58, %d, v58" % ((base + LEN_MID) * 2))
    emit("s_xor_b32 s89, s88, 7")
    if posslot:
        posslot_request(True)
    tree_update(3, ["v42"], addr="v58", off=(base + LEN_MID) * 2, dump="0")
    emit("s_branch %s" % L(tag + "end"))
    label(tag + "hi")
    hbit_one(lane_c2)
    hbm = base == P_REP_LEN  # the rep-length coder's high tree lives in the model's HBM part (xlz_format.h: kRepHigh)
    if hbm:
        # (vmcnt(0) also waits for a pending match copy's load: its data is then simply there earlier)
        emit("""
        global_load_ushort v43, v56, %[mptr]
        global_load_ushort v44, v56, %[mptr] offset:128
        global_load_ushort v45, v56, %[mptr] offset:256
        global_load_ushort v46, v56, %[mptr] offset:384
        v_mov_b32 v58, 0
        s_waitcnt vmcnt(0)
        """)
    else:
        emit("""
        ds_read_u16 v43, v56 offset:%d
        ds_read_u16 v44, v56 offset:%d
        ds_read_u16 v45, v56 offset:%d
        ds_read_u16 v46, v56 offset:%d
        v_mov_b32 v58, %d
        s_waitcnt lgkmcnt(0)
        """ % ((base + LEN_HIGH) * 2, (base + LEN_HIGH) * 2 + 128, (base + LEN_HIGH) * 2 + 256, (base + LEN_HIGH) * 2 + 384,
               (base + LEN_HIGH) * 2))
    walk_rec(8, ["v43", "v44", "v45", "v46"])
    emit("s_andn2_b32 s89, 0xff, s88\ns_add_u32 s89, s89, 16")
    if posslot:
        posslot_request(True)
    if hbm:
        tree_update_rec_hbm("v58")
    else:
        tree_update_rec(8, "v58")
    emit("s_branch %s" % L(tag + "end"))


def tree_update_rec_hbm(base):
    """tree_update_rec for an 8-level tree in the model's HBM part (%[mptr] + base): the eight probabilities the walk met
    are gathered from there (lane k = level k; the other lanes address entry 0, a tree slot no walk visits), updated
    in one vector operation and stored back"""
    emit("""
    v_lshrrev_b32 v60, v19, s88
    v_lshl_add_u32 v60, v60, 1, %s
    G8MASK
    global_load_ushort v54, v60, %%[mptr]
    v_bfe_u32 v61, s88, v18, 1
    MUL
    s_waitcnt vmcnt(0)
    SUB
    v_ashrrev_i32 v61, 5, v61
    v_sub_u32 v61, v54, v61
    global_store_short v60, v61, %%[mptr]
    """.replace("G8MASK", "" if g8() else "v_cndmask_b32_e64 v60, 0, v60, s[76:77]")
       .replace("MUL", "" if "tu8" in VARIANT else "v_mul_u32_u24 v61, 0x7e1, v61")
       .replace("SUB", "v_mad_i32_i24 v61, v61, s85, v54" if "tu8" in VARIANT else "v_sub_u32 v61, v54, v61") % base)


wstubs = []
deferred = []


def wpos_advance(amount):
    """window.pos += amount with the wrap of window.go:38-41 out of line"""
    uid[0] += 1
    k = "w%d" % uid[0]
    if "flim" in VARIANT:  # s96 = the next EVENT in window.pos terms: the dictionary's end or the output limit
        emit("s_add_u32 %%[wpos], %%[wpos], %s\ns_cmp_ge_u32 %%[wpos], s96\ns_cbranch_scc1 %s" % (amount, L(k)))
    else:
        emit("s_add_u32 %%[wpos], %%[wpos], %s\ns_cmp_ge_u32 %%[wpos], %%[dict]\ns_cbranch_scc1 %s" % (amount, L(k)))
    label(k + "b")
    wstubs.append(k)


def event_limit(pos="%[pos]"):
    """s96 = min(dictSize, window.pos + (pos_lim - pos)), the sum saturated"""
    emit("""
    s_sub_u32 s80, %[pos_lim], POS
    s_add_u32 s80, s80, %[wpos]
    s_cselect_b32 s80, -1, s80
    s_min_u32 s96, s80, %[dict]
    """.replace("POS", pos))


def nopos():
    """nopos: the output position lives in v17 (wave-uniform) inside the loop -- the addresses of stores and copies
    are vector operands anyway -- and %[pos] is written back where the loop is left.  What the scalar side needs is
    how full the window is: window.pos while it has not wrapped, dictSize from then on (s94 = 0 / dictSize, the
    fill is max(window.pos, s94)).  One scalar instruction less per packet."""
    if "nopos" in VARIANT:
        assert "flim" in VARIANT and "cflag" in VARIANT and "vnorm" not in VARIANT and "litrun" not in VARIANT
        return True
    return False


def packet_limits(head_lane, breg="s80"):
    """the limit tests of a packet head + the lane read of the first decision's bound (bounds("v40") was
    just emitted).  flim: ONE test -- the output limit is folded into the window-wrap test of the packet
    before (wpos_advance), which turns s99 into -1."""
    if "flim" in VARIANT:
        emit("""
        s_cmp_gt_i32 AREL, s99
        v_readlane_b32 %s, HSRC, %d
        s_cbranch_scc1 %s
        """.replace("AREL", "s90" if "warel" in VARIANT else "%%[arel]").replace("HSRC", head_src()) % (breg, head_lane, L("x0")))
        emit(slot_init())  # (a wait state between the lane read and the compare that uses its result)
        if hoist0():
            assert head_lane == H_IS_MATCH
            emit("v_add_u32 v63, 0xfffff81f, v40")
        return
    emit("""
    s_cmp_gt_u32 %%[arel], %%[arel_lim]
    s_cbranch_scc1 %s
    v_readlane_b32 %s, HSRC, %d
    s_cmp_ge_u32 %%[pos], %%[pos_lim]
    s_cbranch_scc1 %s
    """.replace("HSRC", head_src()) % (L("x0"), breg, head_lane, L("x0")))


def emit_wstubs():
    for k in wstubs:
        label(k)
        if "flim" in VARIANT:
            # window.pos wrapped (window.go:38-41) and / or the output limit is reached: the latter makes the
            # next packet head leave (its input test cannot pass against s99 = -1)
            emit("""
            s_cmp_ge_u32 %%[wpos], %%[dict]
            s_cbranch_scc0 %s
            s_sub_u32 %%[wpos], %%[wpos], %%[dict]
            """ % L(k + "n"))
            if nopos():
                emit("s_mov_b32 s94, %[dict]")  # the window is full from here on
            label(k + "n")
            if nopos():
                emit("v_readfirstlane_b32 s81, v17\ns_cmp_ge_u32 s81, %[pos_lim]\ns_cselect_b32 s99, -1, s99")
                event_limit(pos="s81")
            else:
                emit("s_cmp_ge_u32 %[pos], %[pos_lim]\ns_cselect_b32 s99, -1, s99")
                event_limit()
            emit("s_branch %s" % L(k + "b"))
            continue
        emit("s_sub_u32 %%[wpos], %%[wpos], %%[dict]\ns_branch %s" % L(k + "b"))


def literal_context(prev_v=None, part=None):
    """literal table base -> v39 (decompress.go:56-57; byte address in LDS) and its four 64-prob
    blocks requested.  All on the VALU: the scalar port is the busy one.  prev_v = VGPR that
    already holds prevByte.  part: "addr" / "reads" emit only the address arithmetic / the requests."""
    if part != "reads":
        if prev_v is None and "vprev" in VARIANT:
            prev_v = "v32"
        if prev_v is None:
            emit("v_mov_b32 v55, %[prev]")
            prev_v = "v55"
        if "lctx" in VARIANT:
            # lctx (round 4): litState = ((window.pos & lpMask) << lc) + (prevByte >> (8 - lc)) (decompress.go:56) is ONE bit
            # field of (window.pos << 8 | prevByte): a byte permute (the normalisation's selector v13 puts the low byte of
            # its second source under the low three bytes of the first) and a bit-field extract of lc + lp bits (%[vlpm] holds
            # lc + lp in this variant).  Only the low byte of prev_v is looked at, so the literal's byte needs no mask.
            emit("""
            v_perm_b32 v57, %%[wpos], %s, v13
            v_bfe_u32 v55, v57, %%[lc8], %%[vlpm]
            v_lshlrev_b32 v39, 9, v55
            v_add_u32 v59, v39, v56
            """ % prev_v)
        else:
          emit("""
        v_and_b32 v57, %%[wpos], %%[vlpm]
        v_lshlrev_b32 v57, %%[lc], v57
        v_lshrrev_b32 v55, %%[lc8], %s
        v_add_lshl_u32 v57, v57, v55, 9
        v_add_u32 v39, %d, v57
        v_add_u32 v59, v39, v56
        """ % (prev_v, P_LIT * 2))
    if part != "addr":
        o = P_LIT * 2 if "lctx" in VARIANT else 0
        emit("""
        ds_read_u16 v50, v59%s
        ds_read_u16 v51, v59 offset:%d
        ds_read_u16 v52, v59 offset:%d
        ds_read_u16 v53, v59 offset:%d
        """ % ((" offset:%d" % o) if o else "", o + 128, o + 256, o + 384))


def head_issue(first=False):
    """the finished packet's head probabilities go back to LDS; the head gather of the packet about
    to start (addresses from its posState / state2, on the VALU)"""
    if not first:
        emit("ds_write_b16 v47, v40")
    if "hiss" in VARIANT:
        # hiss (round 4): address = hc + state * hms + (state << 4 | posState) * hm2 = hc + state * (hms + 16 hm2) + posState * hm2;
        # %[vhms] holds hms + 16 hm2 in this variant: one instruction less per packet
        emit("""
        v_and_b32 v55, %[wpos], %[vpm]
        v_mad_u32_u24 v47, %[state], %[vhms], %[vhc]
        v_mad_u32_u24 v47, v55, %[vhm2], v47
        ds_read_u16 v40, v47
        """)
        return
    emit("""
    v_and_b32 v55, %[wpos], %[vpm]
    v_lshl_add_u32 v55, %[state], 4, v55
    v_mad_u32_u24 v47, %[state], %[vhms], %[vhc]
    v_mad_u32_u24 v47, v55, %[vhm2], v47
    ds_read_u16 v40, v47
    """)


LIT_BLOCKS = ["v50", "v51", "v52", "v53"]


def literal_tail(run_entry=None):
    """window.PutByte (:168), state (:171), then the next packet's head gather; the caller then
    applies the model update and requests the next literal blocks (which may be the very table
    just updated, hence after the update's store).  v32 = the byte.
    run_entry: label to leave to when the new state is 0 (literal run, sec_literal_run)."""
    if "vprev" in VARIANT:
        # the byte (the complement of the slot's low eight bits) is formed on the VALU; the scalar copy of prevByte
        # is only brought up to date where the loop is left (sec_exits): one scalar instruction less per literal
        emit("v_not_b32 v32, s88" + ("" if "lctx" in VARIANT else "\nv_and_b32 v32, 0xff, v32"))
    else:
        emit("s_andn2_b32 %[prev], 0xff, s88\nv_mov_b32 v32, %[prev]")
    if nopos():
        emit("global_store_byte v17, v32, %[outp]\nv_add_u32 v17, 1, v17")
    else:
        emit("""
        v_mov_b32 v61, %[pos]
        global_store_byte v61, v32, %[outp]
        s_add_u32 %[pos], %[pos], 1
        """)
    wpos_advance("1")
    emit("v_readlane_b32 %[state], %[vlitnext], %[state]")  # stateUpdateLiteral as a 12-lane table
    if run_entry:
        emit("s_cmp_eq_u32 %%[state], 0\ns_cbranch_scc1 %s" % run_entry)
    head_issue()


def plain_literal(run_entry=None, range0=None, loop=True):
    """plain literal (:127-175) with its blocks in v50..v53 and base in v39; ends at pktl (loop=False: falls into what follows)"""
    if "lit8" in VARIANT or "lit8g" in VARIANT:
        walk8(LIT_BLOCKS)
        if lgather():
            rec_gather_issue("v39", masked=False)
        literal_tail()
        if "lit8" in VARIANT:
            gather8(LIT_BLOCKS)
    else:
        if "lwait" in VARIANT:
            emit("s_waitcnt lgkmcnt(0)")
        walk_rec(8, LIT_BLOCKS, range0=range0)
        if lgather():
            rec_gather_issue("v39", masked=False)
        literal_tail(run_entry=run_entry)
    if lgather():  # behind the gather: the head's write and its next gather
        tree_update_rec(8, "v39", issued=True, pending=2, filler=lambda: literal_context(prev_v="v32", part="addr"))
        literal_context(part="reads")
    else:
        tree_update_rec(8, "v39")
        literal_context(prev_v="v32")
    if loop:
        emit("s_branch %s" % L("pktl"))


def sec_packet_after_literal():
    """packet head when the previous packet was a literal, and the plain literal"""
    # ------------------------------------------------------------- packet after a literal
    # state < 7, no copy pending, literal blocks requested: the three tests of the general
    # packet head are known
    if rot():
        label("litb")  # isMatch = 0 has been decided against the bound in s82, which is the new range
        head_update(H_IS_MATCH, 0, hoisted=hoist0())
        nchk(rreg="s82")
        plain_literal(range0="s82", loop=False)
    label("pktl")
    # lwait: both ways here end with the next literal's four block reads as their YOUNGEST LDS operations (literal_context);
    # LDS returns in order, so lgkmcnt(4) says the head gather is back while the blocks are still on their way -- their
    # round trip then overlaps the isMatch decision; the literal waits for them in front of its first level (plain_literal)
    emit("s_waitcnt lgkmcnt(%d)" % (4 if "lwait" in VARIANT else 0))
    bounds("v40")
    if "rmov" in VARIANT:
        # the isMatch bound is read into s82 and STAYS there as the new range when the bit is 0: the literal's first
        # level takes it from s82 and writes %[range] itself -- no s_mov per literal (lit8 forms read %[range]: excluded)
        assert "lit8" not in VARIANT and "lit8g" not in VARIANT and "litrun" not in VARIANT
        packet_limits(H_IS_MATCH, breg="s82")
        if rot():
            emit("v_cmp_gt_u32 vcc, s82, v29\ns_cbranch_vccnz %s" % L("litb"))  # falls into sec_match
            return
        hbit(H_IS_MATCH, L("match"), stage=2, breg="s82", keep=True, hoisted=hoist0())
        plain_literal(range0="s82")
        return
    packet_limits(H_IS_MATCH)
    hbit(H_IS_MATCH, L("match"), stage=2, hoisted=hoist0())
    plain_literal(run_entry=L("lrent") if "litrun" in VARIANT else None)


IM0 = "v34"  # literal run: isMatch[state 0][posState], lane = posState


def im0_update(bit):
    """new value of isMatch[0][posState s90] inside IM0 (decompress.go:30 / :177)"""
    emit("v_cmp_eq_u32 vcc, s90, %[vlane]")
    if bit == 0:
        emit("v_add_u32 v63, 0xfffff81f, %s\nv_ashrrev_i32 v63, 5, v63" % IM0)
    else:
        emit("v_ashrrev_i32 v63, 5, %s" % IM0)
    emit("v_sub_u32 v63, %s, v63\nv_cndmask_b32 %s, %s, v63, vcc" % (IM0, IM0, IM0))


def im0_writeback():
    emit("v_cndmask_b32 v60, v38, v56, s[78:79]\nds_write_b16 v60, %s" % IM0)


def sec_literal_run():
    """Runs of literals (state 0: at least three literals since the last match, state.go:153-163).  The only
    head probability such a packet needs is isMatch[0][posState]: the sixteen of them stay in a register
    (lane = posState), so a literal of a run has no head gather, no head write-back and no wait for them.
    Entered from the plain literal of sec_packet_after_literal when the state reaches 0; left to the
    match path (label match2) at the first match, or to the exits."""
    label("lrent")  # the finished literal's model update and the next literal's blocks are still to do
    emit("ds_write_b16 v47, v40")       # the head of the packet that just ended goes back
    emit("v_mov_b32 v47, v38")          # ... and no later exit writes it again (harmless slot)
    emit("ds_read_u16 %s, v56" % IM0)   # isMatch[0][0..15] (P_IS_MATCH = 0), behind the write above
    tree_update_rec(8, "v39")
    literal_context(prev_v="v32")
    label("lrun")
    emit("s_waitcnt lgkmcnt(0)")
    bounds(IM0)
    emit("""
    s_cmp_gt_u32 %%[arel], %%[arel_lim]
    s_cbranch_scc1 %s
    s_and_b32 s90, %%[wpos], %%[pos_mask]
    s_cmp_ge_u32 %%[pos], %%[pos_lim]
    s_cbranch_scc1 %s
    v_readlane_b32 s80, v55, s90
    v_cmp_gt_u32 vcc, s80, v29
    s_cbranch_vccz %s
    s_mov_b32 %%[range], s80
    """ % (L("lrx0"), L("lrx0"), L("lrmatch")))
    im0_update(0)
    nchk()
    walk_rec(8, LIT_BLOCKS)
    emit("""
    s_andn2_b32 %[prev], 0xff, s88
    v_mov_b32 v32, %[prev]
    v_mov_b32 v61, %[pos]
    global_store_byte v61, v32, %[outp]
    s_add_u32 %[pos], %[pos], 1
    """)
    wpos_advance("1")  # the state stays 0 (stateUpdateLiteral)
    tree_update_rec(8, "v39")
    literal_context(prev_v="v32")
    emit("s_branch %s" % L("lrun"))
    label("lrx0")  # a limit at a packet boundary: the register goes back, the ordinary exit does the rest
    im0_writeback()
    emit("s_branch %s" % L("x0"))
    label("lrmatch")  # isMatch = 1: finish that decision here, gather the packet's head, join the match path
    emit("v_subrev_u32 v29, s80, v29\ns_sub_u32 %[range], %[range], s80")
    im0_update(1)
    im0_writeback()
    head_issue(first=True)
    nchk()
    emit("s_waitcnt lgkmcnt(0)")
    bounds("v40")
    emit("s_nop 0\nv_readlane_b32 s80, %s, %d\ns_branch %s" % (head_src(), H_IS_REP, L("match2")))


def sec_packet_general():
    """general packet head, literal after a match (plain or matched)"""
    # ------------------------------------------------------------- packet head
    label("pkt")
    emit("s_waitcnt lgkmcnt(0)")
    bounds("v40")
    packet_limits(H_IS_MATCH, breg=ismatch_reg())
    hbit(H_IS_MATCH, L("match"), stage=2, breg=ismatch_reg(), hoisted=hoist0())
    # ------------------------------------------------------------- literal (decompress.go:44-175)
    if pkm():
        # pkm (round 4): a pending copy means that the packet before was a match, so the state is >= 7 and this literal is a
        # MATCHED literal: the path falls straight into it; the test of the state (and the plain literal behind it) is only
        # reached from the loop's entry, out of line -- two instructions and a taken branch less per literal after a match
        emit("s_cmp_eq_u32 s95, 0\ns_cbranch_scc1 %s" % L("litready"))
        finish_body()
        literal_context()
        emit("s_waitcnt lgkmcnt(0)")

        def entry_literal():
            label("litready")
            emit("s_cmp_ge_u32 %%[state], 7\ns_cbranch_scc1 %s" % L("mlit"))
            plain_literal()
        deferred.append(entry_literal)
        sec_matched_literal()
        return
    if "cflag" in VARIANT:
        # `pkt` is reached from a copy (pending; prevByte unknown, so no literal blocks yet) or from the loop's
        # entry (nothing pending, blocks requested there): one flag says both
        emit("s_cmp_eq_u32 s95, 0\ns_cbranch_scc1 %s" % L("litready"))
        finish_body()
        literal_context()
        emit("s_waitcnt lgkmcnt(0)")
    else:
        need_copy_done()
        emit("s_cmp_lg_u32 s97, 0\ns_cbranch_scc1 %s" % L("litready"))
        literal_context()
        emit("s_waitcnt lgkmcnt(0)")
    label("litready")
    emit("s_cmp_ge_u32 %%[state], 7\ns_cbranch_scc1 %s" % L("mlit"))
    plain_literal()
    sec_matched_literal()


def pkm():
    return "pkm" in VARIANT and "cflag" in VARIANT


def sec_matched_literal():
    # ------------------------------------------------------------- matched literal (:59-114)
    # The matched half of the literal coder lives in HBM (xlz_format.h): ONE gather fetches the
    # eight probabilities the walk meets as long as the decoded bits follow matchByte (lane k =
    # level k: tree slot (0x1ff ^ mb) >> (8 - k), matchBit (mb >> (7 - k)) & 1).  At the first
    # bit that differs the walk carries on in the plain table (:116-165), whose blocks are
    # already in v50..v53.  s98 = levels decided in the matched table.
    label("mlit")
    if mlv():
        # mlv (A/B, xlz_format.h: XLZ_MLIT_VEB): cell = (slot << 1 | matchBit) for levels 0..3, and for levels 4..7 the level-4
        # node's slot as the block number, the node's index inside that subtree and the match bit below it.  v26 = the bit that
        # leads the index inside the subtree (0 for levels 0..3), v27 = a mask of the lanes of levels 4..7.
        emit("""
        s_xor_b32 s89, %[mb], 0x1ff
        s_andn2_b32 s82, s89, 15
        s_lshl_b32 s82, s82, 1
        v_lshrrev_b32 v60, v19, s89
        v_bfe_u32 v61, %[mb], v18, 1
        v_lshl_or_b32 v60, v60, 1, v61
        v_add_u32 v61, -1, v26
        v_and_or_b32 v60, v60, v61, v26
        v_and_or_b32 v60, s82, v27, v60
        v_lshlrev_b32 v60, 1, v60
        v_lshl_add_u32 v57, v39, 2, v60
        global_load_ushort v54, v57, %[mptr] offset:512
        s_waitcnt vmcnt(0)
        """)
    elif ml4():
        # ml4 (round 4): the per-lane shift counts are the loop constants of tree_update_rec (v19 = 8 - level, 31 in the lanes
        # that are no level: their slot is 0; v18 = 7 - level): four instructions instead of nine
        emit("""
        s_xor_b32 s89, %[mb], 0x1ff
        v_lshrrev_b32 v60, v19, s89
        v_bfe_u32 v61, %[mb], v18, 1
        v_lshl_or_b32 v60, v61, 8, v60
        """)
    else:
        emit("""
        v_min_u32 v55, 7, %[vlane]
        s_xor_b32 s89, %[mb], 0x1ff
        v_sub_u32 v60, 8, v55
        v_lshrrev_b32 v60, v60, s89
        v_sub_u32 v61, 7, v55
        v_lshrrev_b32 v61, v61, %[mb]
        v_and_b32 v61, 1, v61
        v_lshl_or_b32 v60, v61, 8, v60
        """)
    if not mlv():
      emit("""
    MLBASE
    global_load_ushort v54, v57, %[mptr] offset:512
    s_waitcnt vmcnt(0)
    """.replace("MLBASE", "v_add_lshl_u32 v57, v60, v39, 1" if "lctx" in VARIANT else
                "v_add_u32 v61, %d, v39\nv_add_lshl_u32 v57, v60, v61, 1" % ((-P_LIT * 2) & 0xffffffff)))
    # matched levels: the eight candidate probabilities are lanes 0..7 of v54, so the bound of
    # level k is lane k of the VALU product (no record needed: v54 itself is the record)
    bounds("v54")
    emit(slot_init() + "\nv_readlane_b32 s80, v55, 0\ns_nop 0")
    for k in range(8):
        decide()
        slot_step(k)
        if k < 7:
            nchk(prefix=lambda: bounds("v54"), pick=head_pick(k + 1, src="v55"))
        else:
            nchk()
        if k < 7:  # still on matchByte's path?  s88 == (0x1ff ^ mb) >> (7 - k)  (complemented bits)
            emit("s_lshr_b32 s82, s89, %d\ns_cmp_lg_u32 s82, s88\ns_cbranch_scc1 %s" % (7 - k, L("mx%d" % (k + 1))))  # s80 holds the next bound
    emit("s_mov_b32 s98, 8\ns_branch %s" % L("mlfin"))
    for k in range(1, 8):
        label("mx%d" % k)
        if "pwhoist" in VARIANT and "rlhoist" in VARIANT and "lit8" not in VARIANT and "lit8g" not in VARIANT:
            emit("s_mov_b32 s98, %d" % k)  # the plain walk's levels expect their probability in s86
            fetch_level(k, LIT_BLOCKS)
            emit("s_branch %s" % L("pwd%d" % k))
        else:
            emit("s_mov_b32 s98, %d\ns_branch %s" % (k, L("pw%d" % k)))
    if "lit8" in VARIANT or "lit8g" in VARIANT:
        walk8(LIT_BLOCKS, entries="pw")
    else:
        walk_rec(8, LIT_BLOCKS, entries="pw")
    label("mlfin")
    if lgather():
        emit("v_cmp_gt_u32 vcc, s98, %[vlane]")
        rec_gather_issue("v39", masked=True)
    literal_tail()
    if "lit8" in VARIANT:  # levels decided in the plain table: their probabilities, gathered; v54 keeps the matched ones
        gather8(LIT_BLOCKS, dst="v33")
        emit("v_cmp_gt_u32 vcc, s98, %[vlane]\nv_cndmask_b32 v54, v33, v54, vcc")
    # lanes < s98 -> matched table (HBM; the rest to its unused slot 0), lanes s98..7 -> plain table
    if not (ml4() and lgather() and "lit8" not in VARIANT):  # (ml4: VCC still holds it -- nothing since mlfin writes VCC)
        emit("v_cmp_gt_u32 vcc, s98, %[vlane]")
    if lgather():
        tree_update_rec(8, "v39", store=False, issued=True, pending=2)
    else:
        tree_update_rec(8, "v39", store=False)
    if lgather():  # (v60 is masked already)
        emit("""
        v_cndmask_b32 v57, 0, v57, vcc
        global_store_short v57, v61, %%[mptr] offset:512
        ds_write_b16 v60, v61%s
        """ % ((" offset:%d" % lit_off("v39")) if lit_off("v39") else ""))
    else:
        off = lit_off("v39")
        emit("""
        v_cndmask_b32 v57, 0, v57, vcc
        global_store_short v57, v61, %%[mptr] offset:512
        v_cndmask_b32 v60, v60, %s, vcc
        G8MASK
        ds_write_b16 v60, v61%s
        """.replace("G8MASK", "" if g8() else "v_cndmask_b32 v60, %s, v60, s[76:77]" % ("0" if off else "v38"))
             % ("0" if off else "v38", (" offset:%d" % off) if off else ""))
    literal_context(prev_v="v32")
    emit("s_branch %s" % L("pktl"))


def sec_match():
    """simple match up to the validity test of the new distance; falls into sec_copy"""
    # ------------------------------------------------------------- match or rep
    label("match")
    r2 = rmov2()
    hbit_one(H_IS_MATCH, next_head=H_IS_REP, breg=ismatch_reg(), pick_dst="s83" if r2 else "s80")
    label("match2")  # (a literal run joins here with its own isMatch decision done)
    len_request(P_LEN)  # speculative (a rep match asks for its own trees): one LDS round trip earlier
    if r2:
        hbit(H_IS_REP, L("rep"), stage=2, breg="s83", keep=True)
    else:
        hbit(H_IS_REP, L("rep"), stage=2)
    # simple match (:215-668)
    bounds("v40", rin="s83" if r2 else "%[range]")  # for the length coder's first decision
    if "vreps" in VARIANT:
        # the four reps live in lanes 0..3 of v34 (lane 0 mirrors the scalar rep0): the shift of :216 is one DPP move,
        # the rotations of a rep match one quad permute + one lane read; rep1..3 go back to SGPRs where the loop is left
        emit("v_mov_b32_dpp v34, v34 row_shr:1 row_mask:0x1 bank_mask:0x1")
    else:
        emit("s_mov_b32 %[rep3], %[rep2]\ns_mov_b32 %[rep2], %[rep1]\ns_mov_b32 %[rep1], %[rep0]")
    if r2:
        emit("v_readlane_b32 s84, %s, %d" % (head_src(), H_LEN_C))
    else:
        len_pick(H_LEN_C)
    emit("ds_read_u16 v35, v56 offset:%d" % (P_ALIGN * 2))
    len_decode("lm", P_LEN, H_LEN_C, H_LEN_C2, posslot=True, chain=("s83", "s84") if r2 else None)  # leaves the posSlot block requested, s92 = its base
    emit("s_waitcnt lgkmcnt(0)")
    walk(6, ["v36"], filler="v_readlane_b32 %[state], v20, %[state]" + ("" if "tu8" in VARIANT else "\nv_mov_b32 v58, s92"))  # stateUpdateMatch as a table
    emit("""
    s_andn2_b32 s98, 63, s88
    s_cmp_lt_u32 s98, 4
    s_cbranch_scc0 %s
    s_mov_b32 %%[rep0], s98
    """ % L("dist"))
    tree_update(6, ["v36"], addr="v57")
    emit("s_branch %s" % L("distdone"))
    label("dist")
    # numDirectBits = (slot >> 1) - 1 and the distance's base (2 | slot & 1) << numDirectBits
    # (:491-492) from two 64-entry tables held in v24 / v25, lane = slot
    emit("""
    v_readlane_b32 s83, v24, s98
    v_readlane_b32 s93, v25, s98
    s_cmp_lt_u32 s98, 14
    s_cbranch_scc0 %s
    s_sub_u32 s84, s93, s98
    s_add_u32 s84, s84, %d
    s_lshl_b32 s92, s84, 1
    v_add_u32 v59, s92, v56
    ds_read_u16 v37, v59
    """ % (L("direct"), P_POS_DEC))
    tree_update(6, ["v36"], addr="v57")  # posSlot tree, while the posDecoders block is on its way
    # reverse bit tree over posDecoders (:495-546): s83 levels (1..5)
    emit("s_waitcnt lgkmcnt(0)")
    walk(5, ["v37"], early_exit=("s98", L("rtdone")), filler=("" if "tu8" in VARIANT else "v_mov_b32 v58, s92\n") + "s_mov_b32 s98, s83")
    label("rtdone")
    tree_update("s98", ["v37"], addr="v59")
    emit("""
    s_not_b32 s80, s88
    s_brev_b32 s80, s80
    s_sub_u32 s81, 32, s98
    s_lshr_b32 s80, s80, s81
    s_add_u32 %%[rep0], s93, s80
    s_branch %s
    """ % L("distdone"))
    label("direct")  # DecodeDirectBits (:549-577)
    tree_update(6, ["v36"], addr="v57")  # posSlot tree
    # numDirectBits - 4 = s83 - 4 (2..26) halvings, unrolled; entered through a branch table so
    # that no loop counter is kept.  Exactly the reference's arithmetic: t = sign(code - range).
    # Only the range stays on the scalar side (its normalisation test needs SCC): the code (v29)
    # and the collected bits (v33 = 2 * acc + (t >= 0)) are wave-uniform VGPR values,
    # four VALU instructions per bit instead of four scalar ones.
    emit("""
    s_getpc_b64 s[80:81]
    s_sub_u32 s82, 35, s83
    s_lshl2_add_u32 s80, s82, s80
    s_addc_u32 s81, s81, 0
    v_mov_b32 v33, DB0
    s_setpc_b64 s[80:81]
    """.replace("DB0", "-1" if "db6" in VARIANT else "0"))  # s_getpc returns the address of the s_sub; the table starts 5 instructions (all 4 bytes) later:
    # entry e = 26 - (s83 - 4) is at +4 * (e + 5) = 4 * (35 - s83)
    for k in range(26, 0, -1):
        emit("s_branch %s" % L("db%d" % k))
    for k in range(26, 0, -1):
        label("db%d" % k)
        if "db6" in VARIANT:
            # db6 (round 4): the subtract's borrow IS the complemented bit and an unsigned min selects the new code (as in
            # decide()): six instructions per bit instead of eight.  v33 collects the borrows below a run of ones (it
            # starts as -1), so that the bits are its complement: rep0's base takes -(v33 << 4) = 16 * bits + 16, and the
            # align table v23 holds its values minus 16.
            emit("""
            s_lshr_b32 %[range], %[range], 1
            v_subrev_co_u32 v55, vcc, %[range], v29
            v_min_u32 v29, v29, v55
            """)
            nchk(mid="v_addc_co_u32 v33, vcc, v33, v33, vcc")
            continue
        emit("""
        s_lshr_b32 %[range], %[range], 1
        v_subrev_u32 v55, %[range], v29
        v_cmp_le_i32 vcc, 0, v55
        """)
        # the normalisation test sits in the VCC wait states; the stub (which shifts v29) is
        # entered after the bit is done
        nchk(mid="s_nop 0\nv_cndmask_b32 v29, v29, v55, vcc\nv_addc_co_u32 v33, vcc, v33, v33, vcc")
    emit("v_readfirstlane_b32 s84, v33\ns_lshl_b32 s84, s84, 4")
    if "pref" in VARIANT:
        # (measured: -1.5 % on text, profiles/r03/ab_pref.txt -- the copy's source latency is not what the waves wait
        #  for.  A build with this switch needs "v26", "v27" in the clobber list of lzma_fast_loop.)
        # pref: everything of the distance but its four align bits is known here, four decisions (~400 cycles) before the copy
        # can ask for its source: one 256-byte load (discarded: v27) pulls the source's cache lines in now -- a match
        # source is a miss of L2 more often than not (4096 live windows of 64 KiB against 32 MiB of L2; VmemLatency 430
        # cycles on text, profiles/r03).  The offset is clamped (unsigned) to the output position: whatever a damaged
        # stream makes of the distance -- beyond the start of the unit, or wrapped far beyond its end -- the load stays
        # inside [out, out + pos + 4).
        assert nopos()
        emit("""
        s_add_u32 s81, s93, s84
        v_subrev_u32 v26, s81, v17
        v_lshl_add_u32 v26, %[vlane], 2, v26
        v_add_u32 v26, -20, v26
        v_min_u32 v26, v17, v26
        global_load_dword v27, v26, %[outp]
        """)  # (v27 is written by nothing else: a prefetch still in flight cannot disturb the next one's address in v26)
    # reverse tree over alignDecoderProbs (:579-625): M = 1 b0 b1 b2 b3
    walk(4, ["v35"], filler="%s s93, s93, s84" % ("s_sub_u32" if "db6" in VARIANT else "s_add_u32")
         + ("" if "tu8" in VARIANT else "\nv_mov_b32 v58, %d" % (P_ALIGN * 2)))
    tree_update(4, ["v35"], addr="v56", off=P_ALIGN * 2, dump="%d" % ((P_LEN + 2) * 2 - P_ALIGN * 2))
    emit("v_readlane_b32 s80, v23, s88\ns_add_u32 %[rep0], s93, s80")  # v23: slot -> the four bits, reversed
    label("distdone")
    if "vreps" in VARIANT:
        emit("v_writelane_b32 v34, %[rep0], 0")
    if cchk():
        # cchk (round 4): ONE window test for a new distance.  rep0 < fill (fill = bytes in the window: window.pos, or
        # dictSize once it is full) says at once that the distance is valid (:651-653: rep0 < dictSize, rep0 <= pos) AND that
        # the copy's source lies inside the window's bytes -- the two tests sec_match and sec_copy made one after the
        # other.  The rare rest (rep0 == fill: the reference's off-by-one distance, window.go:89-91; an invalid
        # distance; the end marker) is told apart out of line.  The copy then only asks whether it is one of its own
        # (shorter than 64 bytes and than its distance).  Four scalar instructions less per simple match.
        emit("""
        s_max_u32 s81, %%[wpos], s94
        s_cmp_lt_u32 %%[rep0], s81
        s_cbranch_scc0 %s
        s_add_u32 s89, s89, 2
        s_add_u32 s93, %%[rep0], 1
        s_min_u32 s80, s93, 64
        s_cmp_lt_u32 s89, s80
        s_cbranch_scc0 %s
        """ % (L("dslow"), L("x3")))  # falls into the copy itself (sec_copy: copygo); the rep paths' full test lies out of line

        def dslow():
            label("dslow")  # rep0 >= fill: valid iff rep0 <= min(fill, dictSize - 1); then the source reaches in front of the window
            emit("""
            s_min_u32 s80, s81, %%[dictm1]
            s_cmp_le_u32 %%[rep0], s80
            s_cbranch_scc0 %s
            s_add_u32 s89, s89, 2
            s_branch %s
            """ % (L("dbad"), L("x3")))
        deferred.append(dslow)
        return
    # :633-653 in one test.  rep0 is valid iff rep0 < dictSize and (window full or rep0 <= window.pos);
    # while the window is not full window.pos = pos - wbase < dictSize, so both say
    # rep0 <= min(pos - wbase, dictSize - 1).  The end marker (rep0 = 0xFFFFFFFF) fails it too
    # and is told apart out of line.
    emit("s_max_u32 s80, %[wpos], s94" if nopos() else "s_sub_u32 s80, %[pos], %[wbase]")
    emit("""
    s_min_u32 s80, s80, %%[dictm1]
    s_cmp_le_u32 %%[rep0], s80
    s_cbranch_scc0 %s
    """ % L("dbad"))
    emit("s_add_u32 s89, s89, 2")  # falls into the copy (the rep paths branch to it)


def sec_rep():
    """rep matches; out of line, ends with a branch to the copy"""
    # ------------------------------------------------------------- rep match (:685-1123)
    label("rep")
    hbit_one(H_IS_REP, next_head=H_G0, breg="s83" if rmov2() else "s80")
    if nopos():
        emit("s_or_b32 s81, %%[wpos], s94\ns_cbranch_scc0 %s" % L("x1"))  # nothing in the window: pos == wbase
    else:
        emit("s_cmp_eq_u32 %%[pos], %%[wbase]\ns_cbranch_scc1 %s" % L("x1"))
    hbit(H_G0, L("g1"), stage=2, next_head=H_REP0_LONG)
    hbit(H_REP0_LONG, L("r0long"), stage=2)
    emit("""
    s_cmp_lt_u32 %%[state], 7
    s_cselect_b32 %%[state], 9, 11
    s_mov_b32 s89, 1
    s_branch %s
    """ % L("copy"))  # short rep: one byte
    label("r0long")
    hbit_one(H_REP0_LONG)
    emit("s_branch %s" % L("replen"))
    label("g1")
    hbit_one(H_G0, next_head=H_G1)
    hbit(H_G1, L("g2"), stage=2)
    if "vreps" in VARIANT:
        emit("v_readlane_b32 %%[rep0], v34, 1\nv_mov_b32_dpp v34, v34 quad_perm:[1,0,2,3] row_mask:0x1 bank_mask:0x1\ns_branch %s"
             % L("replen"))
    else:
        emit("s_mov_b32 s80, %%[rep1]\ns_mov_b32 %%[rep1], %%[rep0]\ns_mov_b32 %%[rep0], s80\ns_branch %s" % L("replen"))
    label("g2")
    hbit_one(H_G1, next_head=H_G2)
    hbit(H_G2, L("g3"), stage=2)
    if "vreps" in VARIANT:
        emit("v_readlane_b32 %%[rep0], v34, 2\nv_mov_b32_dpp v34, v34 quad_perm:[2,0,1,3] row_mask:0x1 bank_mask:0x1\ns_branch %s"
             % L("replen"))
    else:
        emit("""
        s_mov_b32 s80, %%[rep2]
        s_mov_b32 %%[rep2], %%[rep1]
        s_mov_b32 %%[rep1], %%[rep0]
        s_mov_b32 %%[rep0], s80
        s_branch %s
        """ % L("replen"))
    label("g3")
    hbit_one(H_G2)
    if "vreps" in VARIANT:
        emit("v_readlane_b32 %[rep0], v34, 3\nv_mov_b32_dpp v34, v34 quad_perm:[3,0,1,2] row_mask:0x1 bank_mask:0x1")
    else:
        emit("""
        s_mov_b32 s80, %[rep3]
        s_mov_b32 %[rep3], %[rep2]
        s_mov_b32 %[rep2], %[rep1]
        s_mov_b32 %[rep1], %[rep0]
        s_mov_b32 %[rep0], s80
        """)
    label("replen")
    bounds("v40")
    len_request(P_REP_LEN)
    len_pick(H_REP_C)
    len_decode("lr", P_REP_LEN, H_REP_C, H_REP_C2, posslot=False)
    emit("s_cmp_lt_u32 %%[state], 7\ns_cselect_b32 %%[state], 8, 11\ns_add_u32 s89, s89, 2\ns_branch %s" % L("copy"))


def sec_copy():
    """window.CopyMatch of short non-overlapping matches; falls into the general packet head"""
    # ------------------------------------------------------------- window.CopyMatch (window.go:55-87)
    if cchk():  # (the simple match has made its own test and falls into copygo)
        deferred.append(lambda: (copy_test(), emit("s_branch %s" % L("copygo"))))
    else:
        copy_test()
    copy_body()


def cchk():
    return "cchk" in VARIANT and nopos()


def copy_test():
    label("copy")
    # done here if len < 64, len < distance (no overlap) and the source lies inside the window's
    # bytes; one test: the distance, or "too far" when the first two fail, against pos - wbase
    emit("""
    s_add_u32 s93, %%[rep0], 1
    s_min_u32 s80, s93, 64
    s_cmp_lt_u32 s89, s80
    s_cselect_b32 s80, s93, -1
    FILL
    s_cmp_le_u32 s80, s81
    s_cbranch_scc0 %s
    """.replace("FILL", "s_max_u32 s81, %%[wpos], s94" if nopos() else "s_sub_u32 s81, %%[pos], %%[wbase]") % L("x3"))


def copy_body():
    if cchk():
        label("copygo")
    need_copy_done(inline=True)  # the new source may overlap the bytes the pending copy still has to store
    emit("v_add_u32 v48, %s, %%[vlane]\nv_subrev_u32 v61, s93, v48" % ("v17" if nopos() else "%[pos]"))
    if "nocmask" in VARIANT:
        emit("global_load_ubyte v49, v61, %[outp]")
    else:
        # only lanes 0..len fetch (byte len is the next matchByte): a 64-lane gather touches two or three
        # 32-byte sectors of a window that is rarely in L2, a 7-byte one touches one (FETCH_SIZE per launch
        # 43.7 -> 31.8 GB on the bench workload, throughput unchanged)
        emit("""
        s_sub_u32 s80, 63, s89
        s_lshr_b64 exec, -1, s80
        global_load_ubyte v49, v61, %[outp]
        s_mov_b64 exec, -1
        """)
    if "cflag" not in VARIANT:
        emit("s_mov_b32 s94, 1")
    emit("s_mov_b32 s95, s89")
    emit("v_add_u32 v17, s89, v17" if nopos() else "s_add_u32 %[pos], %[pos], s89")
    wpos_advance("s89")
    head_issue()  # next packet's head gather; its literal blocks wait for the copy (prevByte)
    if "cflag" not in VARIANT:
        emit("s_mov_b32 s97, 0")  # falls into the general packet head


def sec_exits():
    """exit codes, completion of pending work, out-of-line stubs"""
    # ------------------------------------------------------------- exits
    label("dbad")
    emit("s_cmp_eq_u32 %%[rep0], -1\ns_cbranch_scc1 %s\ns_branch %s" % (L("x2"), L("x1")))
    label("x3")
    emit("s_mov_b32 %%[lenout], s89\ns_mov_b32 %%[exitc], 3\ns_branch %s" % L("fin"))
    label("x2")
    emit("s_mov_b32 %%[lenout], s89\ns_mov_b32 %%[exitc], 2\ns_branch %s" % L("fin"))
    label("x1")
    emit("s_mov_b32 %%[exitc], 1\ns_branch %s" % L("fin"))
    label("x0")
    emit("s_mov_b32 %[exitc], 0")
    label("fin")
    emit("v_readfirstlane_b32 %[code], v29")
    need_copy_done()
    if "vprev" in VARIANT:
        emit("v_readfirstlane_b32 %[prev], v32" + ("\ns_and_b32 %[prev], %[prev], 0xff" if "lctx" in VARIANT else ""))
    if nopos():
        emit("v_readfirstlane_b32 %[pos], v17")
    if vcur():
        emit("v_readfirstlane_b32 %[cur], v30")
    if "warel" in VARIANT:
        emit("s_sub_u32 s80, 3, s91\ns_lshl2_add_u32 %[arel], s90, s80")
    if "vreps" in VARIANT:
        emit("v_readlane_b32 %[rep1], v34, 1\nv_readlane_b32 %[rep2], v34, 2\nv_readlane_b32 %[rep3], v34, 3")
    emit("s_waitcnt lgkmcnt(0)\nds_write_b16 v47, v40\ns_waitcnt lgkmcnt(0)\ns_branch %s" % L("end"))
    emit_stubs()
    emit_wstubs()
    emit_finish_blocks()
    label("end")





def set_head_lanes():
    """lane of each head probability (xlz_kernel.hip: head_vectors builds the gather addresses for the same lanes)"""
    global H_IS_MATCH, H_IS_REP, H_G0, H_G1, H_G2, H_REP0_LONG, H_LEN_C, H_LEN_C2, H_REP_C, H_REP_C2
    lanes = [hdpp_lane(j) for j in range(10)] if "hdpp" in VARIANT else list(range(10))
    (H_IS_MATCH, H_IS_REP, H_G0, H_G1, H_G2, H_REP0_LONG, H_LEN_C, H_LEN_C2, H_REP_C, H_REP_C2) = lanes


def gen():
    set_head_lanes()
    # The gathers of packet n+1 are issued from the tail of packet n (software pipelining):
    # by the time the loop top has done its limit checks the probabilities have arrived.
    emit("v_lshlrev_b32 v56, 1, %%[vlane]\ns_mov_b32 PEND, 0\nv_mov_b32 v38, %d\nv_mov_b32 v29, %%[code]".replace("PEND", pend()) % ((P_LEN + 2) * 2))
    # per-lane constants of tree_update: v31 = floor(log2(lane)) + 1, v30 = lane (lane 0: never a slot)
    emit("v_cmp_eq_u32 vcc, 0, %[vlane]\nv_ffbh_u32 v31, %[vlane]\nv_sub_u32 v31, 32, v31\n"
         "v_cndmask_b32 v30, %[vlane], -1, vcc")
    # distance tables, lane = posSlot: v24 = (slot >> 1) - 1, v25 = (2 | slot & 1) << v24;
    # v23, lane = final slot of the align tree (1, then the complemented bits b0..b3): the value
    # b0 | b1 << 1 | b2 << 2 | b3 << 3 (reverse tree, bit_tree_decoder.go:42-70)
    emit("""
    v_lshrrev_b32 v24, 1, %[vlane]
    v_add_u32 v24, -1, v24
    v_and_b32 v25, 1, %[vlane]
    v_or_b32 v25, 2, v25
    v_lshlrev_b32 v25, v24, v25
    v_not_b32 v23, %[vlane]
    v_and_b32 v23, 15, v23
    v_bfrev_b32 v23, v23
    v_lshrrev_b32 v23, 28, v23
    """)
    if "db6" in VARIANT:
        assert "pref" not in VARIANT
        emit("v_add_u32 v23, -16, v23")
    if "tuc" in VARIANT:
        emit("v_sub_u32 v14, 3, v31\nv_sub_u32 v15, 4, v31\nv_sub_u32 v16, 6, v31")
    if "tu8" in VARIANT:
        emit("s_movk_i32 s85, -2017")
    if hd2():
        emit("s_movk_i32 s87, 2048\nv_mov_b32 v30, 5")
    if mlv():
        # lane k = level k: v26 = 1 << (k - 3) for k = 4..7, else 0; v27 = 0xffff in lanes 4..7, else 0
        emit("""
        v_add_u32 v26, -3, %[vlane]
        v_lshlrev_b32 v26, v26, 1
        v_add_u32 v27, -4, %[vlane]
        v_cmp_gt_u32 vcc, 4, v27
        v_mov_b32 v27, 0xffff
        s_nop 1
        v_cndmask_b32 v27, 0, v27, vcc
        v_cndmask_b32 v26, 0, v26, vcc
        """)
    if "vnorm" in VARIANT:
        emit("v_mov_b32 v17, 0x1000000")
    if "vperm" in VARIANT or "lctx" in VARIANT:
        emit("v_mov_b32 v13, 0x06050400")
    # constants of tree_update_rec
    emit("v_sub_u32 v19, 8, %[vlane]\nv_sub_u32 v18, 7, %[vlane]\nv_cmp_gt_u32 s[76:77], 8, %[vlane]")
    if g8():
        emit("s_nop 1\nv_cndmask_b32 v19, 31, v19, s[76:77]")
    if "lit8" in VARIANT:
        emit("v_cmp_eq_u32 s[78:79], 6, %[vlane]")  # gather8: lane 6 takes block 1
    if "litrun" in VARIANT:
        emit("v_cmp_gt_u32 s[78:79], 16, %[vlane]")  # the sixteen isMatch[0][posState] lanes
    # v21, lane = raw length 0..7: byte address of posSlot[min(len, 3)]; v20, lane = state:
    # stateUpdateMatch (state.go:165-171)
    emit("""
    v_min_u32 v21, 3, %%[vlane]
    v_lshlrev_b32 v21, 7, v21
    v_add_u32 v21, %d, v21
    v_cmp_gt_u32 vcc, 7, %%[vlane]
    v_mov_b32 v20, 10
    s_nop 1
    v_cndmask_b32 v20, v20, 7, vcc
    """ % (P_POS_SLOT * 2))
    if "flim" in VARIANT:
        if "warel" in VARIANT:
            # word-granular input limit: a packet may start up to 3 bytes later than arel_lim says (>= 29 of the 32
            # bytes of slack remain; a packet reads at most 20, types.go:38)
            assert "litrun" not in VARIANT
            emit("""
            s_ashr_i32 s99, %[arel_lim], 2
            s_lshr_b32 s90, %[arel], 2
            s_and_b32 s91, %[arel], 3
            s_sub_u32 s91, 3, s91
            """)
        else:
            emit("s_mov_b32 s99, %[arel_lim]")
        event_limit()
    if "vreps" in VARIANT:
        assert "litrun" not in VARIANT  # (v34)
        emit("""
        v_mov_b32 v34, %[rep0]
        v_writelane_b32 v34, %[rep1], 1
        v_writelane_b32 v34, %[rep2], 2
        v_writelane_b32 v34, %[rep3], 3
        """)
    if nopos():
        emit("""
        v_mov_b32 v17, %[pos]
        s_sub_u32 s80, %[pos], %[wbase]
        s_cmp_ge_u32 s80, %[dict]
        s_cselect_b32 s94, %[dict], 0
        """)
    head_issue(first=True)
    if "vprev" in VARIANT:
        emit("v_mov_b32 v32, %[prev]")
    if vcur():
        emit("v_mov_b32 v30, %[cur]")
    literal_context()  # no copy is pending on entry: prevByte is valid
    if "cflag" in VARIANT:
        emit("s_branch %s" % L("pkt"))
    else:
        emit("s_mov_b32 s97, 1\ns_branch %s" % L("pkt"))
    # Layout: the match path runs straight through -- match, distance, copy, next packet's head --
    # and everything rarer is out of line (length coder's mid / high trees: deferred blocks; rep
    # matches).  A taken branch costs about as much as a scalar instruction plus a fetch bubble.
    sec_packet_after_literal()
    sec_match()
    sec_copy()
    sec_packet_general()
    sec_rep()
    if "litrun" in VARIANT:
        sec_literal_run()
    for f in deferred:
        f()
    sec_exits()

import hazards  # noqa: E402  (tools/hazards.py: the gfx950 wait states the assembler does not insert in inline asm)

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lzma_amd", "csrc", "xlz_fastpath.inc")


def render():
    """-> (text of xlz_fastpath.inc, final instruction lines, number of s_nop the hazard pass added)"""
    set_layout("compact" in VARIANT)
    gen()
    final, n_nops = hazards.fix(lines, verbose=bool(os.environ.get("XLZ_GEN_VERBOSE")))
    assert not hazards.analyse(final)
    if "bralign" in VARIANT:
        import layout
        good = tuple(int(x) for x in os.environ.get("XLZ_LAYOUT_GOOD", "0,4").split(","))  # (experiments)
        targets = []
        for v in VARIANT:  # (experiments) stubN: normalisation stubs on N-byte boundaries, headN: the out-of-line blocks
            if v.startswith("stub"):
                targets.append((r"\.Ln\d+_%=:$", int(v[4:])))
            if v.startswith("head"):
                targets.append((r"\.L(match|mlit|rep|lmc2|lmhi|lrc2|lrhi|g1|g2|g3|r0long|dist|direct)_%=:$", int(v[4:])))
        final, promos, nops, dead = layout.align_branches(final, good=good, targets=targets)
        assert not hazards.analyse(final)
        n, bad = layout.report(final)
        print("layout: %d conditional branches, %d re-encoded instructions, %d s_nop (+%d never executed), %d left in an upper half"
              % (n, promos, nops, dead, bad))
    text = ["// GENERATED by tools/gen_fastpath.py -- do not edit.  %d instructions, %d normalisation stubs.\n"
            % (sum(1 for l in final if not l.endswith(":")), len(stubs)),
            "// Included inside lzma_fast_loop() in xlz_kernel.hip as the body of one asm volatile statement.\n"]
    for l in final:
        sep = "\\n" if l.endswith(":") else "\\n\\t"
        text.append('    "%s%s"\n' % (l, sep))
    return "".join(text), final, n_nops


if __name__ == "__main__":
    # dev: --variant a,b / --without a,b select code paths (VARIANT), --out writes another .inc for an A/B build
    # (a build without hdpp also needs -DXLZ_HEAD_PLAIN for xlz_kernel.hip)
    if "--variant" in sys.argv:
        VARIANT.update(v for v in sys.argv[sys.argv.index("--variant") + 1].split(",") if v)
    if "--out" in sys.argv:
        OUT = sys.argv[sys.argv.index("--out") + 1]
    if "--without" in sys.argv:
        VARIANT.difference_update(sys.argv[sys.argv.index("--without") + 1].split(","))
    if "next" in VARIANT:
        VARIANT.discard("next")
        VARIANT.update(NEXT_VARIANT)
    text, final, n_nops = render()
    with open(OUT, "w") as f:
        f.write(text)
    print("hazards: %d s_nop inserted" % n_nops)
    print(OUT, len(final), "lines,", len(stubs), "stubs")
