#!/usr/bin/env python3
"""Generates lzma_amd/csrc/xlz_fastpath.inc: the hand-scheduled gfx950 fast loop of the
LZMA packet decoder as ONE inline-asm statement (run from the repo root).

Why generated: ~50 decision sites each need the same 11-instruction core, a look-ahead
variant inside bit trees, and an out-of-line normalisation stub with its own labels.  Writing
that by hand invites typos; this script is the source, the .inc file is committed next to it.

The loop decodes whole packets (decompress.go:13 ff.) while
    arel <= arel_lim   (>= 32 readable bytes left in the 256-byte input window)
    pos  <  pos_lim    (>= 288 bytes of output room and of bytesLeft)
and leaves with an exit code:
    0  a limit was reached at a packet boundary (caller refills the window / re-checks)
    1  ErrResultError condition (bad distance, rep match on an empty window)
    2  distance 0xFFFFFFFF decoded (end marker): caller applies decompress.go:633-645
    3  a match copy the loop does not do itself (len >= 64, overlapping, or reaching in
       front of the dictionary epoch): caller copies `lenout` bytes and carries on
State mutations happen in the reference's order; normalisation order is irrelevant here
because no input exhaustion is possible inside the loop.

Register conventions (fixed temporaries, declared as clobbers):
  s80,s81 core temps   s82 address temp   s83,s84 temps   s86 P (probability)   s87 BIT
  s88 M (tree index / symbol)   s89 LEN   s90 posState   s91 state2   s92 tree base (bytes)
  s93 dist   s94 copy pending   s95 its length   v48 its destination   v49 its bytes
  v50..v53 tree blocks   v54 probabilities met on a walk   v55 temp   s85 2M+1
  v56 2*lane   v57 fbit address   v58 tree base (uniform)   v59 tree base + 2*lane
  v60 write-back address / copy dst   v61 children address / copy src   v62 loaded   v63 new prob
"""
import os

# probability-table layout: must match xlz_format.h (checked by static_asserts in the .hip)
P_IS_MATCH, P_IS_REP, P_IS_REP_G0, P_IS_REP_G1, P_IS_REP_G2, P_IS_REP0_LONG = 0, 192, 204, 216, 228, 240
P_POS_SLOT, P_POS_DEC, P_ALIGN, P_LEN, P_REP_LEN, P_LIT = 432, 688, 804, 820, 1336, 1852
LEN_CHOICE, LEN_CHOICE2, LEN_LOW, LEN_MID, LEN_HIGH = 0, 1, 4, 132, 260

lines = []
stubs = []
uid = [0]


def emit(s):
    for l in s.strip("\n").split("\n"):
        l = l.strip()
        if l:
            lines.append(l)


def label(name):
    lines.append(".L%s_%%=:" % name)


def L(name):
    return ".L%s_%%=" % name


def core():
    emit("""
    s_lshr_b32 s80, %[range], 11
    s_mul_i32 s80, s80, s86
    s_sub_u32 s81, %[range], s80
    s_sub_u32 s87, %[code], s80
    s_cselect_b32 %[range], s80, s81
    s_cselect_b32 %[code], %[code], s87
    s_cselect_b32 s81, 0x7e1, 0
    s_cselect_b32 s87, 0, 1
    s_sub_u32 s81, s86, s81
    v_ashrrev_i32 v63, 5, s81
    v_sub_u32 v63, s86, v63
    """)


def nchk():
    """normalisation test; the stub is emitted out of line at the end of the block"""
    uid[0] += 1
    k = "n%d" % uid[0]
    emit("s_lshr_b32 s80, %%[range], 24\ns_cbranch_scc0 %s" % L(k))
    label(k + "b")
    stubs.append(k)


def emit_stubs():
    for k in stubs:
        label(k)
        emit("""
        s_lshl_b32 %%[range], %%[range], 8
        s_lshl_b32 %%[code], %%[code], 8
        s_and_b32 s80, %%[cur], 0xff
        s_or_b32 %%[code], %%[code], s80
        s_lshr_b32 %%[cur], %%[cur], 8
        s_add_u32 %%[arel], %%[arel], 1
        s_and_b32 s80, %%[arel], 3
        s_cbranch_scc1 %s
        s_lshr_b32 s80, %%[arel], 2
        v_readlane_b32 %%[cur], %%[vin], s80
        s_branch %s
        """ % (L(k + "b"), L(k + "b")))


def fbit(addr_sgpr):
    """one decision on the prob at LDS byte address in addr_sgpr; result in s87"""
    emit("v_mov_b32 v57, %s\nds_read_u16 v62, v57\ns_waitcnt lgkmcnt(0)\nv_readfirstlane_b32 s86, v62" % addr_sgpr)
    core()
    emit("ds_write_b16 v57, v63")
    nchk()


def fbit_const(prob_index):
    emit("s_movk_i32 s82, %d" % (prob_index * 2))
    fbit("s82")


def walk_core():
    """decision core without bit / probability update: SCC = (code < bound) on exit of the
    two subtractions, range and code selected; the tree index is advanced by the caller"""
    emit("""
    s_lshr_b32 s80, %[range], 11
    s_mul_i32 s80, s80, s86
    s_sub_u32 s81, %[range], s80
    s_sub_u32 s87, %[code], s80
    s_cselect_b32 %[range], s80, s81
    s_cselect_b32 %[code], %[code], s87
    """)


def tree_update(nb):
    """Apply the model updates of a finished tree walk in ONE vector operation.
    s88 = final index (leading 1 + nb decided bits), v54 lane k = probability seen at level k,
    v58 = tree base address.  nb: int, or the name of an SGPR holding the level count.
    Lane k: node = s88 >> (nb-k), bit = (s88 >> (nb-k-1)) & 1,
            new = p - ((p - (bit ? 0 : 2017)) >>a 5)   (decompress.go:30 / :177)."""
    emit("""
    v_sub_u32 v55, %s, %%[vlane]
    v_lshrrev_b32 v60, v55, s88
    v_add_u32 v61, -1, v55
    v_lshrrev_b32 v61, v61, s88
    v_and_b32 v61, 1, v61
    v_lshl_add_u32 v60, v60, 1, v58
    v_mul_u32_u24 v61, 0x7e1, v61
    v_sub_u32 v61, 0x7e1, v61
    v_sub_u32 v61, v54, v61
    v_ashrrev_i32 v61, 5, v61
    v_sub_u32 v61, v54, v61
    """ % nb)
    if isinstance(nb, int):
        emit("s_mov_b64 exec, %d" % ((1 << nb) - 1))
    else:
        emit("s_bfm_b64 exec, %s, 0" % nb)
    emit("ds_write_b16 v60, v61\ns_mov_b64 exec, -1")


def tree(nbits):
    """Bit tree of nbits levels rooted at LDS byte address s92 (bit_tree_decoder.go:18-40).
    One LDS read fetches a whole 64-prob block of the tree (lane j = node 64b + j); a level is
    then nine scalar instructions plus `v_readlane p, block, M` -- the lane select IS the node
    index, so neither the decoded bit nor an address is ever materialised.  The probabilities
    met on the way are parked in v54 (v_writelane) and updated together by tree_update."""
    emit("v_add_u32 v59, s92, v56\nds_read_u16 v50, v59")
    if nbits == 8:
        emit("ds_read_u16 v51, v59 offset:128\nds_read_u16 v52, v59 offset:256\nds_read_u16 v53, v59 offset:384")
    emit("v_mov_b32 v58, s92\ns_mov_b32 s88, 1\ns_waitcnt lgkmcnt(0)\nv_readlane_b32 s86, v50, 1")
    for k in range(nbits):
        emit("v_writelane_b32 v54, s86, %d\ns_lshl1_add_u32 s85, s88, 1" % k)
        walk_core()
        emit("s_subb_u32 s88, s85, 0")  # M = 2M + 1 - SCC = 2M + bit
        nchk()
        if k + 1 < nbits:
            if k + 1 <= 5:
                emit("v_readlane_b32 s86, v50, s88")
            elif k + 1 == 6:
                emit("v_readlane_b32 s86, v51, s88")
            else:  # node 128..255: block 2 or 3 by bit 6 of M
                emit("v_readlane_b32 s86, v52, s88\nv_readlane_b32 s84, v53, s88\ns_bitcmp1_b32 s88, 6\ns_cselect_b32 s86, s84, s86")
    tree_update(nbits)


def level(last=False):
    """one level with child look-ahead and immediate update (used by the matched-literal tail)"""
    if not last:
        emit("v_lshl_add_u32 v61, s88, 2, v59\nds_read_u16 v62, v61")
    core()
    emit("v_lshl_add_u32 v60, s88, 1, v58\nds_write_b16 v60, v63\ns_lshl1_add_u32 s88, s88, s87")
    nchk()
    if not last:
        emit("s_waitcnt lgkmcnt(0)\nv_readlane_b32 s86, v62, s87")


def len_decode(tag, base):
    """lenDecoder.Decode (len_decoder.go:34-60): raw length -> s89.  posState in s90."""
    fbit_const(base + LEN_CHOICE)
    emit("s_cmp_lg_u32 s87, 0\ns_cbranch_scc1 %s" % L(tag + "c2"))
    emit("s_lshl_b32 s92, s90, 4\ns_add_u32 s92, s92, %d" % ((base + LEN_LOW) * 2))
    tree(3)
    emit("s_sub_u32 s89, s88, 8\ns_branch %s" % L(tag + "end"))
    label(tag + "c2")
    fbit_const(base + LEN_CHOICE2)
    emit("s_cmp_lg_u32 s87, 0\ns_cbranch_scc1 %s" % L(tag + "hi"))
    emit("s_lshl_b32 s92, s90, 4\ns_add_u32 s92, s92, %d" % ((base + LEN_MID) * 2))
    tree(3)
    emit("s_mov_b32 s89, s88\ns_branch %s" % L(tag + "end"))
    label(tag + "hi")
    emit("s_movk_i32 s92, %d" % ((base + LEN_HIGH) * 2))
    tree(8)
    emit("s_sub_u32 s89, s88, 240")
    label(tag + "end")


finish_sites = []


def need_copy_done():
    """The previous match copy's load is left in flight while the next packet decodes; whoever
    needs its bytes (prevByte / matchByte for a literal, the next copy, any exit) comes here
    first.  s94 = copy pending, v48 = its destination offsets, v49 = loaded bytes, s95 = len."""
    uid[0] += 1
    k = "f%d" % uid[0]
    emit("s_cmp_eq_u32 s94, 0\ns_cbranch_scc0 %s" % L(k))
    label(k + "b")
    finish_sites.append(k)


def emit_finish_blocks():
    for k in finish_sites:
        label(k)
        emit("""
        s_waitcnt vmcnt(0)
        global_store_byte v48, v49, %%[outp]
        s_sub_u32 s80, s95, 1
        v_readlane_b32 %%[prev], v49, s80
        v_readlane_b32 %%[mb], v49, s95
        s_mov_b32 s94, 0
        s_branch %s
        """ % L(k + "b"))


def wpos_advance(amount):
    emit("""
    s_add_u32 %%[wpos], %%[wpos], %s
    s_cmp_ge_u32 %%[wpos], %%[dict]
    s_cselect_b32 s80, %%[dict], 0
    s_sub_u32 %%[wpos], %%[wpos], s80
    """ % amount)


def gen():
    emit("v_lshlrev_b32 v56, 1, %[vlane]\ns_mov_b32 s94, 0")
    # ------------------------------------------------------------- packet head
    label("pkt")
    emit("""
    s_cmp_gt_u32 %%[arel], %%[arel_lim]
    s_cbranch_scc1 %s
    s_cmp_ge_u32 %%[pos], %%[pos_lim]
    s_cbranch_scc1 %s
    s_and_b32 s90, %%[wpos], %%[pos_mask]
    s_lshl_b32 s91, %%[state], 4
    s_add_u32 s91, s91, s90
    s_lshl_b32 s82, s91, 1
    """ % (L("x0"), L("x0")))
    fbit("s82")  # isMatch[state2]  (P_IS_MATCH == 0)
    emit("s_cmp_lg_u32 s87, 0\ns_cbranch_scc1 %s" % L("match"))
    # ------------------------------------------------------------- literal (decompress.go:44-175)
    need_copy_done()
    emit("""
    s_and_b32 s83, %%[wpos], %%[lp_mask]
    s_lshl_b32 s83, s83, %%[lc]
    s_sub_u32 s84, 8, %%[lc]
    s_lshr_b32 s84, %%[prev], s84
    s_add_u32 s83, s83, s84
    s_mulk_i32 s83, 0x600
    s_add_u32 s92, s83, %d
    s_cmp_ge_u32 %%[state], 7
    s_cbranch_scc1 %s
    """ % (P_LIT * 2, L("mlit")))
    tree(8)
    label("litdone")
    emit("""
    s_and_b32 %[prev], s88, 0xff
    v_mov_b32 v60, %[prev]
    v_mov_b32 v61, %[pos]
    global_store_byte v61, v60, %[outp]
    s_add_u32 %[pos], %[pos], 1
    """)
    wpos_advance("1")
    emit("""
    s_cmp_lt_u32 %%[state], 10
    s_cselect_b32 s80, 3, 6
    s_sub_u32 s80, %%[state], s80
    s_cmp_lt_u32 %%[state], 4
    s_cselect_b32 %%[state], 0, s80
    s_branch %s
    """ % L("pkt"))
    # ------------------------------------------------------------- matched literal (:59-114)
    label("mlit")
    emit("s_mov_b32 s88, 1\ns_mov_b32 s89, %[mb]")
    label("ml")
    emit("""
    s_bfe_u32 s83, s89, 0x10007
    s_lshl_b32 s89, s89, 1
    s_add_u32 s84, s83, 1
    s_lshl_b32 s84, s84, 9
    s_add_u32 s84, s84, s92
    s_lshl_b32 s82, s88, 1
    s_add_u32 s82, s82, s84
    """)
    fbit("s82")
    emit("""
    s_lshl1_add_u32 s88, s88, s87
    s_cmp_lg_u32 s83, s87
    s_cbranch_scc1 %s
    s_cmpk_lt_u32 s88, 0x100
    s_cbranch_scc1 %s
    s_branch %s
    """ % (L("mlrest"), L("ml"), L("litdone")))
    label("mlrest")
    emit("""
    s_cmpk_lt_u32 s88, 0x100
    s_cbranch_scc0 %s
    v_mov_b32 v58, s92
    v_add_u32 v59, s92, v56
    v_lshl_add_u32 v60, s88, 1, v58
    ds_read_u16 v62, v60
    s_waitcnt lgkmcnt(0)
    v_readfirstlane_b32 s86, v62
    """ % L("litdone"))
    label("mlr")
    level()
    emit("s_cmpk_lt_u32 s88, 0x100\ns_cbranch_scc1 %s\ns_branch %s" % (L("mlr"), L("litdone")))
    # ------------------------------------------------------------- match or rep
    label("match")
    emit("s_add_u32 s82, %%[state], %d\ns_lshl_b32 s82, s82, 1" % P_IS_REP)
    fbit("s82")
    emit("s_cmp_lg_u32 s87, 0\ns_cbranch_scc1 %s" % L("rep"))
    # simple match (:215-668)
    emit("s_mov_b32 %[rep3], %[rep2]\ns_mov_b32 %[rep2], %[rep1]\ns_mov_b32 %[rep1], %[rep0]")
    len_decode("lm", P_LEN)
    emit("""
    s_cmp_lt_u32 %%[state], 7
    s_cselect_b32 %%[state], 7, 10
    s_min_u32 s83, s89, 3
    s_lshl_b32 s83, s83, 7
    s_add_u32 s92, s83, %d
    """ % (P_POS_SLOT * 2))
    tree(6)
    emit("""
    s_sub_u32 s88, s88, 64
    s_cmp_lt_u32 s88, 4
    s_cbranch_scc0 %s
    s_mov_b32 %%[rep0], s88
    s_branch %s
    """ % (L("dist"), L("distdone")))
    label("dist")
    emit("""
    s_lshr_b32 s83, s88, 1
    s_sub_u32 s83, s83, 1
    s_and_b32 s84, s88, 1
    s_or_b32 s84, s84, 2
    s_lshl_b32 s93, s84, s83
    s_cmp_lt_u32 s88, 14
    s_cbranch_scc0 %s
    s_sub_u32 s84, s93, s88
    s_add_u32 s84, s84, %d
    s_lshl_b32 s92, s84, 1
    """ % (L("direct"), P_POS_DEC))
    # reverse bit tree over posDecoders (:495-546): s83 levels (1..5) from base s92; block walk,
    # unrolled with an early exit (v_writelane cannot take two different SGPRs)
    emit("v_add_u32 v59, s92, v56\nds_read_u16 v50, v59\nv_mov_b32 v58, s92\ns_mov_b32 s88, 1\ns_waitcnt lgkmcnt(0)\nv_readlane_b32 s86, v50, 1")
    for k in range(5):
        emit("v_writelane_b32 v54, s86, %d\ns_lshl1_add_u32 s85, s88, 1" % k)
        walk_core()
        emit("s_subb_u32 s88, s85, 0")
        nchk()
        if k < 4:
            emit("s_cmp_eq_u32 s83, %d\ns_cbranch_scc1 %s\nv_readlane_b32 s86, v50, s88" % (k + 1, L("rtdone")))
    label("rtdone")
    tree_update("s83")
    # symbol = the s83 decided bits of M in reverse order (bit_tree_decoder.go:42-70)
    emit("""
    s_brev_b32 s80, s88
    s_sub_u32 s81, 32, s83
    s_lshr_b32 s80, s80, s81
    s_add_u32 %%[rep0], s93, s80
    s_branch %s
    """ % L("distdone"))
    label("direct")  # DecodeDirectBits (:549-577)
    emit("s_sub_u32 s83, s83, 4\ns_mov_b32 s84, 0")
    label("db")
    emit("""
    s_lshr_b32 %[range], %[range], 1
    s_sub_u32 %[code], %[code], %[range]
    s_ashr_i32 s80, %[code], 31
    s_and_b32 s81, %[range], s80
    s_add_u32 %[code], %[code], s81
    s_lshl_b32 s84, s84, 1
    s_add_u32 s84, s84, s80
    s_add_u32 s84, s84, 1
    """)
    nchk()
    emit("""
    s_sub_u32 s83, s83, 1
    s_cmp_lg_u32 s83, 0
    s_cbranch_scc1 %s
    s_lshl_b32 s84, s84, 4
    s_add_u32 s93, s93, s84
    s_movk_i32 s92, %d
    """ % (L("db"), P_ALIGN * 2))
    tree(4)  # reverse tree over alignDecoderProbs (:579-625): M = 1 b0 b1 b2 b3
    emit("s_brev_b32 s80, s88\ns_lshr_b32 s80, s80, 28\ns_add_u32 %[rep0], s93, s80")
    label("distdone")
    emit("""
    s_cmp_eq_u32 %%[rep0], -1
    s_cbranch_scc1 %s
    s_cmp_ge_u32 %%[rep0], %%[dict]
    s_cbranch_scc1 %s
    s_sub_u32 s80, %%[pos], %%[wbase]
    s_cmp_ge_u32 s80, %%[dict]
    s_cbranch_scc1 %s
    s_cmp_le_u32 %%[rep0], %%[wpos]
    s_cbranch_scc0 %s
    """ % (L("x2"), L("x1"), L("dok"), L("x1")))
    label("dok")
    emit("s_add_u32 s89, s89, 2\ns_branch %s" % L("copy"))
    # ------------------------------------------------------------- rep match (:685-1123)
    label("rep")
    emit("s_cmp_eq_u32 %%[pos], %%[wbase]\ns_cbranch_scc1 %s" % L("x1"))
    emit("s_add_u32 s82, %%[state], %d\ns_lshl_b32 s82, s82, 1" % P_IS_REP_G0)
    fbit("s82")
    emit("s_cmp_lg_u32 s87, 0\ns_cbranch_scc1 %s" % L("g1"))
    emit("s_add_u32 s82, s91, %d\ns_lshl_b32 s82, s82, 1" % P_IS_REP0_LONG)
    fbit("s82")
    emit("s_cmp_lg_u32 s87, 0\ns_cbranch_scc1 %s" % L("replen"))
    emit("""
    s_cmp_lt_u32 %%[state], 7
    s_cselect_b32 %%[state], 9, 11
    s_mov_b32 s89, 1
    s_branch %s
    """ % L("copy"))  # short rep: one byte
    label("g1")
    emit("s_add_u32 s82, %%[state], %d\ns_lshl_b32 s82, s82, 1" % P_IS_REP_G1)
    fbit("s82")
    emit("s_cmp_lg_u32 s87, 0\ns_cbranch_scc1 %s" % L("g2"))
    emit("s_mov_b32 s80, %%[rep1]\ns_mov_b32 %%[rep1], %%[rep0]\ns_mov_b32 %%[rep0], s80\ns_branch %s" % L("replen"))
    label("g2")
    emit("s_add_u32 s82, %%[state], %d\ns_lshl_b32 s82, s82, 1" % P_IS_REP_G2)
    fbit("s82")
    emit("s_cmp_lg_u32 s87, 0\ns_cbranch_scc1 %s" % L("g3"))
    emit("""
    s_mov_b32 s80, %%[rep2]
    s_mov_b32 %%[rep2], %%[rep1]
    s_mov_b32 %%[rep1], %%[rep0]
    s_mov_b32 %%[rep0], s80
    s_branch %s
    """ % L("replen"))
    label("g3")
    emit("""
    s_mov_b32 s80, %[rep3]
    s_mov_b32 %[rep3], %[rep2]
    s_mov_b32 %[rep2], %[rep1]
    s_mov_b32 %[rep1], %[rep0]
    s_mov_b32 %[rep0], s80
    """)
    label("replen")
    len_decode("lr", P_REP_LEN)
    emit("s_cmp_lt_u32 %[state], 7\ns_cselect_b32 %[state], 8, 11\ns_add_u32 s89, s89, 2")
    # ------------------------------------------------------------- window.CopyMatch (window.go:55-87)
    label("copy")
    emit("""
    s_add_u32 s93, %%[rep0], 1
    s_cmp_eq_u32 s93, 0
    s_cselect_b32 s93, %%[dict], s93
    s_cmp_ge_u32 s89, 64
    s_cbranch_scc1 %s
    s_cmp_le_u32 s93, s89
    s_cbranch_scc1 %s
    s_sub_u32 s80, %%[pos], %%[wbase]
    s_cmp_lt_u32 s80, s93
    s_cbranch_scc1 %s
    """ % (L("x3"), L("x3"), L("x3")))
    need_copy_done()  # the new source may overlap the bytes the pending copy still has to store
    emit("""
    v_add_u32 v48, %[pos], %[vlane]
    v_subrev_u32 v61, s93, v48
    global_load_ubyte v49, v61, %[outp]
    s_mov_b32 s94, 1
    s_mov_b32 s95, s89
    s_add_u32 %[pos], %[pos], s89
    """)
    wpos_advance("s89")
    emit("s_branch %s" % L("pkt"))
    # ------------------------------------------------------------- exits
    label("x3")
    emit("s_mov_b32 %%[lenout], s89\ns_mov_b32 %%[exitc], 3\ns_branch %s" % L("fin"))
    label("x2")
    emit("s_mov_b32 %%[lenout], s89\ns_mov_b32 %%[exitc], 2\ns_branch %s" % L("fin"))
    label("x1")
    emit("s_mov_b32 %%[exitc], 1\ns_branch %s" % L("fin"))
    label("x0")
    emit("s_mov_b32 %[exitc], 0")
    label("fin")
    need_copy_done()
    emit("s_branch %s" % L("end"))
    emit_stubs()
    emit_finish_blocks()
    label("end")


gen()
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lzma_amd", "csrc", "xlz_fastpath.inc")
with open(out, "w") as f:
    f.write("// GENERATED by tools/gen_fastpath.py -- do not edit.  %d instructions, %d normalisation stubs.\n"
            % (sum(1 for l in lines if not l.endswith(":")), len(stubs)))
    f.write("// Included inside lzma_fast_loop() in xlz_kernel.hip as the body of one asm volatile statement.\n")
    for l in lines:
        sep = "\\n" if l.endswith(":") else "\\n\\t"
        f.write('    "%s%s"\n' % (l, sep))
print(out, len(lines), "lines,", len(stubs), "stubs")
