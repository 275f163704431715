"""CPU: the COMMITTED fast loop (lzma_amd/csrc/xlz_fastpath.inc as tools/gen_fastpath.py renders it) executed
instruction by instruction by tools/gcn_emu.py on real LZMA1 streams -- head gather, tree walks on lane selects, the
LDS gather of the literal walk, DPP head updates, matched literals through the HBM table, deferred match copies, the
direct-bit branch table, normalisation stubs, the folded limit tests -- against the oracle: every decoded byte, and
range / code / state / reps where the loop hands back to the checked path.  The C++ around the loop (lzma_run: window
refills, the copies the loop leaves to its caller, the end of the stream) is restated here in a few lines of Python.

This is the part of the product no other CPU test reaches (there is no GPU in CI); on the GPU box the same loop runs
for real in the -m gpu tests."""
import importlib.util
import os
import sys

import numpy as np
import pytest

import corpus
import lzma_pydec
import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

# (table bases: tools/gen_fastpath.py: model_layout = xlz_format.h: ModelLayout; the rep-length high tree is the first 256
#  entries of the model's HBM part)
K_IN_WINDOW, K_FAST_INPUT, K_FAST_OUTPUT = 256, 32, 128   # (xlz_kernel.hip: kInWindow, kFastInput, kFastOutput)


def _render(add=(), remove=()):
    spec = importlib.util.spec_from_file_location("gen_fastpath_emu", os.path.join(ROOT, "tools", "gen_fastpath.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    g.VARIANT.update(add)
    g.VARIANT.difference_update(remove)
    _, final, _ = g.render()
    import layout
    prog = Program((final, layout.sizes(final), layout), set(g.VARIANT))
    prog.lay = g.model_layout("compact" in g.VARIANT)
    return prog


class Program(tuple):
    """(instructions, their sizes, the layout module) + the generator switches the loop was made with: some of them
    change what the C++ around the loop passes in (xlz_kernel.hip: lzma_fast_loop), and so what this harness passes"""

    def __new__(cls, items, variant):
        self = super().__new__(cls, items)
        self.variant = variant
        self.lay = None   # table bases of the layout the loop was rendered over (_render sets it)
        return self


@pytest.fixture(scope="module")
def program():
    return _render()


def _head_vectors(lane, dpp=True, lay=None):
    """xlz_kernel.hip: head_vectors<L> (DPP cells; dpp=False: the XLZ_HEAD_PLAIN build, head probability j at lane j)"""
    P_LEN = lay["P_LEN"]
    base_of = [lay["P_IS_MATCH"], lay["P_IS_REP"], lay["P_IS_REP_G0"], lay["P_IS_REP_G1"], lay["P_IS_REP_G2"], lay["P_IS_REP0_LONG"],
               P_LEN + 0, P_LEN + 1, lay["P_REP_LEN"] + 0, lay["P_REP_LEN"] + 1]
    hc, hms, hm2, litnext = [], [], [], []
    for l in range(64):
        if dpp:
            hj = (l // 16) * 4 + (l % 16) // 4 if (l % 4 == 0 and l < 40) else 10
        else:
            hj = min(l, 10)
        base = base_of[hj] if hj < 10 else P_LEN + 2
        hc.append(base * 2)
        hms.append(2 if 1 <= hj <= 4 else 0)
        hm2.append(2 if hj in (0, 5) else 0)
        s = l if l < 12 else 0
        litnext.append(0 if s < 4 else (s - 3 if s < 10 else s - 6))
    return [np.array(x, dtype=np.uint32) for x in (hc, hms, hm2, litnext)]


def run_fast_loop(program, payload, lc, lp, pb, dict_size, size, expect, dpp=True, strict_waits=False, base=0):
    """-> (bytes decoded by the emulated loop, machine, number of loop entries, exits by code)"""
    from gcn_emu import Machine
    final, sizes, layout = program
    m = Machine(final, sizes, layout.SGPR_OPS, layout.VGPR_OPS, layout.SGPR64_OPS)
    m.strict_waits = strict_waits
    lay = program.lay
    assert pb <= (2 if lay["POS_STATES"] == 4 else 4), "the compact layout has room for four posStates"
    n_probs = lay["P_LIT"] + (0x100 << (lc + lp))
    m.lds[0:2 * n_probs:2] = 0x00
    m.lds[1:2 * n_probs:2] = 0x04          # every probability 1024 (state.go:79-121)
    out = bytearray(size + 1024)
    out[:base] = expect[:base]              # (base > 0: an LZMA2 unit whose dictionary epoch starts at `base`)
    mp = bytearray(b"\x00\x04" * (256 + ((0x400 if "mlv" in getattr(program, "variant", ()) else 0x200) << (lc + lp))))
    m.mem["outp"], m.mem["mptr"] = out, mp
    lane = np.arange(64, dtype=np.uint32)
    hc, hms, hm2, litnext = _head_vectors(lane, dpp, lay)
    variant = getattr(program, "variant", ())
    if "hiss" in variant:       # head address = hc + state * (hms + posStates hm2) + posState * hm2
        hms = hms + lay["POS_STATES"] * hm2
    m.v.update(vlane=lane, vhc=hc, vhms=hms, vhm2=hm2, vlitnext=litnext,
               vlpm=np.full(64, (lc + lp) if "lctx" in variant else (1 << lp) - 1, dtype=np.uint32),   # lctx: the bit field's width
               vpm=np.full(64, (1 << pb) - 1, dtype=np.uint32))
    s = m.s
    s.update(range=0xFFFFFFFF, code=int.from_bytes(payload[1:5], "big"), state=0, rep0=0, rep1=0, rep2=0, rep3=0,
             pos=base, wpos=0, prev=0, mb=0, exitc=0, lenout=0, dict=dict_size, dictm1=dict_size - 1,
             pos_mask=(1 << pb) - 1, lc=lc, lc8=8 - lc, wbase=base)
    assert payload[0] == 0
    p = 5                                   # input position (rc.Init took five bytes, range_decoder.go:27-46)
    data = payload + b"\0" * 512
    entries, exits = 0, {0: 0, 1: 0, 2: 0, 3: 0}
    while True:
        in_left = len(payload) - p
        if in_left < K_FAST_INPUT or size - s["pos"] < K_FAST_OUTPUT:
            break                           # the checked path takes over here (lzma_run)
        win0, arel = p & ~3, p & 3
        m.v["vin"] = np.frombuffer(data[win0:win0 + 256], dtype="<u4").astype(np.uint32)
        s["cur"] = int(m.v["vin"][0]) >> (8 * arel)
        s["arel"] = arel
        s["arel_lim"] = min(K_IN_WINDOW - K_FAST_INPUT, arel + (in_left - K_FAST_INPUT))
        s["pos_lim"] = s["pos"] + (size - s["pos"] - K_FAST_OUTPUT) + 1
        pos0 = s["pos"]
        m.run()
        entries += 1
        ec = s["exitc"]
        exits[ec] += 1
        p = win0 + s["arel"]
        assert bytes(out[pos0:s["pos"]]) == expect[pos0:s["pos"]], "bytes differ behind position %d" % pos0
        if ec == 3:                         # a copy the loop leaves to its caller (wave_copy)
            n, dist = s["lenout"], s["rep0"] + 1
            for i in range(n):
                out[s["pos"] + i] = out[s["pos"] + i - dist] if s["pos"] + i >= dist + base else 0
            s["pos"] += n
            s["wpos"] = (s["wpos"] + n) % dict_size
            s["prev"] = out[s["pos"] - 1]
            s["mb"] = out[s["pos"] - dist] if s["pos"] >= dist + base else 0
        elif ec != 0:
            break
    return bytes(out[base:s["pos"]]) if base else bytes(out[:s["pos"]]), m, entries, exits, p


def _reference_state_at(payload, lc, lp, pb, dict_size, size, stop):
    """the Python restatement, stopped in front of the packet that starts at output position `stop`"""
    st = lzma_pydec._State(lc, lp, pb)
    w = lzma_pydec.Window(dict_size)
    rc = lzma_pydec._Rc(payload, 0, len(payload))
    assert rc.init()
    res, _ = lzma_pydec._run(rc, st, w, size, stop_at=stop)
    assert res == "stop" and len(w.total) == stop
    return rc, st


@pytest.mark.parametrize("family,n,lc,lp,pb,ds", [("T", 6000, 3, 0, 2, 1 << 16), ("R", 1500, 3, 0, 2, 1 << 16),
                                                  ("M", 5000, 0, 2, 0, 4096), ("Z", 9000, 1, 1, 1, 4096),
                                                  ("T", 5000, 4, 0, 4, 5000)])
def test_committed_loop_decodes_real_streams_on_the_emulator(program, family, n, lc, lp, pb, ds):
    p = corpus.plain(family, 4242 + n, n)
    blob = corpus.compress_alone(p, dict_size=ds, lc=lc, lp=lp, pb=pb, known_size=True, preset=6 if family == "T" else 0)
    assert oracle.lzma1_alone(blob, n)[0] == p
    ds = max(4096, int.from_bytes(blob[1:5], "little"))     # (liblzma rounds the dictionary size in the header up)
    payload = blob[13:]
    out, m, entries, exits, in_pos = run_fast_loop(program, payload, lc, lp, pb, ds, n, p)
    assert out == p[:len(out)]
    # the loop ran until the checked path's margins: nearly everything was decoded by it, over many entries
    assert len(out) > n - 336 - 300 or len(payload) - in_pos < 64, (len(out), n, in_pos, len(payload))
    assert exits[1] == 0 and exits[2] == 0 and entries >= 1
    rc, st = _reference_state_at(payload, lc, lp, pb, ds, n, len(out))
    s = m.s
    assert (s["range"], s["code"], s["state"]) == (rc.range, rc.code, st.state)
    assert [s["rep0"], s["rep1"], s["rep2"], s["rep3"]] == st.reps
    assert in_pos == rc.p and s["prev"] == p[len(out) - 1]


@pytest.mark.parametrize("family,n,lc,lp,pb,ds,branchy", [
    ("T", 5000, 3, 0, 2, 1 << 16, False), ("R", 1500, 3, 0, 2, 1 << 16, False), ("M", 4000, 0, 2, 0, 4096, False),
    ("Z", 9000, 1, 1, 1, 4096, False), ("M", 3000, 2, 2, 1, 1 << 16, False),
    ("T", 5000, 3, 0, 2, 1 << 16, True), ("R", 1500, 3, 0, 2, 1 << 16, True), ("Z", 9000, 1, 1, 1, 4096, True), ("M", 3000, 2, 2, 1, 5000, True)])
def test_compact_loop_decodes_real_streams_on_the_emulator(family, n, lc, lp, pb, ds, branchy):
    """round 5: the COMMITTED loops over the compact model layout (xlz_fastpath_pb2.inc: room for 4 posStates in every
    table a posState indexes; what launches use when every unit's pb is <= 2 -- and xlz_fastpath_pb2_br.inc, the same with
    branchy decisions, what launches of 24 workgroups per CU run) on the same streams as the full loop -- every byte, and
    range / code / state / reps / input position where it hands back."""
    program = _render(("compact", "dbr", "dbrs") if branchy else ("compact",))
    assert program.lay["P_LIT"] == 924
    p = corpus.plain(family, 4242 + n, n)
    blob = corpus.compress_alone(p, dict_size=ds, lc=lc, lp=lp, pb=pb, known_size=True, preset=6 if family == "T" else 0)
    ds = max(4096, int.from_bytes(blob[1:5], "little"))
    payload = blob[13:]
    out, m, entries, exits, in_pos = run_fast_loop(program, payload, lc, lp, pb, ds, n, p, strict_waits=(family == "M"))
    assert out == p[:len(out)]
    assert len(out) > n - 336 - 300 or len(payload) - in_pos < 64, (len(out), n, in_pos, len(payload))
    assert exits[1] == 0 and exits[2] == 0 and entries >= 1
    rc, st = _reference_state_at(payload, lc, lp, pb, ds, n, len(out))
    s = m.s
    assert (s["range"], s["code"], s["state"]) == (rc.range, rc.code, st.state)
    assert [s["rep0"], s["rep1"], s["rep2"], s["rep3"]] == st.reps
    assert in_pos == rc.p and s["prev"] == p[len(out) - 1]


NEXT = ("slot0", "vprev", "rmov", "nopos", "l7blk", "warel", "vreps")   # round 2's prepared variants: the default since round 3


@pytest.mark.parametrize("add,remove", [((), ("lgather",) + NEXT), ((), ("hdpp",) + NEXT), ((), ("flim", "cflag") + NEXT),
                                        ((), ("rlhoist", "vperm", "tuc") + NEXT), (("lit8g",), NEXT), (("order3", "pwhoist"), NEXT),
                                        (("order1",), ("bralign",) + NEXT), (("litrun",), ("flim",) + NEXT),
                                        (("slot0", "lit8g"), ("cflag",) + NEXT), ((), ("hdpp",)), ((), ("rlhoist", "vperm", "tuc")),
                                        ((), ("slot0",)), ((), ("vprev",)), ((), ("rmov",)), ((), ("nopos",)), ((), ("l7blk",)),
                                        ((), ("warel",)), ((), ("vreps",)), ((), NEXT), (("dbr",), ()), (("dbr",), ("l7blk", "slot0")), (("dbr", "dbrw"), ()),
                                        (("dbrw",), ("slot0",)), ((), ("wsb",)), (("dbr",), ("rlhoist",)), (("scode",), ()),
                                        (("scode",), ("rlhoist", "wsb")), (("hsb",), ()), (("hsb", "scode"), ()), ((), ("lwait",)), (("dbr", "dbrs"), ()), (("pref",), ("db6",)),
                                        ((), ("db6",)), ((), ("tu8",)), ((), ("cchk",)), ((), ("lctx",)), ((), ("hiss",)),
                                        ((), ("db6", "tu8", "cchk", "lctx", "hiss", "g8", "hd2", "rmov2")), ((), ("tuc",)), ((), ("vperm",)),
                                        ((), ("g8",)), ((), ("hd2",)), ((), ("rmov2",)), (("rot",), ()), ((), ("ml4",)), ((), ("pkm",)), ((), ("ml4", "pkm", "g8")), (("mlv",), ()), (("vcur",), ("hd2",)),
                                        (("hoist0",), ("hd2",)), (("rot", "vcur", "hoist0"), ("hd2", "rmov2"))])
def test_generator_switches_still_decode(add, remove):
    """the code paths kept in the generator as measured alternatives (profiles/r02/layout_scan.md)
    are not dead code: each of them decodes a stream correctly on the emulator"""
    lc, lp, pb, ds, n = 3, 0, 2, 1 << 16, 1800
    p = corpus.plain("T", 99, n)
    blob = corpus.compress_alone(p, dict_size=ds, lc=lc, lp=lp, pb=pb, known_size=True, preset=6)
    out, m, entries, exits, in_pos = run_fast_loop(_render(add, remove), blob[13:], lc, lp, pb, ds, n, p, dpp="hdpp" not in remove)
    assert out == p[:len(out)] and len(out) > n - 700 and exits[1] == 0
    rc, st = _reference_state_at(blob[13:], lc, lp, pb, ds, n, len(out))
    assert (m.s["range"], m.s["code"], m.s["state"], m.s["rep0"]) == (rc.range, rc.code, st.state, st.reps[0])


@pytest.mark.parametrize("without", [(), NEXT])
def test_end_marker_and_error_exits(program, without):
    """exit 2 (distance 0xFFFFFFFF: the end marker, decompress.go:633-645) on a stream of unknown size, and exit 1
    (a distance the window does not hold, :651-653) on corrupted streams: same position and bytes as the oracle"""
    if without:
        program = _render((), without)   # round 2's loop
    lc, lp, pb, ds, n = 3, 0, 2, 1 << 16, 3000
    p = corpus.plain("T", 5, n)
    blob = corpus.compress_alone(p, dict_size=ds, known_size=False, preset=6)       # ends with the marker
    assert blob[5:13] == b"\xff" * 8
    # (64 bytes behind the stream: with fewer than 32 left the kernel hands over to the checked path)
    out, m, entries, exits, in_pos = run_fast_loop(program, blob[13:] + b"\0" * 64, lc, lp, pb, ds, n + 4096, p + b"\0" * 4096)
    assert in_pos == len(blob) - 13
    assert out == p and exits[2] == 1 and m.s["code"] == 0 and m.s["rep0"] == 0xFFFFFFFF
    caught = 0
    for k in range(40):
        bad = bytearray(blob)
        bad[13 + 200 + 37 * k] ^= 1 << (k % 8)
        bad = bytes(bad)
        want, status, _ = oracle.lzma1_alone(bad, n + 4096)
        if status != oracle.ERR_RESULT or len(want) > n - 400:
            continue
        out, m, entries, exits, in_pos = run_fast_loop(program, bad[13:] + b"\0" * 64, lc, lp, pb, ds, n + 4096, want + b"\0" * 8192)
        assert exits[1] + exits[2] == 1 and out == want, k
        caught += 1
        if caught == 4:
            break
    assert caught >= 2


def test_every_load_is_waited_for_and_may_complete_as_late_as_its_wait(program):
    """the emulator's strict mode: the result of an LDS read or a global load is not in its register before the
    s_waitcnt that covers it (lgkmcnt / vmcnt counted as the hardware counts them, LDS in order), reading or
    overwriting it earlier is an error, and global loads read memory AT the wait -- the latest the hardware could.
    The committed loop decodes the same bytes as with loads that complete at issue: its s_waitcnt counts (the
    lgkmcnt(2) behind the literal walk's gather among them) are sufficient, and the deferred match copy does not
    depend on when its load lands."""
    for fam, n, lc, lp, pb, ds in (("T", 4000, 3, 0, 2, 1 << 16), ("M", 3000, 0, 2, 0, 4096)):
        p = corpus.plain(fam, 777 + n, n)
        blob = corpus.compress_alone(p, dict_size=ds, lc=lc, lp=lp, pb=pb, known_size=True, preset=6)
        out, m, entries, exits, in_pos = run_fast_loop(program, blob[13:], lc, lp, pb, ds, n, p, strict_waits=True)
        assert out == p[:len(out)] and len(out) > n - 700
        assert not m.pending and not m.vm                      # every exit has waited for what it started


def test_dictionary_epoch_that_starts_inside_the_output(program):
    """an LZMA2 unit behind a dictionary reset: positions are absolute in the output, the window starts at wbase
    (distance checks, the `window full` test and the copies' source test all use pos - wbase); 70 000 bytes in
    front make the absolute positions differ from the window's in more than the low bits"""
    lc, lp, pb, ds, n, base = 3, 0, 2, 4096, 7000, 70_000
    p = corpus.plain("M", 31, n)
    blob = corpus.compress_alone(p, dict_size=ds, known_size=True, preset=0)
    junk = corpus.plain("R", 32, base)
    out, m, entries, exits, in_pos = run_fast_loop(program, blob[13:], lc, lp, pb, ds, base + n, junk + p, base=base)
    assert out == p[:len(out)] and len(out) > n - 700 and exits[1] == 0
    rc, st = _reference_state_at(blob[13:], lc, lp, pb, ds, n, len(out))
    assert (m.s["range"], m.s["code"], m.s["state"], m.s["rep0"]) == (rc.range, rc.code, st.state, st.reps[0])


@pytest.mark.parametrize("without", [(), NEXT])
def test_prepared_variants_on_a_window_that_wraps(program, without):
    """slot0 + vprev + rmov + nopos + l7blk + warel + vreps (the committed loop since round 3; `without`: round 2's loop) on the two
    configurations the switch test does not reach: a 4 KiB dictionary that wraps several times (nopos keeps the
    window's fill as max(window.pos, 0 or dictSize)) and an epoch that starts 70 000 bytes into the output, with
    late-landing loads"""
    prog = _render((), without) if without else program
    for fam, n, lc, lp, pb, ds, base in (("Z", 9000, 1, 1, 1, 4096, 0), ("M", 7000, 3, 0, 2, 4096, 70_000)):
        p = corpus.plain(fam, 4242 + n, n)
        blob = corpus.compress_alone(p, dict_size=ds, lc=lc, lp=lp, pb=pb, known_size=True, preset=0)
        junk = corpus.plain("R", 32, base) if base else b""
        out, m, entries, exits, in_pos = run_fast_loop(prog, blob[13:], lc, lp, pb, ds, base + n, junk + p, strict_waits=True, base=base)
        assert out == p[:len(out)] and exits[1] == 0 and entries > 5
        rc, st = _reference_state_at(blob[13:], lc, lp, pb, ds, n, len(out))
        assert (m.s["range"], m.s["code"], m.s["state"], m.s["prev"]) == (rc.range, rc.code, st.state, p[len(out) - 1])
        assert [m.s["rep0"], m.s["rep1"], m.s["rep2"], m.s["rep3"]] == st.reps
