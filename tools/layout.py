"""Code-layout pass for the generated fast path (tools/gen_fastpath.py).

Measured on MI355X (profiles/r02/layout_scan.md): a conditional branch that sits in the upper half of a
16-byte block (address mod 16 = 8 or 12) costs the wave a few cycles even when it is not taken -- the
literal loop with all of its per-level normalisation branches in the lower halves decodes 3.4 % faster
than the same code moved by 8 bytes.  This pass moves every conditional branch into a lower half
WITHOUT adding instructions where it can: a 4-byte VALU instruction in front of the branch is
re-encoded in its 8-byte VOP3 form (`_e64`, same operation); only where no such instruction is
available does it insert `s_nop 0`.

Instruction sizes come from the assembler itself (llvm-mc --show-encoding on the instruction stream with
the asm operands replaced by registers)."""
import os
import re
import subprocess
import tempfile

LLVM_MC = os.environ.get("LLVM_MC", "/opt/rocm/lib/llvm/bin/llvm-mc")
SGPR_OPS = ["range", "code", "cur", "arel", "state", "rep0", "rep1", "rep2", "rep3", "pos", "wpos", "prev", "mb",
            "exitc", "lenout", "arel_lim", "pos_lim", "dict", "dictm1", "pos_mask", "lc", "lc8", "wbase"]
SGPR64_OPS = ["outp", "mptr"]
VGPR_OPS = ["vin", "vlane", "vhc", "vhms", "vhm2", "vlitnext", "vlpm", "vpm"]


def _subst(line):
    def rep(m):
        n = m.group(1)
        if n in SGPR_OPS:
            return "s%d" % (10 + SGPR_OPS.index(n))
        if n in SGPR64_OPS:
            k = 40 + 2 * SGPR64_OPS.index(n)
            return "s[%d:%d]" % (k, k + 1)
        return "v%d" % (1 + VGPR_OPS.index(n))
    return re.sub(r"%\[(\w+)\]", rep, line).replace("%=", "0")


def sizes(lines):
    """byte size of every line (labels and directives: 0), None where the assembler rejects a line"""
    with tempfile.NamedTemporaryFile("w", suffix=".s", delete=False) as f:
        for l in lines:
            f.write(_subst(l) + "\n")
        path = f.name
    try:
        r = subprocess.run([LLVM_MC, "-triple=amdgcn-amd-amdhsa", "-mcpu=gfx950", "-show-encoding", path],
                           capture_output=True, text=True)
    finally:
        os.unlink(path)
    if r.returncode != 0:
        raise RuntimeError("llvm-mc failed:\n" + r.stderr[-2000:])
    enc = [len(m.group(1).split(",")) for m in re.finditer(r"encoding: \[([^\]]*)\]", r.stdout)]
    out, k = [], 0
    for l in lines:
        if l.endswith(":") or l.startswith("."):
            out.append(0)
        else:
            out.append(enc[k])
            k += 1
    assert k == len(enc), (k, len(enc))
    return out


def _promotable(line):
    """a 4-byte VALU instruction that also exists in VOP3 form with the same operands"""
    m = line.split()[0]
    if not m.startswith("v_") or m.endswith("_e64") or "_dpp" in m or "sdwa" in m:
        return False
    if m in ("v_readlane_b32", "v_writelane_b32", "v_readfirstlane_b32", "v_nop"):
        return False
    return True


def _promote(line):
    parts = line.split(None, 1)
    return parts[0] + "_e64 " + parts[1]


def align_branches(lines, good=(0, 4), modulo=16, start_directive=".p2align 4", targets=()):
    """-> (new lines, promotions, executed nops, never-executed padding nops).  The stream is laid out from a `start_directive` boundary that the
    caller places where execution never falls through (behind an unconditional branch)."""
    lines = list(lines)
    sz = sizes(lines)
    # which promotions does the assembler accept?  (one batch call)
    cand = [i for i, l in enumerate(lines) if sz[i] == 4 and _promotable(l)]
    ok = set()
    if cand:
        trial = [_promote(lines[i]) for i in cand]
        with tempfile.NamedTemporaryFile("w", suffix=".s", delete=False) as f:
            for t in trial:
                f.write(_subst(t) + "\n")
            path = f.name
        r = subprocess.run([LLVM_MC, "-triple=amdgcn-amd-amdhsa", "-mcpu=gfx950", "-show-encoding", path],
                           capture_output=True, text=True)
        os.unlink(path)
        bad_lines = set(int(m.group(1)) for m in re.finditer(r":(\d+):\d+: error", r.stderr))
        encs = {}
        # with errors present llvm-mc still prints the good ones, in order; map them back by line number
        good_idx = [k for k in range(len(trial)) if (k + 1) not in bad_lines]
        found = [len(m.group(1).split(",")) for m in re.finditer(r"encoding: \[([^\]]*)\]", r.stdout)]
        if len(found) == len(good_idx):
            for k, n in zip(good_idx, found):
                if n == 8:
                    ok.add(cand[k])
    start = [i for i, l in enumerate(lines) if l.startswith(".p2align")]
    start = start[0] if start else 0
    out, off, promos, nops, dead_nops = [], 0, 0, 0, 0
    seg = []  # indices (into out) of promotable 4-byte instructions since the last placed branch
    dead = None  # index (into out) behind an unconditional branch since then: padding there is never executed
    for i, l in enumerate(lines):
        if i == start:
            out.append(l)
            off = 0
            seg = []
            continue
        if i < start:
            out.append(l)
            continue
        if l.endswith(":"):
            # targets: ((regex on the label, alignment in bytes), ...) -- only labels that execution reaches by a
            # branch alone (padding goes behind the unconditional branch in front of them)
            if dead is not None and all(x.endswith(":") for x in out[dead:]):
                for pat, al in targets:
                    if re.match(pat, l):
                        k = (-off) % al // 4
                        out[dead:dead] = ["s_nop 0"] * k
                        dead_nops += k
                        off += 4 * k
                        break
            out.append(l)
            continue
        if l.startswith("."):
            raise RuntimeError("a second alignment directive inside the laid-out stream: " + l)
        if l.split()[0].startswith("s_cbranch"):
            need = 0
            while (off + 4 * need) % modulo not in good:
                need += 1
            while need and seg:
                j = seg.pop()
                out[j] = _promote(out[j])
                promos += 1
                need -= 1
                off += 4
            if need and dead is not None:
                out[dead:dead] = ["s_nop 0"] * need
                dead_nops += need
                off += 4 * need
                need = 0
            for _ in range(need):
                out.append("s_nop 0")
                nops += 1
                off += 4
            out.append(l)
            off += sz[i]
            seg = []
            dead = None
            continue
        out.append(l)
        if i in ok:
            seg.append(len(out) - 1)
        off += sz[i]
        if l.split()[0] in ("s_branch", "s_setpc_b64"):
            seg = []
            dead = len(out)
    return out, promos, nops, dead_nops


def report(lines, modulo=16, good=(0, 4)):
    """(conditional branches, those outside the good residues) behind the alignment directive"""
    sz = sizes(lines)
    off, n, bad, started = 0, 0, 0, False
    for l, s in zip(lines, sz):
        if l.startswith(".p2align"):
            off, started = 0, True
            continue
        if started and l.split()[0].startswith("s_cbranch"):
            n += 1
            bad += off % modulo not in good
        off += s
    return n, bad
