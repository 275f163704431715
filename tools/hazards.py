"""gfx950 (gfx940-class) data hazards the assembler does NOT resolve inside inline asm.

Rules (LLVM GCNHazardRecognizer::checkVALUHazards, hasVDecCoExecHazard; confirmed by what hipcc
emits for gfx950 and, for R1, by a decoder that went wrong without it):
  R1  VALU writes VGPR            -> v_readlane / v_readfirstlane reads it:        1 wait state
  R2  VALU writes SGPR            -> VALU reads it as an operand:                  2
  R3  VALU writes VCC             -> VALU reads VCC (v_cndmask, v_addc, ...):      2
  R4  VALU writes SGPR / VCC      -> v_readlane / v_writelane lane select:         4
  R6  VALU writes SGPR            -> VMEM reads it:                                5
  R7  VALU writes VGPR            -> DPP instruction reads it as src0:             2
A wait state is any instruction in between; s_nop N counts N + 1.

check(lines) walks the straight-line order and every branch edge (a taken branch counts as one
wait state) and returns the violations; fix(lines) inserts s_nop in front of the consumer.
"""
import re

SGPR_OPS = {"range", "code", "cur", "arel", "state", "rep0", "rep1", "rep2", "rep3", "pos", "wpos", "prev", "mb",
            "exitc", "lenout", "arel_lim", "pos_lim", "dict", "dictm1", "pos_mask", "lc", "lc8", "wbase", "outp", "mptr"}
VGPR_OPS = {"vin", "vlane", "vhc", "vhms", "vhm2", "vlitnext", "vlpm", "vpm"}
WINDOW = 6


def reg_of(tok):
    tok = tok.strip()
    m = re.match(r"%+\[(\w+)\]", tok)
    if m:
        n = m.group(1)
        return ("s:" + n) if n in SGPR_OPS else ("v:" + n) if n in VGPR_OPS else None
    if re.match(r"v\d+$", tok):
        return "v:" + tok
    if re.match(r"s\d+$", tok):
        return "s:" + tok
    if tok in ("vcc", "vcc_lo", "vcc_hi"):
        return "vcc"
    m = re.match(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return "s:s" + m.group(1)
    return None


class Ins:
    def __init__(self, text):
        self.text = text
        parts = text.split(None, 1)
        self.m = parts[0]
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        ops = [o.split()[0] if o and not o.startswith("s[") else o for o in ops]  # drop "offset:..."
        self.ops = ops
        self.valu = self.m.startswith("v_")
        self.vmem = self.m.startswith("global_") or self.m.startswith("buffer_") or self.m.startswith("flat_")
        self.lane = self.m in ("v_readlane_b32", "v_writelane_b32")
        self.rdlane = self.m in ("v_readlane_b32", "v_readfirstlane_b32")
        self.dpp = self.m.endswith("_dpp")
        self.branch = self.m.startswith("s_cbranch") or self.m in ("s_branch", "s_setpc_b64")
        self.ws = 1
        if self.m.startswith("."):  # an assembler directive (.p2align): no instruction, no wait state
            self.ws = 0
        if self.m == "s_nop":
            self.ws = int(ops[0]) + 1
        regs = [reg_of(o) for o in ops]
        self.writes, self.reads, self.lanesel = [], [], None
        if self.valu:
            if self.m.startswith("v_cmp"):
                self.writes = [regs[0]]
                self.reads = regs[1:]
            elif re.match(r"v_(add|sub|subrev|addc|subb|subbrev)_co_", self.m):
                self.writes = [regs[0], regs[1]]
                self.reads = regs[2:]
            else:
                self.writes = [regs[0]]
                self.reads = regs[1:]
                if self.m == "v_writelane_b32":
                    self.reads = regs[1:2]  # the partially written vdst is not a hazard source here
            if self.lane:
                self.lanesel = regs[2] if len(regs) > 2 else None
                self.reads = regs[1:2]
        elif self.vmem or self.m.startswith("ds_"):
            self.reads = regs  # conservative: every named register
        self.writes = [r for r in self.writes if r]
        self.reads = [r for r in self.reads if r]


def need(prod, cons, reg):
    """wait states required between VALU `prod` writing reg and `cons`"""
    n = 0
    if reg.startswith("v:"):
        if cons.rdlane and reg in cons.reads:
            n = 1
        if cons.dpp and cons.reads and cons.reads[0] == reg:
            n = 2
        return n
    # SGPR or VCC written by a VALU instruction
    if cons.lane and cons.lanesel == reg:
        n = max(n, 4)
    if cons.valu and reg in cons.reads:
        n = max(n, 2)
    if cons.vmem and reg in cons.reads:
        n = max(n, 5)
    return n


def violations_in(seq):
    """seq: list of (index, Ins); returns [(consumer_index, missing, producer_text)]"""
    out = []
    for a in range(len(seq)):
        ia, pa = seq[a]
        if not pa.valu:
            continue
        for reg in pa.writes:
            ws = 0
            for b in range(a + 1, min(len(seq), a + 1 + WINDOW)):
                ib, pb = seq[b]
                n = need(pa, pb, reg)
                if n > ws:
                    out.append((ib, n - ws, pa.text, reg))
                if reg in pb.writes and pb.m != "v_writelane_b32":
                    break
                ws += pb.ws
                if ws >= 5:
                    break
    return out


def analyse(lines):
    ins = [(i, Ins(l)) for i, l in enumerate(lines) if not l.endswith(":")]
    labels = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(":")}
    viol = violations_in(ins)
    pos_of = {i: k for k, (i, _) in enumerate(ins)}
    for k, (i, p) in enumerate(ins):
        if p.branch and p.m != "s_setpc_b64":
            tgt = p.ops[-1]
            if tgt not in labels:
                continue
            t = labels[tgt]
            nxt = [(j, q) for (j, q) in ins if j > t][:WINDOW]
            seq = ins[max(0, k - WINDOW):k + 1] + nxt
            for v in violations_in(seq):
                if v[0] > t and v[0] in [j for j, _ in nxt]:
                    # only edges that cross the branch
                    prod_idx = [j for j, q in seq if q.text == v[2]]
                    if prod_idx and prod_idx[0] <= i:
                        viol.append(v)
    best = {}
    for (idx, missing, ptxt, reg) in viol:
        if idx not in best or best[idx][0] < missing:
            best[idx] = (missing, ptxt, reg)
    return best


def fix(lines, verbose=False):
    """insert s_nop in front of every consumer that is too close to its producer"""
    total = 0
    while True:
        best = analyse(lines)
        if not best:
            return lines, total
        out = []
        for i, l in enumerate(lines):
            if i in best:
                if verbose:
                    print("hazard: +%d before `%s` (after `%s`, %s)" % (best[i][0], l, best[i][1], best[i][2]))
                out.append("s_nop %d" % (best[i][0] - 1))
                total += 1
            out.append(l)
        lines = out
