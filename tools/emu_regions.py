"""Where the fast loop's instructions are executed (no GPU: tools/gcn_emu.py): executed instructions per REGION of the
generated code -- a region starts at a named label (match, rep, copy, pkt, mlit, dist, direct, ...); normalisation
stubs, window-wrap stubs and copy-completion blocks are regions of their own -- with the number of times each region
was entered, by class (scalar / vector / lane / branch / LDS+memory / wait).
usage: python tools/emu_regions.py [family] [bytes] [--variant a,b] [--without a,b] [--fast]
(--fast: the bench corpora's encoder setting instead of preset 6)"""
import os
import re
import sys
from collections import Counter, defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import corpus
import gcn_emu
import test_fastpath_emulated as T

STUB = re.compile(r"^\.L(n\d+b?|w\d+[bn]?|f\d+b?|d\d+j?|db\d+)_%=$")


def klass(mn):
    if mn in ("v_readlane_b32", "v_writelane_b32", "v_readfirstlane_b32"):
        return "lane"
    if mn in ("s_nop", "s_waitcnt"):
        return "wait"
    if mn.startswith("s_cbranch") or mn in ("s_branch", "s_setpc_b64"):
        return "branch"
    return "salu" if mn.startswith("s_") else "valu" if mn.startswith("v_") else "mem"


def main():
    args = sys.argv[1:]
    add = args[args.index("--variant") + 1].split(",") if "--variant" in args else []
    rem = args[args.index("--without") + 1].split(",") if "--without" in args else []
    pos = [a for i, a in enumerate(args) if not a.startswith("--") and (i == 0 or args[i - 1] not in ("--variant", "--without"))]
    fam = pos[0] if pos else "T"
    n = int(pos[1]) if len(pos) > 1 else 8000
    prog = T._render(add, rem)
    region_of = {}
    stats = defaultdict(Counter)
    entered = Counter()

    def run(self, start_label=None, max_steps=50_000_000):
        if not region_of:
            at = defaultdict(list)
            for name, pc in self.labels.items():
                at[pc].append(name)
            cur, named = "entry", "entry"
            for pc in range(len(self.ins) + 1):
                for name in at.get(pc, []):
                    m = STUB.match(name)
                    if m:
                        k = m.group(1)
                        if k[0] == "n":
                            cur = named if k.endswith("b") else "norm stub"
                        elif k[0] == "w":
                            cur = named if k.endswith("b") else "wrap stub"
                        elif k[0] == "f":
                            cur = named if k.endswith("b") else "copy finish"
                        elif k.startswith("db"):
                            cur = named = "direct bits"
                        else:
                            cur = named
                    else:
                        cur = named = name[2:-3]
                region_of[pc] = cur
        pc, steps, ni = 0, 0, len(self.ins)
        last = None
        while pc < ni:
            fn, ops, mods, text = self.ins[pc]
            self.pc = pc
            r = region_of[pc]
            if r != last:
                entered[r] += 1
                last = r
            stats[r][klass(text.split()[0])] += 1
            nxt = fn(ops, mods)
            pc = pc + 1 if nxt is None else nxt
            steps += 1
        self.n_exec += steps
        return steps
    gcn_emu.Machine.run = run
    p = corpus.plain(fam, 4242 + n, n)
    enc = {"mode": 1, "mf": 3, "nice_len": 32, "depth": 2} if "--fast" in args else (6 if fam == "T" else 0)
    blob = corpus.compress_alone(p, dict_size=1 << 16, known_size=True, preset=enc)
    out, m, entries, exits, ip = T.run_fast_loop(prog, blob[13:], 3, 0, 2, 1 << 16, n, p, dpp="hdpp" not in rem)
    nb = len(out)
    total = sum(sum(c.values()) for c in stats.values())
    print("family %s, %d bytes (ratio %.3f) decoded by the loop in %d entries: %d instructions = %.2f per byte"
          % (fam, nb, len(blob) / n, entries, total, total / nb))
    print("%-14s %8s %8s %7s | %6s %6s %6s %6s %6s %6s" % ("region", "entered", "instr", "/byte", "salu", "valu", "lane", "branch", "mem", "wait"))
    for r, c in sorted(stats.items(), key=lambda x: -sum(x[1].values())):
        t = sum(c.values())
        print("%-14s %8d %8d %7.2f | %6.1f %6.1f %6.1f %6.1f %6.1f %6.1f   (per entry: %.1f)"
              % (r, entered[r], t, t / nb, *(c[k] / entered[r] for k in ("salu", "valu", "lane", "branch", "mem", "wait")), t / entered[r]))
    tot = Counter()
    for c in stats.values():
        tot.update(c)
    print("%-14s %8s %8d %7.2f | %s" % ("all", "", total, total / nb, "  ".join("%s %.2f/B" % (k, v / nb) for k, v in tot.most_common())))


if __name__ == "__main__":
    main()
