"""Dynamic instruction mix of the generated fast loop on a real stream, from the emulator (no GPU): executed
instructions per decoded byte by class and by mnemonic, for the committed loop or any generator variant.
usage: python tools/emu_mix.py [family T|R|M|Z] [bytes] [--variant a,b] [--without a,b]
(round 2, committed loop: R 1500 -> 150.2 per byte: 70.4 SALU, 43.2 VALU, 14.3 branch, 11.3 lane; the hardware
counters of cfg2-R say 149.07.)"""
import os
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import corpus
import gcn_emu
import test_fastpath_emulated as T


def main():
    args = [a for a in sys.argv[1:]]
    add = args[args.index("--variant") + 1].split(",") if "--variant" in args else []
    rem = args[args.index("--without") + 1].split(",") if "--without" in args else []
    pos = [a for i, a in enumerate(args) if not a.startswith("--") and (i == 0 or args[i - 1] not in ("--variant", "--without"))]
    fam = pos[0] if pos else "T"
    n = int(pos[1]) if len(pos) > 1 else 4000
    prog = T._render(add, rem)
    cnt = Counter()

    def run(self, start_label=None, max_steps=50_000_000):
        pc, steps, ni = 0, 0, len(self.ins)
        while pc < ni:
            fn, ops, mods, text = self.ins[pc]
            self.pc = pc
            cnt[text.split()[0]] += 1
            nxt = fn(ops, mods)
            pc = pc + 1 if nxt is None else nxt
            steps += 1
        self.n_exec += steps
        return steps
    gcn_emu.Machine.run = run
    p = corpus.plain(fam, 4242 + n, n)
    blob = corpus.compress_alone(p, dict_size=1 << 16, known_size=True, preset=6 if fam == "T" else 0)
    out, m, entries, exits, ip = T.run_fast_loop(prog, blob[13:], 3, 0, 2, 1 << 16, n, p, dpp="hdpp" not in rem)
    nb = len(out)

    def klass(mn):
        if mn in ("v_readlane_b32", "v_writelane_b32", "v_readfirstlane_b32"):
            return "lane"
        if mn in ("s_nop", "s_waitcnt"):
            return "nop / wait"
        if mn.startswith("s_cbranch") or mn in ("s_branch", "s_setpc_b64"):
            return "branch"
        return "salu" if mn.startswith("s_") else "valu" if mn.startswith("v_") else "lds / memory"
    cat = Counter()
    for mn, c in cnt.items():
        cat[klass(mn)] += c
    print("family %s: %d bytes decoded by the loop in %d entries, %d instructions = %.1f per byte"
          % (fam, nb, entries, sum(cnt.values()), sum(cnt.values()) / nb))
    for k, v in cat.most_common():
        print("  %-13s %6.1f per byte" % (k, v / nb))
    print("  most frequent:", ", ".join("%s %.2f" % (k, v / nb) for k, v in cnt.most_common(16)))


if __name__ == "__main__":
    main()
