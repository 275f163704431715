"""Dev tool (GPU box): time several builds of libxlz.so on the same corpora in ONE process.
Every library is loaded with its own ctypes handle; a corpus is generated once and decoded by each.
    python tools/ab_bench.py name=path/to/lib.so [name=...] [--fams T,R,S] [--steps 3]
Workloads (all 4096 or more streams so that the launch fills the chip; M = mixed segments, Z = long repeats): T = cfg2-T shape (1 MiB text, preset 6,
1024 distinct x 4), R = cfg2-R shape (512 distinct x 8), S = cfg3 shape (64 KiB text, 8192 distinct x 8).
Prints decompressed GiB/s from HIP events on the kernel's stream; every variant's output is checked (SHA-256)."""
import ctypes, hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import corpus
from lzma_amd import _native as N

WORK = {"T": ("T", 1024, 4, 1 << 20, 6), "R": ("R", 512, 8, 1 << 20, 0), "S": ("T", 8192, 8, 65536, 0),
        "M": ("M", 1024, 4, 1 << 20, 0), "Z": ("Z", 256, 16, 1 << 20, 0),
        # the cfg3 shape at the sizes of an N-GPU shard (bench.py: scaling_projection): 32 768 / 16 384 / 8 192 streams
        "S2": ("T", 8192, 4, 65536, 0), "S4": ("T", 8192, 2, 65536, 0), "S8": ("T", 8192, 1, 65536, 0),
        # the cfg3 shape behind lc = 2: the model is 5240 bytes = five LDS granules, 25 fit a CU -- what 24 waves per CU
        # would buy, measured without touching the layout (round 5)
        "L2": ("T", 8192, 8, 65536, 0, {"lc": 2})}


def load(path):
    L = ctypes.CDLL(os.path.abspath(path))
    vp, sz, i32 = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    L.xlz_ctx_create.argtypes = [i32, ctypes.POINTER(vp)]
    L.xlz_ctx_destroy.argtypes = [vp]
    L.xlz_ctx_event_record.argtypes = [vp, i32]
    L.xlz_ctx_event_elapsed_ms.argtypes = [vp, i32, i32, ctypes.POINTER(ctypes.c_float)]
    L.xlz_batch_create.argtypes = [vp, ctypes.POINTER(N.StreamDesc), sz, ctypes.POINTER(vp)]
    L.xlz_batch_run.argtypes = [vp]
    L.xlz_batch_sync.argtypes = [vp]
    L.xlz_batch_results.argtypes = [vp, ctypes.POINTER(N.Result)]
    L.xlz_batch_download.argtypes = [vp, sz, vp, sz]
    L.xlz_batch_destroy.argtypes = [vp]
    return L


CORPUS_FILE = "/dev/shm/xlz_ab_corpus.pkl"


def main():
    import pickle
    import subprocess
    libs, fams, steps, child = [], ["T", "R", "S"], 3, False
    args = sys.argv[1:]
    while args:
        a = args.pop(0)
        if a == "--fams":
            fams = args.pop(0).split(",")
        elif a == "--steps":
            steps = int(args.pop(0))
        elif a == "--child":
            child = True
        else:
            name, path = a.split("=", 1)
            libs.append((name, path))
    if not child:
        # parent: corpora once, then every library in a child process of its own under a timeout: a
        # kernel variant that never finishes costs two minutes, not the whole call
        corp = {}
        for f in fams:
            fam, nd, rep, size, preset = WORK[f][:5]
            extra = WORK[f][5] if len(WORK[f]) > 5 else {}
            t0 = time.time()
            cs, hs = corpus.make_alone_batch(fam, nd, size, base_seed=77, workers=min(os.cpu_count() or 1, 64), preset=preset, **extra)
            corp[f] = (cs, hs, rep, size)
            print("corpus %s: %d distinct x%d of %d B, ratio %.3f, %.1f s" % (f, nd, rep, size, sum(map(len, cs)) / (nd * size),
                                                                          time.time() - t0), flush=True)
        pickle.dump(corp, open(CORPUS_FILE, "wb"))
        res = {}
        for name, path in libs:
            try:
                out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--fams", ",".join(fams), "--steps",
                                      str(steps), "%s=%s" % (name, path)], capture_output=True, text=True, timeout=150).stdout
            except subprocess.TimeoutExpired as e:
                print("%-14s *** TIMEOUT (hung kernel?) *** %s" % (name, (e.stdout or b"").decode()[-300:]), flush=True)
                continue
            sys.stdout.write(out)
            sys.stdout.flush()
            for l in out.splitlines():
                w = l.split()
                if len(w) >= 4 and w[3] == "GiB/s":
                    res[(w[0], w[1])] = float(w[2])
        os.unlink(CORPUS_FILE)
        base = libs[0][0]
        print("\nrelative to %s:" % base)
        for name, _ in libs:
            if all((name, f) in res and (base, f) in res for f in fams):
                print("%-14s " % name + "  ".join("%s %+5.1f%%" % (f, (res[(name, f)] / res[(base, f)] - 1) * 100) for f in fams))
        return
    corp = pickle.load(open(CORPUS_FILE, "rb"))
    res = {}
    for name, path in libs:
        L = load(path)
        ctx = ctypes.c_void_p()
        assert L.xlz_ctx_create(0, ctypes.byref(ctx)) == 0
        for f in fams:
            cs, hs, rep, size = corp[f]
            n = len(cs) * rep
            descs = (N.StreamDesc * n)()
            keep = [ctypes.create_string_buffer(c, len(c)) for c in cs]
            for i in range(n):
                descs[i].inp = ctypes.cast(keep[i % len(cs)], ctypes.c_void_p)
                descs[i].in_len = len(cs[i % len(cs)])
                descs[i].out_cap = size
                descs[i].format = 0
            b = ctypes.c_void_p()
            assert L.xlz_batch_create(ctx, descs, n, ctypes.byref(b)) == 0
            L.xlz_batch_run(b)
            L.xlz_batch_sync(b)
            L.xlz_ctx_event_record(ctx, 0)
            for _ in range(steps):
                L.xlz_batch_run(b)
            L.xlz_ctx_event_record(ctx, 1)
            L.xlz_batch_sync(b)
            ms = ctypes.c_float()
            L.xlz_ctx_event_elapsed_ms(ctx, 0, 1, ctypes.byref(ms))
            r = (N.Result * n)()
            assert L.xlz_batch_results(b, r) == 0
            ok = all(r[i].status == 0 and r[i].out_len == size for i in range(n))
            buf = ctypes.create_string_buffer(size)
            for i in list(range(0, n, max(1, n // 48))) + [n - 1]:
                L.xlz_batch_download(b, i, ctypes.cast(buf, ctypes.c_void_p), size)
                ok = ok and hashlib.sha256(buf.raw).digest() == hs[i % len(cs)]
            L.xlz_batch_destroy(b)
            gibs = n * size * steps / (1 << 30) / (ms.value / 1e3)
            res[(name, f)] = gibs
            print("%-14s %s  %7.3f GiB/s  (%.2f ms/launch)%s" % (name, f, gibs, ms.value / steps, "" if ok else "   *** WRONG OUTPUT ***"),
                  flush=True)
        L.xlz_ctx_destroy(ctx)


if __name__ == "__main__":
    main()
