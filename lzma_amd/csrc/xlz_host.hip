// xlz_host.hip -- host side of libxlz.so: the C ABI of include/xlz.h.
//
// Host work is limited to what the reference does outside its hot loop:
// parsing the 13-byte .lzma header (reader1.go:77-147), scanning LZMA2 chunk
// headers to place units (reader2.go:100-214), moving bytes to and from HBM and
// launching the gfx950 kernels.  No byte of any stream is DECODED on the host:
// if no HIP device is usable every decode entry point returns XLZ_ERR_DEVICE.
//
// file:line citations are into the reference repository (kulaginds/lzma).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <deque>
#include <functional>
#include <memory>
#include <thread>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <numeric>
#include <vector>

#include "../../include/xlz.h"
#include "xlz_format.h"

using namespace xlz;

namespace {

constexpr uint32_t kLzmaDicMin = 1u << 12;        // types.go:8
constexpr uint64_t kUnknownSize = ~(uint64_t)0;   // state.go:135-151
constexpr uint64_t kMaxUnitBytes = 0xFFFF0000ull; // 32-bit offsets inside a unit
constexpr size_t kArenaAlign = 256;
constexpr size_t kStoredUnitBytes = 256 * 1024; // scan_lzma2: a run of stored chunks is cut into units of at least this much (four
                                               // or five of liblzma's chunks: finer units cost a launch of such units more in
                                               // per-unit work than their evener tail gives back, profiles/r03/ab_stored_unit.txt)
constexpr size_t kArenaTailPad = 1024; // the 256-byte input window may start near a unit's end
constexpr size_t kOutTailPad = 64;    // wave_copy stores whole 64-lane rows: scratch bytes past a unit's end
// xlz_decode_batch, a call of one wave round: run it as a sequence of launches (slices) from this much output on, a slice
// for every kSliceBytes of it, at most kMaxSlices (profiles/r05/slices_scan.txt)
constexpr uint32_t kWideGrid = 32;                // workgroups of the widest-model re-run (6 MiB of model each)
constexpr int XLZ_ERR_NO_MEMORY_INTERNAL = -1000; // run_units: the widest models could not be allocated (never leaves this file)
constexpr uint64_t kSlicedCallBytes = 256ull << 20;
constexpr uint64_t kSliceBytes = 128ull << 20;
constexpr uint64_t kMaxSlices = 8;
// ... and its first launch starts when 1.5 x the first slice's share of the output (+ 4 KiB, xlz_format.h: slice_head) is
// there of every unit's INPUT -- a stream's first bytes compress worst; a unit that runs out of input pauses and goes on
// with the second launch
// xlz_batch_advice: a host core decodes a stream this many times as fast as one wave decodes a unit (65-80 MB/s of output
// against 4-6: bench.py's stream_count_sweep puts the break-even of equal LZMA1 streams at 256 units per 16 threads)
constexpr uint32_t kCoreOverWave = 16;
static inline uint32_t head_frac_for(uint32_t slices) { return slices >= 2 ? (uint32_t)(65536ull * 3 / (2 * slices)) : 0; }

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

#define HIP_TRY(expr)                                                                             \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            if (getenv("XLZ_DEBUG"))                                                              \
                fprintf(stderr, "xlz: %s failed: %s\n", #expr, hipGetErrorString(e_));            \
            return XLZ_ERR_DEVICE;                                                                \
        }                                                                                         \
    } while (0)

} // namespace

struct xlz_reader;
struct Batcher;
static void batcher_shutdown(Batcher *bt); // defined next to the readers

// Device (and pinned) memory of the (sub-)batches xlz_decode_batch makes is kept by the context between calls: a call of
// nine sub-batches makes and drops some sixty allocations, hipFree waits for the device every time, and the next call of
// the same shape wants the same sizes again.  A block is reused for a request of at least two thirds of its size; what
// two calls in a row did not touch is released (trim), xlz_ctx_trim releases everything.  Batches made through the
// public xlz_batch_create are not pooled (they live as long as their owner wants).
struct MemPool {
    struct Blk {
        void *p;
        size_t cap;
        uint64_t gen;
        bool pinned;
    };
    std::mutex mu;
    std::vector<Blk> idle;
    std::vector<Blk> live;
    uint64_t gen = 1;
    void *take(size_t n, bool pinned)
    {
        if (n == 0) n = 1;
        {
            std::lock_guard<std::mutex> lk(mu);
            size_t best = idle.size();
            for (size_t i = 0; i < idle.size(); i++)
                if (idle[i].pinned == pinned && idle[i].cap >= n && idle[i].cap <= n + n / 2 + (1u << 16) &&
                    (best == idle.size() || idle[i].cap < idle[best].cap))
                    best = i;
            if (best != idle.size()) {
                Blk b = idle[best];
                idle.erase(idle.begin() + (long)best);
                b.gen = gen;
                live.push_back(b);
                return b.p;
            }
        }
        void *p = nullptr;
        const hipError_t e = pinned ? hipHostMalloc(&p, n, hipHostMallocDefault) : hipMalloc(&p, n);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            drop_idle(); // the cache may be what is in the way: release it and try once more
            if ((pinned ? hipHostMalloc(&p, n, hipHostMallocDefault) : hipMalloc(&p, n)) != hipSuccess) return nullptr;
        }
        std::lock_guard<std::mutex> lk(mu);
        live.push_back(Blk{p, n, gen, pinned});
        return p;
    }
    void give(void *p) // (the caller knows that no work on the device still uses the block)
    {
        if (!p) return;
        std::lock_guard<std::mutex> lk(mu);
        for (size_t i = 0; i < live.size(); i++)
            if (live[i].p == p) {
                idle.push_back(live[i]);
                live.erase(live.begin() + (long)i);
                return;
            }
    }
    void end_of_call() // a call is over: blocks that this call and the one before did not use go back to the system
    {
        std::vector<Blk> out;
        {
            std::lock_guard<std::mutex> lk(mu);
            for (size_t i = 0; i < idle.size();)
                if (idle[i].gen + 1 < gen) {
                    out.push_back(idle[i]);
                    idle.erase(idle.begin() + (long)i);
                } else {
                    i++;
                }
            gen++;
        }
        for (const Blk &b : out) (void)(b.pinned ? hipHostFree(b.p) : hipFree(b.p));
    }
    void drop_idle()
    {
        std::vector<Blk> out;
        {
            std::lock_guard<std::mutex> lk(mu);
            out.swap(idle);
        }
        for (const Blk &b : out) (void)(b.pinned ? hipHostFree(b.p) : hipFree(b.p));
    }
    uint64_t idle_bytes()
    {
        std::lock_guard<std::mutex> lk(mu);
        uint64_t t = 0;
        for (const Blk &b : idle) t += b.cap;
        return t;
    }
};

// Host side of the boundary (host buffers in, host buffers out): pinned staging that lives as
// long as the context, so that neither page pinning nor pageable copies sit on the path.
//  * input: one grow-only pinned image, packed by several host threads, one async H2D copy;
//  * output: a ring of pinned buffers; D2H copies of arena chunks run on their own stream while
//    host threads scatter the chunks that have arrived into the callers' buffers.
constexpr int kEventSlots = 64; // xlz_ctx_event_record: enough for one event per timed step of bench.py

struct HostPipe {
    static constexpr int kRing = 8;                 // (round 5: eight slots of 32 MiB instead of four of 64: as many scatter threads
    static constexpr size_t kRingBytes = 32u << 20; //  work on the LAST slice of a call, whose scatter nothing overlaps)
    std::mutex mu_in;  // the pinned input image: one pack + upload at a time per context (PinLease)
    std::condition_variable cv_in;
    bool in_busy = false;
    std::mutex mu_out; // the pinned output ring: one download at a time per context
    uint8_t *pin_in = nullptr;
    size_t pin_in_cap = 0;
    uint8_t *ring[kRing] = {};
    hipEvent_t ring_ev[kRing] = {};
    hipStream_t copy_stream = nullptr;
    hipStream_t up_stream = nullptr; // H2D of the pinned input image, piece by piece while the rest is still being packed
};

// The pinned input image is taken for one pack + upload; a sliced batch keeps it until the tails of its inputs have left it
// (they are uploaded while the first launch decodes), which may be on another thread than the one that took it.
struct PinLease {
    HostPipe *hp = nullptr;
    void take(HostPipe &h)
    {
        std::unique_lock<std::mutex> lk(h.mu_in);
        h.cv_in.wait(lk, [&] { return !h.in_busy; });
        h.in_busy = true;
        hp = &h;
    }
    void release()
    {
        if (!hp) return;
        {
            std::lock_guard<std::mutex> lk(hp->mu_in);
            hp->in_busy = false;
        }
        hp->cv_in.notify_one();
        hp = nullptr;
    }
    PinLease() = default;
    PinLease(const PinLease &) = delete;
    PinLease &operator=(const PinLease &) = delete;
    ~PinLease() { release(); }
};

struct xlz_ctx {
    Batcher *batcher = nullptr; // coalesces pull-style readers into batches (xlz_ctx_enable_batching)
    int device = 0;
    int num_cus = 0;
    hipStream_t stream = nullptr;
    uint32_t *queue = nullptr; // work-queue head, re-zeroed on the stream before each launch
    // the sub-batches of a pipelined xlz_decode_batch alternate between two streams (each with a queue of its own): launch
    // k + 1 is not held behind launch k, its workgroups take the wave slots that k's workgroups leave when k's queue has
    // run dry -- no round of a sub-batch ends with idle slots while the next sub-batch waits
    hipStream_t stream2 = nullptr;
    uint32_t *queue2 = nullptr;
    uint32_t *prio_tab = nullptr; // LaunchParams.prio_tab: one word per hardware wave slot, zero when idle
    hipEvent_t ev[kEventSlots] = {};
    std::mutex mu;
    HostPipe pipe;
    MemPool pool; // device / pinned blocks of xlz_decode_batch's (sub-)batches, kept between calls
    xlz_call_stats last_call = {}; // of the most recent xlz_decode_batch on this context (xlz_ctx_last_call_stats)
    bool have_last_call = false;
    // xlz_ctx_set_slicing: when a call of one wave round runs as a sequence of launches, and as how many
    uint64_t sliced_call_bytes = kSlicedCallBytes, slice_bytes = kSliceBytes;
    uint32_t max_slices = kMaxSlices;
};

// per-stream bookkeeping of a batch
struct StreamPlan {
    int32_t host_status = 1; // 1 = decided on the device; otherwise final status from parsing
    uint64_t host_in_consumed = 0;
    uint32_t header_len = 0;
    uint32_t first_unit = 0, n_units = 0;
    uint64_t out_off = 0; // into the output arena
    uint64_t out_cap = 0; // arena bytes reserved
    uint64_t in_off = 0;  // LZMA2: arena offset of the stream's first byte
    uint32_t in_len = 0;  // LZMA2: stream length
    uint32_t dict_size = 0;
    bool lzma2 = false;
    bool slice = false;    // XLZ_STREAM_F_LZMA2_SLICE: earlier units of the stream are not in this call
    bool oversize = false; // >= 4 GiB of input or output: not in the arenas, decoded as a session by xlz_decode_batch
};

struct xlz_batch {
    xlz_ctx *ctx = nullptr;
    size_t n = 0;
    std::vector<StreamPlan> plans;
    std::vector<Unit> units;
    std::vector<uint32_t> order;
    uint8_t *d_in = nullptr;
    uint8_t *d_out = nullptr;
    Unit *d_units = nullptr;
    uint32_t *d_order = nullptr;
    UnitResult *d_results = nullptr;
    size_t in_bytes = 0, out_bytes = 0;
    uint32_t max_pb = 0;        // over the units whose model fits LDS
    bool compact = false;       // ... and so their launch uses the compact model layout (xlz_format.h: ModelLayout<true>, pb <= 2)
    uint32_t max_lc_lp = 0;     // over the units whose model fits LDS
    uint32_t max_lc_lp_big = 0; // over the units decoded with an HBM-resident model
    uint32_t n_normal = 0;      // order[0, n_normal) = LDS units, the rest = big-model units
    uint16_t *d_scratch = nullptr; // HBM-model launch: model + matched-literal part per workgroup
    uint32_t scratch_stride = 0;
    uint16_t *d_mlit = nullptr;    // LDS-model launch: matched-literal part per workgroup
    uint32_t mlit_stride = 0;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipStream_t run_stream = nullptr; // where the batch's launches go (the context's stream, or its second one)
    uint32_t *queue = nullptr;        // ... and their work-queue head
    bool pooled = false;      // made by xlz_decode_batch: its memory comes from and returns to the context's pool
    bool quiet = false;       // ... and nothing on the device uses it any more (else batch_free waits for the device first)
    uint32_t call_units = 0;  // a sub-batch of a pipelined call of that many streams: wave slots per CU as for ONE launch over all of them
    bool ran = false;
    uint64_t algo_in = 0; // compressed payload bytes handed to the device
    // per-stream results of the latest run (filled lazily by collect())
    std::vector<xlz_result> final_results;
    std::vector<UnitResult> unit_results; // of the main launch (timestamps: xlz_batch_unit_trace)
    bool collected = false;
    uint64_t sum_in = 0, sum_out = 0;
    std::vector<size_t> rewritten; // streams whose bytes a re-run of collect() wrote again (exact / widest-model launches)
    std::vector<char> wants_wide;  // per stream: it ended in front of a chunk whose properties need a larger model (AUX_GROW)
    // A SLICED batch (xlz_decode_batch, a call of one wave round: LaunchParams.slice_*): the run is a sequence of launches,
    // launch k advances every unit to its k-th output bound; every unit has a state block; slice_ev[k] is recorded behind
    // launch k, and the bytes in front of the k-th bounds are downloaded while launch k + 1 decodes (download_sliced).
    std::vector<uint32_t> slice_fracs; // bound of launch k in 1/65536 of a unit's out_cap; the last one is kSliceOne
    std::vector<hipEvent_t> slice_ev;
    uint8_t *d_states = nullptr;
    size_t state_stride = 0;
    std::vector<std::vector<SlicePiece>> slice_pieces; // per launch: what it finishes of every unit, packed back to back
    std::vector<uint64_t> slice_pack_bytes;
    SlicePiece *d_pieces = nullptr; // all launches' tables, one after the other
    uint8_t *d_pack = nullptr;      // the packed image of one launch's pieces (xlz_gather_kernel), downloaded linearly
    UnitResult *pin_res = nullptr;  // pinned: the unit results behind every launch (slice_fracs.size() x units)
    // heads first: the first launch starts when the first head_frac / 65536 (+ 4 KiB) of every unit's input is on the
    // device (LaunchParams.head_frac); the tails follow while it decodes.  The pinned image and the device staging range
    // are packed heads | tails; a scatter kernel puts each part into the input arena.
    uint32_t head_frac = 0;
    uint8_t *d_stage = nullptr;
    SlicePiece *d_up_pieces = nullptr;
    hipEvent_t ev_heads = nullptr, ev_tails = nullptr;
    std::function<int()> upload_tails; // pending second half of the upload, run by xlz_batch_run behind the first launch
    PinLease in_lease;
};

// ---------------------------------------------------------------- helpers ----
extern "C" const char *xlz_version(void) { return "xlz 0.1 (gfx950)"; }

#ifndef XLZ_BUILD_ID // lzma_amd/build.py: hash of every source file this library was compiled from
#define XLZ_BUILD_ID "unknown"
#endif
extern "C" const char *xlz_build_id(void) { return XLZ_BUILD_ID; }
#ifndef XLZ_KERNEL_ID // hash of the device code's sources only (xlz_kernel.hip, xlz_fastpath.inc, xlz_format.h)
#define XLZ_KERNEL_ID "unknown"
#endif
extern "C" const char *xlz_kernel_id(void) { return XLZ_KERNEL_ID; }

extern "C" const char *xlz_strerror(int st)
{
    switch (st) {
    case XLZ_OK: return "ok";
    case XLZ_OK_INPUT_EOF: return "ok (input ended; reference reports io.EOF)";
    case XLZ_ERR_RESULT: return "result error";                            // errors.go:8
    case XLZ_ERR_PROPS: return "incorrect LZMA properties";                // errors.go:7
    case XLZ_ERR_HEADER_EOF: return "EOF";                                 // io.EOF out of a constructor
    case XLZ_ERR_RC_INIT: return "rangeDec.Init: result error";            // reader1.go:155
    case XLZ_ERR_UNEXPECTED_EOF: return "unexpected EOF";                  // io.ErrUnexpectedEOF
    case XLZ_ERR_OUT_CAP: return "output capacity too small";
    case XLZ_ERR_BAD_ARG: return "bad argument";
    case XLZ_ERR_DEVICE: return "HIP device error";
    case XLZ_ERR_UNSUPPORTED: return "stream not supported by the GPU path";
    case XLZ_ERR_CLOSED: return "lzma: already closed";                    // readcloser.go:14
    case XLZ_ERR_NEED_ONE_READER: return "lzma: need exactly one reader";  // reader1.go:26
    case XLZ_ERR_INSUFFICIENT_PROPS: return "lzma2: not enough properties"; // reader2.go:43
    case XLZ_EOF: return "EOF";
    }
    return "unknown status";
}

extern "C" int xlz_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// DecodeProp, reader1.go:210-221
extern "C" int xlz_decode_prop(uint8_t d, uint8_t *lc, uint8_t *pb, uint8_t *lp)
{
    if (d >= 9 * 5 * 5) return XLZ_ERR_PROPS;
    if (lc) *lc = d % 9;
    d /= 9;
    if (pb) *pb = d / 5;
    if (lp) *lp = d % 5;
    return XLZ_OK;
}

// DecodeDictSize, reader1.go:193-208
extern "C" uint32_t xlz_decode_dict_size(const uint8_t p[4])
{
    uint32_t d = 0;
    for (int i = 0; i < 4; i++) d |= (uint32_t)p[i] << (8 * i);
    return d < kLzmaDicMin ? kLzmaDicMin : d;
}

// DecodeDictSize2, reader2.go:296-298
extern "C" uint32_t xlz_decode_dict_size2(uint8_t b)
{
    const unsigned sh = (unsigned)(b / 2 + 11);
    return sh >= 32 ? 0u : (uint32_t)(2u | (b & 1u)) << sh;
}

// DecodeUnpackSize, reader1.go:178-191
extern "C" uint64_t xlz_decode_unpack_size(const uint8_t h[8])
{
    uint64_t u = 0;
    for (int i = 0; i < 8; i++) u |= (uint64_t)h[i] << (8 * i);
    return u;
}

// ---------------------------------------------------------------- context ----
namespace {

// host threads for a staging job of `bytes` bytes: one per 32 MiB, at most 8
unsigned host_threads(size_t bytes)
{
    unsigned cap = 8;
    const unsigned hw = std::thread::hardware_concurrency();
    if (hw && hw < cap) cap = hw;
    const unsigned want = (unsigned)(bytes / (32u << 20)) + 1;
    return want < cap ? want : cap;
}

template <class F> void run_threads(unsigned n, F &&f)
{
    if (n <= 1) {
        f(0u);
        return;
    }
    std::vector<std::thread> th;
    th.reserve(n - 1);
    for (unsigned t = 1; t < n; t++) th.emplace_back([&f, t] { f(t); });
    f(0u);
    for (auto &x : th) x.join();
}

} // namespace

extern "C" int xlz_ctx_create(int device, xlz_ctx **out)
{
    if (!out) return XLZ_ERR_BAD_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return XLZ_ERR_DEVICE;
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    xlz_ctx *c = new (std::nothrow) xlz_ctx;
    if (!c) return XLZ_ERR_BAD_ARG;
    c->device = device;
    c->num_cus = prop.multiProcessorCount;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess ||
        hipMalloc(&c->queue, 512) != hipSuccess || hipMalloc(&c->queue2, 512) != hipSuccess ||
        hipMalloc(&c->prio_tab, kPrioTabWords * sizeof(uint32_t)) != hipSuccess ||
        hipMemset(c->prio_tab, 0, kPrioTabWords * sizeof(uint32_t)) != hipSuccess) {
        if (c->queue) (void)hipFree(c->queue);
        if (c->queue2) (void)hipFree(c->queue2);
        if (c->stream) (void)hipStreamDestroy(c->stream);
        if (c->stream2) (void)hipStreamDestroy(c->stream2);
        if (c->prio_tab) (void)hipFree(c->prio_tab);
        delete c;
        return XLZ_ERR_DEVICE;
    }
    *out = c;
    return XLZ_OK;
}

extern "C" void xlz_ctx_destroy(xlz_ctx *c)
{
    if (!c) return;
    if (c->batcher) batcher_shutdown(c->batcher);
    (void)hipSetDevice(c->device);
    c->pool.drop_idle();
    if (c->queue) (void)hipFree(c->queue);
    if (c->queue2) (void)hipFree(c->queue2);
    if (c->prio_tab) (void)hipFree(c->prio_tab);
    if (c->pipe.pin_in) (void)hipHostFree(c->pipe.pin_in);
    for (int i = 0; i < HostPipe::kRing; i++) {
        if (c->pipe.ring[i]) (void)hipHostFree(c->pipe.ring[i]);
        if (c->pipe.ring_ev[i]) (void)hipEventDestroy(c->pipe.ring_ev[i]);
    }
    if (c->pipe.copy_stream) (void)hipStreamDestroy(c->pipe.copy_stream);
    if (c->pipe.up_stream) (void)hipStreamDestroy(c->pipe.up_stream);
    for (hipEvent_t e : c->ev)
        if (e) (void)hipEventDestroy(e);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    delete c;
}

extern "C" int xlz_ctx_device(const xlz_ctx *c) { return c ? c->device : -1; }

extern "C" int xlz_ctx_event_record(xlz_ctx *c, int slot)
{
    if (!c || slot < 0 || slot >= kEventSlots) return XLZ_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(c->mu);
    HIP_TRY(hipSetDevice(c->device));
    if (!c->ev[slot]) HIP_TRY(hipEventCreate(&c->ev[slot]));
    HIP_TRY(hipEventRecord(c->ev[slot], c->stream));
    return XLZ_OK;
}

extern "C" int xlz_ctx_event_elapsed_ms(xlz_ctx *c, int a, int b, float *ms)
{
    if (!c || !ms || a < 0 || a >= kEventSlots || b < 0 || b >= kEventSlots || !c->ev[a] || !c->ev[b]) return XLZ_ERR_BAD_ARG;
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipEventSynchronize(c->ev[b]));
    HIP_TRY(hipEventElapsedTime(ms, c->ev[a], c->ev[b]));
    return XLZ_OK;
}

// ------------------------------------------------------------------ batch ----
namespace {

// What NewReader1's initializeFull does before the first Read (reader1.go:77-101).
// Fills `u` for the device, or settles the stream's status on the host.
void plan_lzma_alone(const xlz_stream_desc &s, StreamPlan &pl, Unit &u, bool &has_unit)
{
    has_unit = false;
    if (s.in_len == 0) { // ReadByte fails at once: constructor returns io.EOF
        pl.host_status = XLZ_ERR_HEADER_EOF;
        pl.host_in_consumed = 0;
        return;
    }
    uint8_t lc, pb, lp;
    if (xlz_decode_prop(s.in[0], &lc, &pb, &lp) != XLZ_OK) { // "decode prop: %w"
        pl.host_status = XLZ_ERR_PROPS;
        pl.host_in_consumed = 1;
        return;
    }
    if (s.in_len < 13) { // "decode dict size" / "decode unpack size": EOF
        pl.host_status = XLZ_ERR_HEADER_EOF;
        pl.host_in_consumed = s.in_len;
        return;
    }
    u.dict_size = xlz_decode_dict_size(s.in + 1);
    u.unpack_size = xlz_decode_unpack_size(s.in + 5);
    u.lc = lc;
    u.lp = lp;
    u.pb = pb;
    u.kind = UNIT_LZMA1;
    pl.header_len = 13;
    has_unit = true;
}

// NewLZMADecompressorForSevenZip's view: header fields out of band (reader1.go:32-61)
void plan_lzma_raw(const xlz_stream_desc &s, StreamPlan &pl, Unit &u, bool &has_unit)
{
    has_unit = false;
    uint8_t lc, pb, lp;
    if (xlz_decode_prop(s.props, &lc, &pb, &lp) != XLZ_OK) {
        pl.host_status = XLZ_ERR_PROPS;
        pl.host_in_consumed = 0;
        return;
    }
    u.dict_size = s.dict_size < kLzmaDicMin ? kLzmaDicMin : s.dict_size; // reader1.go:199-201
    u.unpack_size = s.unpack_size;
    u.lc = lc;
    u.lp = lp;
    u.pb = pb;
    u.kind = UNIT_LZMA1;
    pl.header_len = 0;
    has_unit = true;
}

// LZMA2: walk the chunk headers the way Reader2.startChunk does (reader2.go:100-214), trusting
// each header's sizes, and cut the stream into units that are independent of everything
// before them: a unit starts at a chunk that resets the dictionary AND whose first
// compressed chunk carries new properties (hence resets the model too).  Every chunk
// header states its uncompressed size, so the output offset of each unit is known up front.
// A stream whose real decode does not follow its headers is detected after the launch
// (a unit's produced / consumed counts differ) and re-decoded as ONE unit.
struct Lz2Unit {
    uint64_t in_start, in_len; // 64-bit: xlz_lzma2_units plans streams of any size (the batch path holds a stream in 32 bits)
    uint64_t out_start, expect_out;
    bool have_reader; // an LZMA chunk precedes the unit: Reader2.lzmaReader exists (reader2.go:146-153)
};

// bytes a unit made of stored chunks alone should hold at least.  The shipped library uses kStoredUnitBytes and reads
// nothing from the environment here; an A/B build (-DXLZ_DEV_KNOBS, tools/ab_stored_unit.sh) takes XLZ_STORED_UNIT_KIB
// (0 = cut runs of stored chunks at dictionary resets only; clamped to 1 GiB).
uint64_t stored_unit_bytes()
{
#ifdef XLZ_DEV_KNOBS
    static const uint64_t v = [] {
        const char *e = getenv("XLZ_STORED_UNIT_KIB");
        if (!e) return (uint64_t)kStoredUnitBytes;
        unsigned long long k = strtoull(e, nullptr, 10);
        return (uint64_t)(k > (1u << 20) ? (1u << 20) : k) << 10;
    }();
    return v;
#else
    return kStoredUnitBytes;
#endif
}

// max_pb (optional): the largest pb any properties byte of the stream announces (the model's layout depends on it)
void scan_lzma2(const uint8_t *in, size_t len, std::vector<Lz2Unit> &units, uint32_t &max_lc_lp, uint32_t *max_pb = nullptr)
{
    size_t pos = 0, unit_start = 0;
    uint64_t out = 0, unit_out = 0;
    // pending splits: stored chunks that reset the dictionary.  A unit may start at one iff the first LZMA chunk behind
    // it (if any) brings new properties -- a chunk without them continues the model that LZMA chunks in FRONT of the
    // stored ones left (the state is carried across stored chunks, reader2.go:155-167).  So candidates wait for the next
    // LZMA chunk: new properties make all of them cuts, none does; the end of the stream makes them cuts too (nothing
    // depends on a model any more: a run of stored chunks -- the shape of the reference's own LZMA2 benchmark file,
    // randomfile.dat.lzma2 -- is then copied by one wave per dictionary reset instead of one wave for all of it).
    struct Cand {
        size_t pos;
        uint64_t out;
        bool seen_lzma;
    };
    std::vector<Cand> cands;
    // Inside a run of stored chunks that nothing behind it reads -- the run ends at the next dictionary reset or at the
    // end of the stream -- EVERY chunk boundary can start a unit: a stored chunk reads no history (window.ReadFrom,
    // window.go:142-155) and leaves nothing but window bytes behind, which only a later LZMA chunk WITHOUT a
    // dictionary reset could look at.  `marks` = the stored chunks behind the first pending candidate that do not reset
    // the dictionary themselves; a pure run is cut at them whenever kStoredUnit bytes have gathered (one incompressible
    // file inside one LZMA2 stream is then copied by thousands of waves, not by one).
    struct Mark {
        size_t pos;
        uint64_t out;
    };
    std::vector<Mark> marks;
    const uint64_t stored_unit = stored_unit_bytes();
    bool seen_lzma = false, unit_seen_lzma = false; // an LZMA chunk before: here / the unit
    auto cut = [&](size_t at, uint64_t at_out, bool lzma_before) {
        units.push_back({(uint64_t)unit_start, (uint64_t)(at - unit_start), unit_out, at_out - unit_out, unit_seen_lzma});
        unit_start = at;
        unit_out = at_out;
        unit_seen_lzma = lzma_before;
    };
    // last_pure: the run behind the LAST candidate ends here too (end of the stream, or an LZMA chunk that resets the
    // dictionary); otherwise an LZMA chunk that keeps the dictionary follows it and must find the run in its own unit
    auto cut_at_candidates = [&](bool last_pure) {
        size_t mi = 0;
        for (size_t i = 0; i < cands.size(); i++) {
            const Cand &c = cands[i];
            if (c.pos != unit_start) cut(c.pos, c.out, c.seen_lzma);
            const size_t region_end = i + 1 < cands.size() ? cands[i + 1].pos : (size_t)-1;
            const bool pure = i + 1 < cands.size() || last_pure;
            for (; mi < marks.size() && marks[mi].pos < region_end; mi++)
                if (pure && stored_unit && marks[mi].pos > c.pos && marks[mi].out - unit_out >= stored_unit)
                    cut(marks[mi].pos, marks[mi].out, c.seen_lzma);
        }
        cands.clear();
        marks.clear();
    };
    bool ended = false; // the walk reached the end of the stream (or of the input) without leaving the format
    while (pos < len) {
        const uint8_t c = in[pos];
        if (c == 0 || (c >= 3 && c < 0x80)) { // end of stream (reader2.go:175-199)
            pos++;
            ended = true;
            break;
        }
        const bool stored = c < 3;
        const unsigned sub = c >> 5;
        const size_t hl = stored ? 3 : (sub >= 6 ? 6 : 5);
        if (pos + hl > len) break; // truncated header: the device walker reports it
        uint32_t unc = ((uint32_t)in[pos + 1] << 8) | in[pos + 2];
        if (stored) {
            unc += 1;
            if (c == 1 || pos == 0) // (the stream's first chunk starts a unit anyway: a candidate that is never a cut)
                cands.push_back({pos, out, seen_lzma});
            else if (!cands.empty())
                marks.push_back({pos, out});
            const size_t body = std::min<size_t>(unc, len - pos - hl);
            pos += hl + body;
            out += body;
            continue;
        }
        unc = (unc | ((uint32_t)(c & 0x1F) << 16)) + 1;
        const size_t comp = (((size_t)in[pos + 3] << 8) | in[pos + 4]) + 1;
        if (sub >= 6) {
            const uint8_t props = in[pos + 5];
            if (props >= 225) break; // the walker reports ErrIncorrectProperties here
            max_lc_lp = std::max<uint32_t>(max_lc_lp, (props % 9) + (props / 9) % 5);
            if (max_pb) *max_pb = std::max<uint32_t>(*max_pb, props / 45u);
            if (sub == 7 && pos != 0 && pos != unit_start) {
                cut_at_candidates(true);
                if (pos != unit_start) cut(pos, out, seen_lzma);
            } else {
                cut_at_candidates(sub == 7);
            }
        }
        seen_lzma = true;
        cands.clear(); // a compressed chunk without new props keeps the model: no cut
        marks.clear();
        pos += hl + std::min(comp, len - pos - hl);
        out += unc;
    }
    if (ended || pos >= len) cut_at_candidates(true); // only stored chunks behind them
    units.push_back({(uint64_t)unit_start, (uint64_t)(len - unit_start), unit_out, out - unit_out, unit_seen_lzma});
}

} // namespace

extern "C" int xlz_lzma2_units(const uint8_t *in, size_t len, xlz_lzma2_unit *units, size_t max_units, size_t *n_units)
{
    if ((!in && len) || !n_units || (!units && max_units)) return XLZ_ERR_BAD_ARG;
    std::vector<Lz2Unit> lu;
    uint32_t mx = 0;
    scan_lzma2(in, len, lu, mx);
    *n_units = lu.size();
    for (size_t k = 0; k < lu.size() && k < max_units; k++) {
        units[k].in_off = lu[k].in_start;
        units[k].in_len = lu[k].in_len;
        units[k].out_off = lu[k].out_start;
        units[k].out_len = lu[k].expect_out;
        units[k].have_reader = lu[k].have_reader ? 1u : 0u;
        units[k].reserved = 0;
    }
    return max_units && lu.size() > max_units ? XLZ_ERR_OUT_CAP : XLZ_OK;
}

extern "C" int xlz_batch_advice(const xlz_ctx *ctx, const xlz_stream_desc *streams, size_t n, uint32_t host_threads,
                                xlz_advice *out)
{
    if (!out || (!streams && n)) return XLZ_ERR_BAD_ARG;
    memset(out, 0, sizeof *out);
    // work is counted in compressed bytes (decode time tracks the number of binary decisions, which tracks them: the key
    // of the kernel's own work queue)
    uint64_t max_unit = 0, max_stream = 0;
    for (size_t i = 0; i < n; i++) {
        const xlz_stream_desc &s = streams[i];
        if (!s.in && s.in_len) return XLZ_ERR_BAD_ARG;
        out->in_bytes += s.in_len;
        size_t units = 0;
        if (s.format == XLZ_FMT_LZMA2_RAW) { // the plan a decode would launch: units between dictionary resets
            std::vector<Lz2Unit> lu;
            uint32_t mx = 0;
            scan_lzma2(s.in, s.in_len, lu, mx);
            units = lu.size();
            for (const Lz2Unit &u : lu) max_unit = std::max<uint64_t>(max_unit, u.in_len);
        } else if (s.format == XLZ_FMT_LZMA_ALONE) {
            units = s.in_len > 13 ? 1 : 0; // (a header alone is settled on the host)
            if (units) max_unit = std::max<uint64_t>(max_unit, s.in_len);
        } else {
            units = s.in_len ? 1 : 0;
            if (units) max_unit = std::max<uint64_t>(max_unit, s.in_len);
        }
        out->units += units;
        if (units) max_stream = std::max<uint64_t>(max_stream, s.in_len);
    }
    const uint32_t cus = ctx ? (uint32_t)ctx->num_cus : 256u;
    out->wave_slots = cus * 16u;
    if (!host_threads) host_threads = std::max(1u, std::thread::hardware_concurrency());
    out->break_even_units = kCoreOverWave * host_threads;
    out->fill = out->wave_slots ? std::min(1.0, (double)out->units / out->wave_slots) : 0.0;
    // The host decodes ONE stream on ONE thread whatever its format (a Reader2 is one goroutine, reader2.go:216-250: the
    // units of an LZMA2 stream are parallel work for the GPU only), kCoreOverWave times as fast as a wave decodes a unit;
    // both sides are list-scheduled: the longest piece of serial work, or the whole call over the workers
    out->cpu_cost = (double)std::max<uint64_t>(max_stream, (out->in_bytes + host_threads - 1) / host_threads) / kCoreOverWave;
    out->gpu_cost = (double)std::max<uint64_t>(max_unit, out->wave_slots ? (out->in_bytes + out->wave_slots - 1) / out->wave_slots : 0);
    out->prefer_cpu = (out->units == 0 || out->cpu_cost < out->gpu_cost) ? 1 : 0;
    return XLZ_OK;
}

namespace {

// memory of a batch: the context's pool for the batches xlz_decode_batch makes, the runtime otherwise
template <class T> bool batch_alloc(xlz_batch *b, T **p, size_t bytes, bool pinned = false)
{
    if (b->pooled) {
        *p = static_cast<T *>(b->ctx->pool.take(bytes, pinned));
        return *p != nullptr;
    }
    return (pinned ? hipHostMalloc(reinterpret_cast<void **>(p), bytes, hipHostMallocDefault) : hipMalloc(reinterpret_cast<void **>(p), bytes)) ==
           hipSuccess;
}
void batch_release(xlz_batch *b, void *p, bool pinned = false)
{
    if (!p) return;
    if (b->pooled)
        b->ctx->pool.give(p);
    else
        (void)(pinned ? hipHostFree(p) : hipFree(p));
}

int batch_free(xlz_batch *b)
{
    if (!b) return XLZ_OK;
    if (b->ctx) (void)hipSetDevice(b->ctx->device);
    if (b->pooled && !b->quiet) (void)hipDeviceSynchronize(); // (a block must not return to the pool while work still uses it)
    batch_release(b, b->d_in);
    batch_release(b, b->d_out);
    batch_release(b, b->d_units);
    batch_release(b, b->d_order);
    batch_release(b, b->d_results);
    batch_release(b, b->d_scratch);
    batch_release(b, b->d_mlit);
    if (b->ev0) (void)hipEventDestroy(b->ev0);
    if (b->ev1) (void)hipEventDestroy(b->ev1);
    for (hipEvent_t e : b->slice_ev)
        if (e) (void)hipEventDestroy(e);
    batch_release(b, b->d_stage); // (hipFree waits for the device: a pending upload has left the pinned image)
    batch_release(b, b->d_up_pieces);
    if (b->ev_heads) (void)hipEventDestroy(b->ev_heads);
    if (b->ev_tails) (void)hipEventDestroy(b->ev_tails);
    b->in_lease.release();
    batch_release(b, b->d_states);
    batch_release(b, b->d_pieces);
    batch_release(b, b->d_pack);
    batch_release(b, b->pin_res, true);
    delete b;
    return XLZ_OK;
}

} // namespace

// How xlz_decode_batch wants a (sub-)batch made.  want_slices > 1: a sliced batch (xlz_batch: slice_*) if every unit's model
// fits LDS and the units are ONE wave round (a launch per slice and round would pay every round's tail per slice);
// head_frac != 0: such a batch's first launch may start when that share (in 1/65536, + 4 KiB) of every unit's input is on
// the device -- the rest is uploaded by xlz_batch_run behind the first launch, and `streams` must stay valid until then.
struct BatchOpts {
    uint32_t want_slices = 1;
    uint32_t head_frac = 0;
    uint32_t call_units = 0;         // xlz_batch::call_units
    bool pooled = false;             // xlz_batch::pooled
    hipStream_t run_stream = nullptr; // xlz_batch::run_stream / queue (nullptr: the context's)
    uint32_t *queue = nullptr;
};
static int batch_create_ex(xlz_ctx *ctx, const xlz_stream_desc *streams, size_t n, xlz_batch **out, const BatchOpts &opts);

extern "C" int xlz_batch_create(xlz_ctx *ctx, const xlz_stream_desc *streams, size_t n, xlz_batch **out)
{
    return batch_create_ex(ctx, streams, n, out, BatchOpts{});
}

static int batch_create_ex(xlz_ctx *ctx, const xlz_stream_desc *streams, size_t n, xlz_batch **out, const BatchOpts &opts)
{
    const uint32_t want_slices = opts.want_slices, head_frac = opts.head_frac;
    if (!ctx || !out || (!streams && n)) return XLZ_ERR_BAD_ARG;
    *out = nullptr;
    for (size_t i = 0; i < n; i++) {
        if (!streams[i].in && streams[i].in_len) return XLZ_ERR_BAD_ARG;
        // flags: only the bits the header defines, and a slice is a slice of a raw LZMA2 stream (the field was
        // `reserved` before round 4 -- a caller's stale bytes must not silently change what a stream means)
        if ((streams[i].flags & ~XLZ_STREAM_F_LZMA2_SLICE) ||
            ((streams[i].flags & XLZ_STREAM_F_LZMA2_SLICE) && streams[i].format != XLZ_FMT_LZMA2_RAW))
            return XLZ_ERR_BAD_ARG;
        for (size_t k = 0; k < sizeof streams[i].reserved; k++)
            if (streams[i].reserved[k]) return XLZ_ERR_BAD_ARG;
    }
    // (no ctx->mu here: planning touches only the new batch, the pinned staging image has its own mutex -- a pipelined
    //  xlz_decode_batch creates sub-batch k+1 while sub-batch k is launched and collected)
    HIP_TRY(hipSetDevice(ctx->device));

    xlz_batch *b = new (std::nothrow) xlz_batch;
    if (!b) return XLZ_ERR_BAD_ARG;
    b->ctx = ctx;
    b->n = n;
    b->plans.resize(n);
    b->call_units = opts.call_units;
    b->pooled = opts.pooled;
    b->run_stream = opts.run_stream;
    b->queue = opts.queue;

    // ---- plan: parse headers, lay out the arenas -------------------------
    auto unit_src_off_p = std::make_shared<std::vector<size_t>>(); // where each unit's payload starts in its stream
    std::vector<size_t> &unit_src_off = *unit_src_off_p;
    size_t in_cursor = 0, out_cursor = 0;
    for (size_t i = 0; i < n; i++) {
        const xlz_stream_desc &s = streams[i];
        StreamPlan &pl = b->plans[i];
        Unit u;
        memset(&u, 0, sizeof u);
        bool has_unit = false;
        switch (s.format) {
        case XLZ_FMT_LZMA_ALONE: plan_lzma_alone(s, pl, u, has_unit); break;
        case XLZ_FMT_LZMA_RAW: plan_lzma_raw(s, pl, u, has_unit); break;
        case XLZ_FMT_LZMA2_RAW:
            u.kind = UNIT_LZMA2;
            u.dict_size = s.dict_size < kLzmaDicMin ? 8u * 1024 * 1024 : s.dict_size; // reader2.go:88-91
            u.unpack_size = kUnknownSize;
            pl.header_len = 0;
            pl.slice = (s.flags & XLZ_STREAM_F_LZMA2_SLICE) != 0;
            has_unit = true;
            break;
        default: pl.host_status = XLZ_ERR_BAD_ARG; break;
        }
        if (!has_unit) continue;

        if (u.kind == UNIT_LZMA2) {
            if (s.in_len > kMaxUnitBytes || s.out_cap > kMaxUnitBytes) {
                pl.host_status = XLZ_ERR_UNSUPPORTED; // (xlz_decode_batch decodes these as sessions)
                pl.oversize = !pl.slice; // ... but not a slice: a session knows nothing of the units in front of it
                continue;
            }
            std::vector<Lz2Unit> lu;
            uint32_t mx = 0, mpb = 0;
            scan_lzma2(s.in, s.in_len, lu, mx, &mpb);
            const bool big = mx > kMaxLcLpLds; // model too large for LDS: HBM-model launch
            if (big)
                b->max_lc_lp_big = std::max(b->max_lc_lp_big, mx);
            else
                b->max_lc_lp = std::max(b->max_lc_lp, mx), b->max_pb = std::max(b->max_pb, mpb);
            pl.lzma2 = true;
            pl.in_off = in_cursor;
            pl.in_len = (uint32_t)s.in_len;
            pl.dict_size = u.dict_size;
            pl.first_unit = (uint32_t)b->units.size();
            pl.n_units = (uint32_t)lu.size();
            pl.out_off = out_cursor;
            pl.out_cap = s.out_cap;
            for (size_t k = 0; k < lu.size(); k++) {
                Unit v = u;
                const bool last = k + 1 == lu.size();
                v.in_off = in_cursor + lu[k].in_start;
                v.in_len = (uint32_t)lu[k].in_len; // (s.in_len <= kMaxUnitBytes: checked above)
                v.out_off = out_cursor + std::min<uint64_t>(lu[k].out_start, s.out_cap);
                const uint64_t room = s.out_cap > lu[k].out_start ? s.out_cap - lu[k].out_start : 0;
                v.out_cap = (uint32_t)(last ? room : std::min<uint64_t>(room, lu[k].expect_out));
                v.expect_out = (uint32_t)lu[k].expect_out;
                v.stream = (uint32_t)i;
                v.flags = (last ? UNIT_F_LAST : 0u) | (lu[k].have_reader ? UNIT_F_HAVE_READER : 0u) |
                          ((k || pl.slice) ? UNIT_F_NOT_FIRST : 0u) | (big ? UNIT_F_BIG_MODEL : 0u);
                v.lc = (uint8_t)mx; // sizes the model storage; the real lc/lp/pb come from the chunk headers
                b->units.push_back(v);
                unit_src_off.push_back(lu[k].in_start);
            }
            b->algo_in += s.in_len;
            in_cursor += align_up(s.in_len + 16, kArenaAlign);
            out_cursor += align_up((size_t)s.out_cap + kOutTailPad, kArenaAlign);
            continue;
        }

        const size_t payload = s.in_len - pl.header_len;
        uint64_t cap = s.out_cap;
        if (u.unpack_size != kUnknownSize && u.unpack_size < cap)
            cap = u.unpack_size; // a defined size bounds the output (decompress.go:657-662)
        // 32-bit byte counters on the device: a defined size must fit them
        const bool size_too_big = u.unpack_size != kUnknownSize && u.unpack_size > kMaxUnitBytes;
        if (payload > kMaxUnitBytes || cap > kMaxUnitBytes || size_too_big) {
            pl.host_status = XLZ_ERR_UNSUPPORTED; // (xlz_decode_batch decodes these as sessions)
            pl.oversize = true;
            continue;
        }
        const bool big = (uint32_t)u.lc + u.lp > kMaxLcLpLds;
        u.in_off = in_cursor;
        u.in_len = (uint32_t)payload;
        u.out_off = out_cursor;
        u.out_cap = (uint32_t)cap;
        u.stream = (uint32_t)i;
        u.flags = UNIT_F_LAST | (big ? UNIT_F_BIG_MODEL : 0u);
        pl.first_unit = (uint32_t)b->units.size();
        pl.n_units = 1;
        pl.out_off = out_cursor;
        pl.out_cap = cap;
        b->units.push_back(u);
        unit_src_off.push_back(pl.header_len);
        if (big)
            b->max_lc_lp_big = std::max(b->max_lc_lp_big, (uint32_t)u.lc + u.lp);
        else
            b->max_lc_lp = std::max(b->max_lc_lp, (uint32_t)u.lc + u.lp), b->max_pb = std::max<uint32_t>(b->max_pb, u.pb);
        b->algo_in += payload;
        in_cursor += align_up(payload + 16, kArenaAlign);
        out_cursor += align_up((size_t)cap + kOutTailPad, kArenaAlign);
    }
    b->in_bytes = in_cursor + kArenaTailPad;
    b->out_bytes = out_cursor + kArenaTailPad;

    // work-queue order: most compressed bytes first (decode time tracks the number of
    // binary decisions, which tracks compressed size)
    b->order.resize(b->units.size());
    std::iota(b->order.begin(), b->order.end(), 0u);
    std::stable_sort(b->order.begin(), b->order.end(), [&](uint32_t a, uint32_t c) {
        const bool ba = b->units[a].flags & UNIT_F_BIG_MODEL, bc = b->units[c].flags & UNIT_F_BIG_MODEL;
        if (ba != bc) return !ba; // LDS-model units first
        return b->units[a].in_len > b->units[c].in_len;
    });
    for (uint32_t idx : b->order)
        if (!(b->units[idx].flags & UNIT_F_BIG_MODEL)) b->n_normal++;

    // every LDS-model unit's pb <= 2 (liblzma's and 7-Zip's default): the launch uses the compact model layout -- five LDS
    // granules instead of six for lc+lp = 3, 24 workgroups per CU instead of 21 (xlz_format.h: ModelLayout)
    b->compact = b->max_pb <= kCompactPosBits && !getenv("XLZ_NO_COMPACT");

    // ---- device memory + upload ----------------------------------------
    const size_t nu = b->units.size();
    auto fail = [&](int st) {
        batch_free(b);
        return st;
    };
    if (!batch_alloc(b, &b->d_in, b->in_bytes)) return fail(XLZ_ERR_DEVICE);
    if (!batch_alloc(b, &b->d_out, b->out_bytes)) return fail(XLZ_ERR_DEVICE);
    if (nu) {
        if (!batch_alloc(b, &b->d_units, nu * sizeof(Unit))) return fail(XLZ_ERR_DEVICE);
        if (!batch_alloc(b, &b->d_order, nu * sizeof(uint32_t))) return fail(XLZ_ERR_DEVICE);
        if (!batch_alloc(b, &b->d_results, nu * sizeof(UnitResult))) return fail(XLZ_ERR_DEVICE);
    }
    if (hipEventCreate(&b->ev0) != hipSuccess || hipEventCreate(&b->ev1) != hipSuccess)
        return fail(XLZ_ERR_DEVICE);
    if (b->n_normal < nu) { // some models live in HBM: one scratch slot per workgroup of that launch
        b->scratch_stride = num_probs(b->max_lc_lp_big) + num_matched_probs(b->max_lc_lp_big);
        if (!batch_alloc(b, &b->d_scratch, (size_t)big_model_grid(ctx->num_cus) * b->scratch_stride * sizeof(uint16_t)))
            return fail(XLZ_ERR_DEVICE);
    }
    if (b->n_normal) { // the matched-literal half of every resident model (xlz_format.h)
        b->mlit_stride = num_matched_probs(b->max_lc_lp);
        // one slot per workgroup of the largest grid any launch over these units can have -- the main launch, or a re-run of
        // some of them (run_units): never more workgroups than units (a call of eight streams does not need 6144 slots: 53 MB)
        const size_t slots = std::min<size_t>(b->n_normal, decode_grid(b->max_lc_lp, ctx->num_cus, ~0u, b->compact));
        if (!batch_alloc(b, &b->d_mlit, slots * b->mlit_stride * sizeof(uint16_t))) return fail(XLZ_ERR_DEVICE);
    }
    // (units whose model lives in HBM -- lc + lp > 8 -- run in their own launch behind the slices; their streams are fetched at
    //  the end like the streams of a re-run)
    if (want_slices > 1 && b->n_normal && b->n_normal <= decode_grid(b->max_lc_lp, ctx->num_cus, b->call_units ? b->call_units : b->n_normal, b->compact)) {
        // sliced batch: equal shares of every unit's output per launch; a state block per unit; per launch the table of
        // the pieces it finishes (unit order = arena order, packed back to back on 256-byte boundaries)
        // equal shares -- but the LAST share is cut in two (from four slices on): its download is the one nothing overlaps
        const uint32_t K0 = std::min<uint32_t>(want_slices, 63);
        for (uint32_t k = 1; k < K0; k++) b->slice_fracs.push_back((uint32_t)((uint64_t)kSliceOne * k / K0));
        if (K0 >= 4) b->slice_fracs.push_back((uint32_t)((uint64_t)kSliceOne * (2 * K0 - 1) / (2 * K0)));
        b->slice_fracs.push_back(kSliceOne);
        const uint32_t K = (uint32_t)b->slice_fracs.size();
        b->state_stride = align_up(state_bytes(b->max_lc_lp), kArenaAlign);
        if (!batch_alloc(b, &b->d_states, nu * b->state_stride)) return fail(XLZ_ERR_DEVICE);
        for (size_t k = 0; k < nu; k++) // (a unit of the HBM-model launch keeps state == 0: it is an ordinary unit there)
            if (!(b->units[k].flags & UNIT_F_BIG_MODEL)) b->units[k].state = (uint64_t)(uintptr_t)(b->d_states + k * b->state_stride);
        b->slice_pieces.resize(K);
        b->slice_pack_bytes.assign(K, 0);
        size_t total_pieces = 0;
        uint64_t max_pack = 0;
        for (uint32_t k = 0; k < K; k++) {
            uint64_t cursor = 0;
            for (size_t ui = 0; ui < nu; ui++) {
                const Unit &u = b->units[ui];
                if (u.flags & UNIT_F_BIG_MODEL) continue; // (not in the sliced launches)
                const uint32_t lo = k == 0 ? 0u : std::min(slice_bound(u.out_cap, b->slice_fracs[k - 1]), u.out_cap);
                const uint32_t hi = std::min(slice_bound(u.out_cap, b->slice_fracs[k]), u.out_cap);
                if (hi <= lo) continue;
                // (a piece sits in the packed image at its source's offset modulo 256: both ends of a 16-byte access are
                //  aligned or neither is -- the units of an LZMA2 stream begin at any byte)
                const uint64_t po = align_up(cursor, kArenaAlign) + ((u.out_off + lo) & (kArenaAlign - 1));
                b->slice_pieces[k].push_back(SlicePiece{u.out_off + lo, po, hi - lo, (uint32_t)ui});
                cursor = po + (hi - lo);
            }
            b->slice_pack_bytes[k] = cursor;
            max_pack = std::max(max_pack, cursor);
            total_pieces += b->slice_pieces[k].size();
        }
        b->slice_ev.assign(K, nullptr);
        for (uint32_t k = 0; k < K; k++)
            if (hipEventCreateWithFlags(&b->slice_ev[k], hipEventDisableTiming) != hipSuccess) return fail(XLZ_ERR_DEVICE);
        if (!batch_alloc(b, &b->d_pieces, std::max<size_t>(total_pieces, 1) * sizeof(SlicePiece)) ||
            !batch_alloc(b, &b->d_pack, (size_t)max_pack + kArenaAlign) ||
            !batch_alloc(b, &b->pin_res, (size_t)K * nu * sizeof(UnitResult), true))
            return fail(XLZ_ERR_DEVICE);
        size_t at = 0;
        for (uint32_t k = 0; k < K; k++) {
            const std::vector<SlicePiece> &pv = b->slice_pieces[k];
            if (!pv.empty() &&
                hipMemcpy(b->d_pieces + at, pv.data(), pv.size() * sizeof(SlicePiece), hipMemcpyHostToDevice) != hipSuccess)
                return fail(XLZ_ERR_DEVICE);
            at += pv.size();
        }
    }
    if (!b->slice_fracs.empty() && head_frac && head_frac < kSliceOne) {
        // heads first (xlz_batch: head_frac): pinned image and device staging hold heads | tails, each packed; a scatter
        // kernel puts a part into the input arena when its bytes have arrived; the tails' half runs behind the first launch
        HostPipe &hp = ctx->pipe;
        b->head_frac = head_frac;
        auto heads = std::make_shared<std::vector<SlicePiece>>(), tails = std::make_shared<std::vector<SlicePiece>>();
        uint64_t A = 0, B = 0;
        auto place = [](std::vector<SlicePiece> &v, uint64_t &cur, uint64_t src, uint32_t len, uint32_t ui) {
            if (!len) return;
            const uint64_t po = align_up(cur, kArenaAlign) + (src & (kArenaAlign - 1));
            v.push_back(SlicePiece{src, po, len, ui});
            cur = po + len;
        };
        for (size_t ui = 0; ui < nu; ui++) {
            const Unit &u = b->units[ui];
            // (a unit of the HBM-model launch, which runs behind all slices, has no head: all of its input is "tail")
            const uint32_t h = (u.flags & UNIT_F_BIG_MODEL) ? 0u : slice_head(u.in_len, head_frac, u.kind == UNIT_LZMA2);
            place(*heads, A, u.in_off, h, (uint32_t)ui);
            place(*tails, B, u.in_off + h, u.in_len - h, (uint32_t)ui);
        }
        A = align_up(A, kArenaAlign);
        B = align_up(B, kArenaAlign);
        if (!batch_alloc(b, &b->d_stage, A + B + kArenaAlign) ||
            !batch_alloc(b, &b->d_up_pieces, std::max<size_t>(heads->size() + tails->size(), 1) * sizeof(SlicePiece)) ||
            hipEventCreateWithFlags(&b->ev_heads, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&b->ev_tails, hipEventDisableTiming) != hipSuccess)
            return fail(XLZ_ERR_DEVICE);
        if ((!heads->empty() && hipMemcpy(b->d_up_pieces, heads->data(), heads->size() * sizeof(SlicePiece), hipMemcpyHostToDevice) != hipSuccess) ||
            (!tails->empty() && hipMemcpy(b->d_up_pieces + heads->size(), tails->data(), tails->size() * sizeof(SlicePiece),
                                          hipMemcpyHostToDevice) != hipSuccess))
            return fail(XLZ_ERR_DEVICE);
        b->in_lease.take(hp);
        if (hp.pin_in_cap < A + B) {
            if (hp.pin_in) (void)hipHostFree(hp.pin_in);
            hp.pin_in = nullptr;
            hp.pin_in_cap = 0;
            const size_t want = (size_t)(A + B) + (size_t)(A + B) / 4 + (1u << 20);
            if (hipHostMalloc(&hp.pin_in, want, hipHostMallocDefault) != hipSuccess) return fail(XLZ_ERR_DEVICE);
            hp.pin_in_cap = want;
        }
        if (!hp.up_stream && hipStreamCreateWithFlags(&hp.up_stream, hipStreamNonBlocking) != hipSuccess) return fail(XLZ_ERR_DEVICE);
        // one part of the upload: pack its pieces into the pinned image (several host threads, each a contiguous run of the
        // pieces, ~16 MiB to the device at a time while the packing goes on), then scatter the part into the arena
        auto upload_part = [b, ctx, streams, unit_src_off_p](const std::vector<SlicePiece> &pv, size_t table_at, uint64_t zone,
                                                             uint64_t zone_bytes, hipEvent_t done) -> int {
            HostPipe &hp = ctx->pipe;
            uint8_t *stage = hp.pin_in + zone;
            const unsigned nth = host_threads((size_t)zone_bytes);
            std::atomic<bool> copy_failed{false};
            const size_t np = pv.size();
            auto pack = [&](unsigned t) {
                (void)hipSetDevice(ctx->device);
                const size_t k0 = np * t / nth, k1 = np * (t + 1) / nth;
                if (k0 >= k1) return;
                uint64_t sent = pv[k0].pack_off;
                auto flush_to = [&](uint64_t end) {
                    if (end > sent && hipMemcpyAsync(b->d_stage + zone + sent, stage + sent, (size_t)(end - sent), hipMemcpyHostToDevice,
                                                     hp.up_stream) != hipSuccess)
                        copy_failed = true;
                    sent = end;
                };
                for (size_t k = k0; k < k1; k++) {
                    const SlicePiece &pc = pv[k];
                    const Unit &u = b->units[pc.unit];
                    memcpy(stage + pc.pack_off, streams[u.stream].in + (*unit_src_off_p)[pc.unit] + (pc.src_off - u.in_off), pc.len);
                    if (pc.pack_off + pc.len - sent >= (16u << 20)) flush_to(pc.pack_off + pc.len);
                }
                flush_to(pv[k1 - 1].pack_off + pv[k1 - 1].len);
            };
            run_threads(nth, pack);
            if (copy_failed) return XLZ_ERR_DEVICE;
            if (launch_gather(b->d_up_pieces + table_at, (uint32_t)np, b->d_in, b->d_stage + zone, zone_bytes, ctx->num_cus, hp.up_stream,
                              true) != 0)
                return XLZ_ERR_DEVICE;
            return hipEventRecord(done, hp.up_stream) == hipSuccess ? XLZ_OK : XLZ_ERR_DEVICE;
        };
        // (the arena's padding reads as zeros: the decoder's input window may run into it)
        if (hipMemsetAsync(b->d_in, 0, b->in_bytes, hp.up_stream) != hipSuccess) return fail(XLZ_ERR_DEVICE);
        const int st_heads = upload_part(*heads, 0, 0, A, b->ev_heads);
        if (st_heads != XLZ_OK) return fail(st_heads);
        const size_t n_heads = heads->size();
        b->upload_tails = [upload_part, tails, n_heads, A, B, b]() { return upload_part(*tails, n_heads, A, B, b->ev_tails); };
    } else {
        // pack the payloads into the context's pinned image (several host threads), one H2D copy
        HostPipe &hp = ctx->pipe;
        PinLease lease;
        lease.take(hp);
        if (hp.pin_in_cap < b->in_bytes) {
            if (hp.pin_in) (void)hipHostFree(hp.pin_in);
            hp.pin_in = nullptr;
            hp.pin_in_cap = 0;
            const size_t want = b->in_bytes + b->in_bytes / 4 + (1u << 20);
            if (hipHostMalloc(&hp.pin_in, want, hipHostMallocDefault) != hipSuccess) return fail(XLZ_ERR_DEVICE);
            hp.pin_in_cap = want;
        }
        uint8_t *stage = hp.pin_in;
        if (!hp.up_stream && hipStreamCreateWithFlags(&hp.up_stream, hipStreamNonBlocking) != hipSuccess) return fail(XLZ_ERR_DEVICE);
        const unsigned nth = host_threads(b->in_bytes);
        std::atomic<bool> copy_failed{false};
        auto pack = [&](unsigned t) {
            // units are laid out in arena order: thread t owns a contiguous run of them and the
            // padding bytes that follow each (zeroed: the decoder's input window may run into them).
            // Every ~16 MiB it has packed go to the device at once: the bus works while the packing goes on.
            (void)hipSetDevice(ctx->device);
            const size_t k0 = nu * t / nth, k1 = nu * (t + 1) / nth;
            if (k0 >= k1) return;
            uint64_t sent = b->units[k0].in_off;
            if (t == 0) sent = 0;
            auto flush_to = [&](uint64_t end) {
                if (end > sent && hipMemcpyAsync(b->d_in + sent, stage + sent, (size_t)(end - sent), hipMemcpyHostToDevice,
                                                 hp.up_stream) != hipSuccess)
                    copy_failed = true;
                sent = end;
            };
            for (size_t k = k0; k < k1; k++) {
                const Unit &u = b->units[k];
                memcpy(stage + u.in_off, streams[u.stream].in + unit_src_off[k], u.in_len);
                const uint64_t end = u.in_off + u.in_len;
                const uint64_t next = k + 1 < nu ? b->units[k + 1].in_off : b->in_bytes;
                if (next > end) memset(stage + end, 0, (size_t)(next - end));
                if (next - sent >= (16u << 20)) flush_to(next);
            }
            flush_to(k1 < nu ? b->units[k1].in_off : b->in_bytes);
        };
        if (nu && b->units[0].in_off) memset(stage, 0, (size_t)b->units[0].in_off);
        if (!nu) {
            memset(stage, 0, b->in_bytes);
            if (hipMemcpyAsync(b->d_in, stage, b->in_bytes, hipMemcpyHostToDevice, hp.up_stream) != hipSuccess) copy_failed = true;
        }
        run_threads(nth, pack);
        if (hipStreamSynchronize(hp.up_stream) != hipSuccess || copy_failed) return fail(XLZ_ERR_DEVICE);
    }
    if (nu) {
        if (hipMemcpy(b->d_units, b->units.data(), nu * sizeof(Unit), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(b->d_order, b->order.data(), nu * sizeof(uint32_t), hipMemcpyHostToDevice) != hipSuccess)
            return fail(XLZ_ERR_DEVICE);
    }
    *out = b;
    return XLZ_OK;
}


extern "C" int xlz_batch_run(xlz_batch *b)
{
    if (!b) return XLZ_ERR_BAD_ARG;
    xlz_ctx *ctx = b->ctx;
    std::lock_guard<std::mutex> lock(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    const hipStream_t rs = b->run_stream ? b->run_stream : ctx->stream;
    uint32_t *const rq = b->queue ? b->queue : ctx->queue;
    HIP_TRY(hipMemsetAsync(rq, 0, 256, rs));
    HIP_TRY(hipEventRecord(b->ev0, rs));
    const uint32_t nu = (uint32_t)b->units.size();
    LaunchParams p;
    memset(&p, 0, sizeof p); // (ready, epochs, ...: off unless set below)
    p.in_arena = b->d_in;
    p.out_arena = b->d_out;
    p.units = b->d_units;
    p.order = b->d_order;
    p.results = b->d_results;
    p.queue = rq;
    p.prio_tab = ctx->prio_tab;
    p.call_units = b->call_units;
    p.epochs = nullptr; // ordinary launch: copies that reach across a dictionary reset are only flagged
    if (b->n_normal) { // models in LDS
        p.n_units = b->n_normal;
        p.max_lc_lp = b->max_lc_lp;
        p.compact = b->compact ? 1u : 0u;
        p.scratch = nullptr;
        p.scratch_stride = 0;
        p.mlit = b->d_mlit;
        p.mlit_stride = b->mlit_stride;
        p.order_base = 0;
        if (b->slice_fracs.empty()) {
            if (launch_decode(p, ctx->num_cus, rs) != 0) return XLZ_ERR_DEVICE;
        } else { // a sequence of launches, each up to the next output bound of every unit (all queued at once)
            for (size_t k = 0; k < b->slice_fracs.size(); k++) {
                if (k) HIP_TRY(hipMemsetAsync(rq, 0, 256, rs));
                if (k == 0 && b->upload_tails) HIP_TRY(hipStreamWaitEvent(rs, b->ev_heads, 0)); // the heads are on their way
                if (k == 1 && b->upload_tails) { // the first launch is queued: now pack and upload the rest of the inputs
                    const int st = b->upload_tails();
                    b->upload_tails = nullptr;
                    if (st != XLZ_OK) return st;
                    HIP_TRY(hipStreamWaitEvent(rs, b->ev_tails, 0));
                }
                p.slice_frac = b->slice_fracs[k];
                p.slice_k = (uint32_t)k;
                p.head_frac = k == 0 ? b->head_frac : 0;
                if (launch_decode(p, ctx->num_cus, rs) != 0) return XLZ_ERR_DEVICE;
                HIP_TRY(hipEventRecord(b->slice_ev[k], rs));
            }
            p.slice_frac = 0;
            p.slice_k = 0;
            p.head_frac = 0;
            if (b->head_frac) { // the pinned image is free again when the tails have left it
                HIP_TRY(hipEventSynchronize(b->ev_tails));
                b->in_lease.release();
                b->head_frac = 0; // (a second run of the batch finds all input in the arena)
            }
        }
    }
    if (nu > b->n_normal) { // models in HBM (lc+lp > 8): the full layout
        p.compact = 0;
        HIP_TRY(hipMemsetAsync(rq, 0, 256, rs));
        p.n_units = nu - b->n_normal;
        p.max_lc_lp = b->max_lc_lp_big;
        p.scratch = b->d_scratch;
        p.scratch_stride = b->scratch_stride;
        p.mlit = nullptr;
        p.mlit_stride = 0;
        p.order_base = b->n_normal;
        if (launch_decode(p, ctx->num_cus, rs) != 0) return XLZ_ERR_DEVICE;
    }
    HIP_TRY(hipEventRecord(b->ev1, rs));
    b->ran = true;
    b->collected = false;
    return XLZ_OK;
}

extern "C" int xlz_batch_sync(xlz_batch *b)
{
    if (!b) return XLZ_ERR_BAD_ARG;
    HIP_TRY(hipSetDevice(b->ctx->device));
    HIP_TRY(hipStreamSynchronize(b->run_stream ? b->run_stream : b->ctx->stream));
    return XLZ_OK;
}

extern "C" int xlz_batch_last_kernel_ms(xlz_batch *b, float *ms)
{
    if (!b || !ms || !b->ran) return XLZ_ERR_BAD_ARG;
    HIP_TRY(hipSetDevice(b->ctx->device));
    HIP_TRY(hipEventSynchronize(b->ev1));
    HIP_TRY(hipEventElapsedTime(ms, b->ev0, b->ev1));
    return XLZ_OK;
}

namespace {

// Launch `units` (already laid out against the batch's arenas) and fetch their results.  These are
// always EXACT launches: every workgroup gets an epoch table, so copies that reach across an LZMA2
// dictionary reset read the bytes the reference's uncleared window holds (window.go:135-140).
// wide_lc_lp != 0: an HBM-model launch of its own with room for models up to that lc + lp (scratch allocated here)
int run_units(xlz_batch *b, const std::vector<Unit> &units, std::vector<UnitResult> &res, bool big, uint32_t wide_lc_lp = 0)
{
    xlz_ctx *ctx = b->ctx;
    const size_t n = units.size();
    res.resize(n);
    if (!n) return XLZ_OK;
    Unit *d_units = nullptr;
    uint32_t *d_order = nullptr;
    UnitResult *d_res = nullptr;
    Epoch *d_epochs = nullptr;
    std::vector<uint32_t> order(n);
    std::iota(order.begin(), order.end(), 0u);
    uint32_t grid = std::min<uint32_t>((uint32_t)n, big ? big_model_grid(ctx->num_cus)
                                                        : decode_grid(b->max_lc_lp, ctx->num_cus, (uint32_t)n, b->compact));
    if (wide_lc_lp) grid = std::min(grid, kWideGrid); // (the launch's grid follows p.n_units: see below)
    int st = XLZ_ERR_DEVICE;
    uint16_t *d_wide = nullptr;
    const uint32_t wide_stride = wide_lc_lp ? num_probs(wide_lc_lp) + num_matched_probs(wide_lc_lp) : 0;
    if (wide_lc_lp && hipMalloc(&d_wide, (size_t)grid * wide_stride * sizeof(uint16_t)) != hipSuccess) {
        (void)hipGetLastError();
        return XLZ_ERR_NO_MEMORY_INTERNAL;
    }
    if (hipMalloc(&d_units, n * sizeof(Unit)) == hipSuccess && hipMalloc(&d_order, n * sizeof(uint32_t)) == hipSuccess &&
        hipMalloc(&d_res, n * sizeof(UnitResult)) == hipSuccess &&
        hipMalloc(&d_epochs, (size_t)grid * kMaxEpochs * sizeof(Epoch)) == hipSuccess &&
        hipMemcpy(d_units, units.data(), n * sizeof(Unit), hipMemcpyHostToDevice) == hipSuccess &&
        hipMemcpy(d_order, order.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice) == hipSuccess &&
        hipMemsetAsync(ctx->queue, 0, 256, ctx->stream) == hipSuccess) {
        LaunchParams p;
        memset(&p, 0, sizeof p); // (epochs, order_base, ...: off unless set below)
        p.in_arena = b->d_in;
        p.out_arena = b->d_out;
        p.units = d_units;
        p.order = d_order;
        p.results = d_res;
        p.queue = ctx->queue;
        p.n_units = (uint32_t)n;
        p.max_lc_lp = wide_lc_lp ? wide_lc_lp : big ? b->max_lc_lp_big : b->max_lc_lp;
        p.scratch = wide_lc_lp ? d_wide : big ? b->d_scratch : nullptr;
        p.scratch_stride = wide_lc_lp ? wide_stride : big ? b->scratch_stride : 0;
        p.mlit = big ? nullptr : b->d_mlit;
        p.mlit_stride = big ? 0 : b->mlit_stride;
        p.compact = (!big && !wide_lc_lp && b->compact) ? 1u : 0u;
        p.order_base = 0;
        p.epochs = d_epochs;
        p.prio_tab = ctx->prio_tab;
        if (launch_decode(p, ctx->num_cus, ctx->stream, wide_lc_lp ? grid : 0) == 0 && hipStreamSynchronize(ctx->stream) == hipSuccess &&
            hipMemcpy(res.data(), d_res, n * sizeof(UnitResult), hipMemcpyDeviceToHost) == hipSuccess)
            st = XLZ_OK;
    }
    if (d_units) (void)hipFree(d_units);
    if (d_order) (void)hipFree(d_order);
    if (d_res) (void)hipFree(d_res);
    if (d_epochs) (void)hipFree(d_epochs);
    if (d_wide) (void)hipFree(d_wide);
    return st;
}

void fold_streams(xlz_batch *b, size_t s0, size_t s1, std::vector<size_t> &redo);
int finish_collect(xlz_batch *b, const std::vector<size_t> &redo);

// Sync, fetch unit results, fold them into per-stream results.  An LZMA2 stream whose
// units did not produce / consume exactly what its chunk headers announced is decoded
// again as ONE unit: that pass follows the reference's framing byte for byte.
int collect(xlz_batch *b)
{
    if (b->collected) return XLZ_OK;
    xlz_ctx *ctx = b->ctx;
    std::lock_guard<std::mutex> lock(ctx->mu);
    HIP_TRY(hipSetDevice(ctx->device));
    HIP_TRY(hipEventSynchronize(b->ev1)); // this batch's launches (a later batch may already be queued behind them)
    std::vector<UnitResult> &ur = b->unit_results;
    ur.resize(b->units.size());
    if (!ur.empty())
        HIP_TRY(hipMemcpy(ur.data(), b->d_results, ur.size() * sizeof(UnitResult), hipMemcpyDeviceToHost));
    b->final_results.assign(b->n, xlz_result{});
    b->rewritten.clear();
    b->wants_wide.assign(b->n, 0);
    std::vector<size_t> redo;
    fold_streams(b, 0, b->n, redo);
    return finish_collect(b, redo);
}

// unit results -> per-stream results for streams [s0, s1); streams that need the exact re-run go to `redo`
void fold_streams(xlz_batch *b, size_t s0, size_t s1, std::vector<size_t> &redo)
{
    const std::vector<UnitResult> &ur = b->unit_results;
    for (size_t i = s0; i < s1; i++) {
        const StreamPlan &pl = b->plans[i];
        xlz_result &r = b->final_results[i];
        if (pl.host_status != 1) { // settled while parsing
            r.status = pl.host_status;
            r.in_consumed = pl.host_in_consumed;
            continue;
        }
        if (!pl.lzma2) {
            const UnitResult &u = ur[pl.first_unit];
            r.status = u.status;
            r.out_len = u.out_len;
            r.in_consumed = (uint64_t)pl.header_len + u.in_consumed;
            continue;
        }
        uint64_t out = 0, in_base = 0;
        bool settled = false;
        // a unit read window bytes of an earlier dictionary epoch (AUX_STALE): the ordinary launch
        // returns zeros there, the reference returns what its uncleared buffer holds -- decode the
        // stream again as one unit of an exact launch
        // (a SLICE of a stream cannot be settled here: the earlier epochs are not in this call, and a re-run of the
        //  slice as one unit would read zeros where the reference reads them -- the caller decodes the whole stream)
        auto again = [&] {
            if (pl.slice) {
                r.status = XLZ_ERR_UNSUPPORTED;
                r.out_len = 0;
                r.in_consumed = 0;
            } else {
                redo.push_back(i);
            }
        };
        for (uint32_t k = 0; k < pl.n_units && !settled; k++)
            if (ur[pl.first_unit + k].aux & AUX_STALE) {
                again();
                settled = true;
            }
        for (uint32_t k = 0; k < pl.n_units && !settled; k++) {
            const Unit &un = b->units[pl.first_unit + k];
            const UnitResult &u = ur[pl.first_unit + k];
            const bool last = k + 1 == pl.n_units;
            if (u.status == ST_ERR_UNSUPPORTED && (u.aux & AUX_GROW)) b->wants_wide[i] = 1;
            if (last) {
                r.status = u.status;
                r.out_len = out + u.out_len;
                r.in_consumed = in_base + u.in_consumed;
                settled = true;
            } else if (u.status == ST_OK && !(u.aux & AUX_END_MARK) && u.out_len == un.expect_out && u.in_consumed == un.in_len) {
                out += u.out_len;
                in_base += u.in_consumed;
            } else if (u.status < 0 && u.status != ST_ERR_OUT_CAP) {
                // everything before this unit matched its headers, so the unit started from the
                // reference's exact state and fails where the reference fails
                r.status = u.status;
                r.out_len = out + u.out_len;
                r.in_consumed = in_base + u.in_consumed;
                settled = true;
            } else {
                again();
                settled = true;
            }
        }
    }
}

// the exact single-unit re-run of the streams in `redo` (malformed LZMA2 only), then the byte sums
int finish_collect(xlz_batch *b, const std::vector<size_t> &redo)
{
    for (int pass = 0; pass < 2 && !redo.empty(); pass++) {
        const bool big = pass == 1;
        std::vector<Unit> units;
        std::vector<size_t> idx;
        for (size_t i : redo) {
            if (((b->units[b->plans[i].first_unit].flags & UNIT_F_BIG_MODEL) != 0) != big) continue;
            idx.push_back(i);
            const StreamPlan &pl = b->plans[i];
            Unit u;
            memset(&u, 0, sizeof u);
            u.kind = UNIT_LZMA2;
            u.in_off = pl.in_off;
            u.in_len = pl.in_len;
            u.out_off = pl.out_off;
            u.out_cap = (uint32_t)pl.out_cap;
            u.dict_size = pl.dict_size;
            u.unpack_size = kUnknownSize;
            u.stream = (uint32_t)i;
            u.lc = b->units[pl.first_unit].lc;
            u.flags = UNIT_F_LAST | (big ? UNIT_F_BIG_MODEL : 0u);
            units.push_back(u);
        }
        if (units.empty()) continue;
        std::vector<UnitResult> res;
        int st = run_units(b, units, res, big);
        if (st != XLZ_OK) return st;
        for (size_t k = 0; k < idx.size(); k++) {
            xlz_result &r = b->final_results[idx[k]];
            r.status = res[k].status;
            r.out_len = res[k].out_len;
            r.in_consumed = res[k].in_consumed;
            b->rewritten.push_back(idx[k]);
            b->wants_wide[idx[k]] = (res[k].status == ST_ERR_UNSUPPORTED && (res[k].aux & AUX_GROW)) ? 1 : 0;
        }
    }
    // The model's storage (LDS, or a workgroup's HBM slot) is sized by the host's scan of the chunk HEADERS.  A malformed stream
    // whose real decode leaves its headers can walk into bytes that read as a chunk with larger properties than any the scan
    // saw: the wave stops in front of that chunk with ST_ERR_UNSUPPORTED -- the reference (reader2.go:155-165) simply renews
    // its model with whatever lc <= 8, lp <= 4 the byte says.  Such a stream is decoded once more, as ONE unit of an HBM-model
    // launch with room for the largest model the reference accepts (found by tools/fuzz_gpu.py in round 4, seed 5501:
    // tests/golden/fuzz_5501_18774.lzma2; sessions have had AUX_GROW for this since round 3, slices of a stream leave it
    // to their caller).
    {
        std::vector<Unit> units;
        std::vector<size_t> idx;
        for (size_t i = 0; i < b->n; i++) {
            const StreamPlan &pl = b->plans[i];
            // (only streams that stopped in front of LARGER PROPERTIES: a re-run cannot help a full epoch table -- ADVICE r4)
            if (pl.host_status != 1 || !pl.lzma2 || pl.slice || b->final_results[i].status != XLZ_ERR_UNSUPPORTED || !b->wants_wide[i]) continue;
            idx.push_back(i);
            Unit u;
            memset(&u, 0, sizeof u);
            u.kind = UNIT_LZMA2;
            u.in_off = pl.in_off;
            u.in_len = pl.in_len;
            u.out_off = pl.out_off;
            u.out_cap = (uint32_t)pl.out_cap;
            u.dict_size = pl.dict_size;
            u.unpack_size = kUnknownSize;
            u.stream = (uint32_t)i;
            u.lc = (uint8_t)kMaxLcLp;
            u.flags = UNIT_F_LAST | UNIT_F_BIG_MODEL;
            units.push_back(u);
        }
        if (!units.empty()) {
            std::vector<UnitResult> res;
            int st = run_units(b, units, res, true, kMaxLcLp);
            // no memory for the widest models (6 MiB per workgroup): these streams stay at XLZ_ERR_UNSUPPORTED, the call and
            // the other streams' results stand -- one crafted stream must not fail a batch
            if (st == XLZ_ERR_NO_MEMORY_INTERNAL) idx.clear();
            else if (st != XLZ_OK) return st;
            for (size_t k = 0; k < idx.size(); k++) {
                xlz_result &r = b->final_results[idx[k]];
                r.status = res[k].status;
                r.out_len = res[k].out_len;
                r.in_consumed = res[k].in_consumed;
                b->rewritten.push_back(idx[k]);
            }
        }
    }
    std::sort(b->rewritten.begin(), b->rewritten.end());
    b->rewritten.erase(std::unique(b->rewritten.begin(), b->rewritten.end()), b->rewritten.end());
    b->sum_in = b->sum_out = 0;
    for (size_t i = 0; i < b->n; i++) {
        const StreamPlan &pl = b->plans[i];
        if (pl.host_status != 1) continue;
        b->sum_out += b->final_results[i].out_len;
        b->sum_in += b->final_results[i].in_consumed - pl.header_len;
    }
    b->collected = true;
    return XLZ_OK;
}

} // namespace

extern "C" int xlz_batch_results(xlz_batch *b, xlz_result *results)
{
    if (!b || (!results && b->n) || !b->ran) return XLZ_ERR_BAD_ARG;
    int st = collect(b);
    if (st != XLZ_OK) return st;
    for (size_t i = 0; i < b->n; i++) results[i] = b->final_results[i];
    return XLZ_OK;
}

extern "C" int xlz_batch_stats(xlz_batch *b, uint64_t *in_bytes, uint64_t *out_bytes, uint64_t *units)
{
    if (!b || !b->ran) return XLZ_ERR_BAD_ARG;
    int st = collect(b);
    if (st != XLZ_OK) return st;
    if (in_bytes) *in_bytes = b->sum_in;
    if (out_bytes) *out_bytes = b->sum_out;
    if (units) *units = b->units.size();
    return XLZ_OK;
}

extern "C" int xlz_batch_unit_trace(xlz_batch *b, uint32_t *t_start, uint32_t *t_end, uint32_t *in_len, size_t cap,
                                    size_t *n_units)
{
    if (!b || !b->ran || !n_units) return XLZ_ERR_BAD_ARG;
    int st = collect(b);
    if (st != XLZ_OK) return st;
    *n_units = b->unit_results.size();
    uint32_t t0 = 0;
    bool first = true;
    for (const UnitResult &u : b->unit_results) // the launch's first start is the origin
        if (first || (int32_t)(u.t_start - t0) < 0) {
            t0 = u.t_start;
            first = false;
        }
    for (size_t k = 0; k < b->unit_results.size() && k < cap; k++) {
        if (t_start) t_start[k] = b->unit_results[k].t_start - t0;
        if (t_end) t_end[k] = b->unit_results[k].t_end - t0;
        if (in_len) in_len[k] = b->units[k].in_len;
    }
    return XLZ_OK;
}

extern "C" int xlz_batch_launch_info(xlz_batch *b, uint32_t *workgroups, uint32_t *lds_bytes)
{
    if (!b) return XLZ_ERR_BAD_ARG;
    const uint32_t grid = std::min<uint32_t>(b->n_normal, decode_grid(b->max_lc_lp, b->ctx->num_cus, b->call_units ? b->call_units : b->n_normal, b->compact));
    if (workgroups) *workgroups = grid;
    if (lds_bytes) *lds_bytes = decode_lds_bytes(b->max_lc_lp, b->compact);
    return XLZ_OK;
}

extern "C" const char *xlz_batch_kernel_name(xlz_batch *b)
{
    if (!b) return "";
    if (!b->n_normal) return "xlz::xlz_decode_kernel_hbm_model";
    if (decode_branchy(b->max_lc_lp, b->ctx->num_cus, b->call_units ? b->call_units : b->n_normal, b->compact)) return "xlz::xlz_decode_kernel_pb2_br";
    return b->compact ? "xlz::xlz_decode_kernel_pb2" : "xlz::xlz_decode_kernel";
}

extern "C" int xlz_batch_device_output(xlz_batch *b, size_t i, void **dptr, size_t *cap)
{
    if (!b || i >= b->n || !dptr) return XLZ_ERR_BAD_ARG;
    const StreamPlan &pl = b->plans[i];
    *dptr = pl.host_status == 1 ? b->d_out + pl.out_off : nullptr;
    if (cap) *cap = pl.host_status == 1 ? (size_t)pl.out_cap : 0;
    return XLZ_OK;
}

extern "C" int xlz_batch_download(xlz_batch *b, size_t i, uint8_t *dst, size_t len)
{
    if (!b || i >= b->n || (!dst && len)) return XLZ_ERR_BAD_ARG;
    const StreamPlan &pl = b->plans[i];
    if (pl.host_status != 1 || len == 0) return XLZ_OK;
    if (len > pl.out_cap) len = (size_t)pl.out_cap;
    HIP_TRY(hipSetDevice(b->ctx->device));
    HIP_TRY(hipStreamSynchronize(b->run_stream ? b->run_stream : b->ctx->stream));
    HIP_TRY(hipMemcpy(dst, b->d_out + pl.out_off, len, hipMemcpyDeviceToHost));
    return XLZ_OK;
}

extern "C" void xlz_batch_destroy(xlz_batch *b) { batch_free(b); }

namespace {

// All decoded streams of a batch to the callers' buffers: the output arena is cut into chunks of
// at most HostPipe::kRingBytes (whole stream regions where they fit, pieces of a region where one
// does not), each chunk is copied D2H into a slot of the pinned ring on the copy stream, and host
// threads scatter the chunks that have arrived while the next ones are in flight.
int download_all(xlz_batch *b, const xlz_stream_desc *streams, const xlz_result *results, const std::vector<size_t> *only = nullptr)
{
    xlz_ctx *ctx = b->ctx;
    HostPipe &hp = ctx->pipe;
    struct Piece {
        uint64_t dev_off;
        uint8_t *dst;
        size_t len;
    };
    struct Chunk {
        uint64_t dev_off = 0;
        size_t len = 0;
        std::vector<Piece> pieces;
    };
    std::vector<Chunk> chunks;
    const size_t n_take = only ? only->size() : b->n; // (`only`: ascending stream indices)
    for (size_t t = 0; t < n_take; t++) {
        const size_t i = only ? (*only)[t] : t;
        const StreamPlan &pl = b->plans[i];
        size_t len = (size_t)results[i].out_len;
        if (pl.host_status != 1 || len == 0) continue;
        if (len > pl.out_cap) len = (size_t)pl.out_cap;
        uint64_t off = pl.out_off;
        uint8_t *dst = streams[i].out;
        while (len) {
            if (chunks.empty() || off + 1 > chunks.back().dev_off + HostPipe::kRingBytes || off < chunks.back().dev_off) {
                chunks.emplace_back();
                chunks.back().dev_off = off;
            }
            Chunk &c = chunks.back();
            const size_t room = (size_t)(c.dev_off + HostPipe::kRingBytes - off);
            const size_t take = len < room ? len : room;
            c.pieces.push_back({off, dst, take});
            c.len = (size_t)(off + take - c.dev_off);
            off += take;
            dst += take;
            len -= take;
        }
    }
    if (chunks.empty()) return XLZ_OK;

    std::lock_guard<std::mutex> pl(hp.mu_out);
    HIP_TRY(hipSetDevice(ctx->device));
    if (!hp.copy_stream) HIP_TRY(hipStreamCreateWithFlags(&hp.copy_stream, hipStreamNonBlocking));
    for (int r = 0; r < HostPipe::kRing; r++) {
        if (!hp.ring[r]) HIP_TRY(hipHostMalloc(&hp.ring[r], HostPipe::kRingBytes, hipHostMallocDefault));
        if (!hp.ring_ev[r]) HIP_TRY(hipEventCreateWithFlags(&hp.ring_ev[r], hipEventDisableTiming));
    }
    // the decode of THIS batch has finished (its results were read).  Not hipStreamSynchronize(ctx->stream): in a
    // pipelined call the next sub-batch is decoding on that stream right now, and waiting for it is what made round 2's
    // overlap experiments look as if "a copy does not start under a running launch" (it does: tools/overlap_probe.py)
    HIP_TRY(hipEventSynchronize(b->ev1));

    // scatter workers: chunk k is claimed by one worker, which waits for its copy event
    const unsigned nworkers = std::min<unsigned>(host_threads(b->out_bytes), HostPipe::kRing);
    std::mutex mu;
    std::condition_variable cv;
    size_t issued = 0, next_claim = 0, n_done = 0; // chunks whose D2H was issued / claimed / scattered
    std::vector<char> done(chunks.size(), 0);
    bool failed = false;
    auto worker = [&] {
        (void)hipSetDevice(ctx->device);
        for (;;) {
            size_t k;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return next_claim < issued || next_claim >= chunks.size() || failed; });
                if (failed || next_claim >= chunks.size()) return;
                k = next_claim++;
            }
            const int slot = (int)(k % HostPipe::kRing);
            const bool ok = hipEventSynchronize(hp.ring_ev[slot]) == hipSuccess;
            if (ok)
                for (const Piece &pc : chunks[k].pieces)
                    memcpy(pc.dst, hp.ring[slot] + (pc.dev_off - chunks[k].dev_off), pc.len);
            {
                std::lock_guard<std::mutex> lk(mu);
                done[k] = 1;
                n_done++;
                if (!ok) failed = true;
            }
            cv.notify_all();
        }
    };
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nworkers; t++) th.emplace_back(worker);
    int st = XLZ_OK;
    for (size_t k = 0; k < chunks.size(); k++) {
        const int slot = (int)(k % HostPipe::kRing);
        if (k >= (size_t)HostPipe::kRing) { // the slot's previous chunk must have been scattered
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return done[k - HostPipe::kRing] || failed; });
            if (failed) break;
        }
        if (hipMemcpyAsync(hp.ring[slot], b->d_out + chunks[k].dev_off, chunks[k].len, hipMemcpyDeviceToHost,
                           hp.copy_stream) != hipSuccess ||
            hipEventRecord(hp.ring_ev[slot], hp.copy_stream) != hipSuccess) {
            std::lock_guard<std::mutex> lk(mu);
            failed = true;
            break;
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            issued = k + 1;
        }
        cv.notify_all();
    }
    {
        std::unique_lock<std::mutex> lk(mu);
        if (failed) {
            st = XLZ_ERR_DEVICE;
        } else {
            cv.wait(lk, [&] { return n_done == chunks.size() || failed; });
            if (failed) st = XLZ_ERR_DEVICE;
        }
        next_claim = std::max(next_claim, chunks.size()); // release idle workers
        if (st != XLZ_OK) failed = true;
    }
    cv.notify_all();
    for (auto &x : th) x.join();
    (void)hipStreamSynchronize(hp.copy_stream);
    return st;
}

// The download of a SLICED batch (xlz_batch: slice_*), running while the launches decode.  Behind launch k (slice_ev[k])
// the copy stream fetches the unit results, packs the pieces the launch finished into one image (xlz_gather_kernel, next
// to the running launch k + 1) and brings the image to the pinned ring in chunks; host threads scatter the chunks into the
// callers' buffers, every piece clamped to what its unit had produced by then (a unit that ended early, or in an error).
// This is the reference's Read pump in batch form: window.ReadPending drains what is there while the decoder keeps its
// state (reader1.go:223-254, window.go:97-133).  What a later re-run writes again (collect(): malformed LZMA2 streams) is
// fetched once more by the caller (xlz_batch::rewritten), and so are the streams in `gaps`: a unit of theirs fell short of
// a launch's bound (its head ran out in the first launch, xlz_batch: head_frac) and produced the bytes up to it only
// later, when that bound's pieces had gone out.  per_slice (optional): slot occupancy of every launch.
int download_sliced(xlz_batch *b, const xlz_stream_desc *streams, std::vector<double> *per_slice, std::vector<size_t> &gaps)
{
    xlz_ctx *ctx = b->ctx;
    HostPipe &hp = ctx->pipe;
    const size_t K = b->slice_fracs.size(), nu = b->units.size();
    struct Chunk {
        uint32_t k;
        uint64_t off;
        size_t len;
    };
    std::vector<Chunk> chunks;
    for (size_t k = 0; k < K; k++)
        for (uint64_t off = 0; off < b->slice_pack_bytes[k]; off += HostPipe::kRingBytes)
            chunks.push_back({(uint32_t)k, off, (size_t)std::min<uint64_t>(HostPipe::kRingBytes, b->slice_pack_bytes[k] - off)});

    std::lock_guard<std::mutex> pl(hp.mu_out);
    HIP_TRY(hipSetDevice(ctx->device));
    if (!hp.copy_stream) HIP_TRY(hipStreamCreateWithFlags(&hp.copy_stream, hipStreamNonBlocking));
    for (int r = 0; r < HostPipe::kRing; r++) {
        if (!hp.ring[r]) HIP_TRY(hipHostMalloc(&hp.ring[r], HostPipe::kRingBytes, hipHostMallocDefault));
        if (!hp.ring_ev[r]) HIP_TRY(hipEventCreateWithFlags(&hp.ring_ev[r], hipEventDisableTiming));
    }
    auto scatter = [&](const Chunk &c, const uint8_t *ring) {
        const std::vector<SlicePiece> &pv = b->slice_pieces[c.k];
        const UnitResult *res = b->pin_res + (size_t)c.k * nu;
        size_t lo = 0, hi = pv.size(); // the piece the chunk starts in
        while (hi - lo > 1) {
            const size_t mid = (lo + hi) / 2;
            if (pv[mid].pack_off <= c.off)
                lo = mid;
            else
                hi = mid;
        }
        for (size_t i = lo; i < pv.size() && pv[i].pack_off < c.off + c.len; i++) {
            const SlicePiece &pc = pv[i];
            const Unit &u = b->units[pc.unit];
            const uint64_t lo_in_unit = pc.src_off - u.out_off;
            const uint64_t got = res[pc.unit].out_len; // the unit's output behind this launch
            const uint64_t valid = got > lo_in_unit ? std::min<uint64_t>(got - lo_in_unit, pc.len) : 0;
            const uint64_t a = std::max<uint64_t>(pc.pack_off, c.off), e = std::min<uint64_t>(pc.pack_off + valid, c.off + c.len);
            if (e <= a) continue;
            uint8_t *dst = streams[u.stream].out + (u.out_off - b->plans[u.stream].out_off) + lo_in_unit + (a - pc.pack_off);
            memcpy(dst, ring + (a - c.off), (size_t)(e - a));
        }
    };
    const unsigned nworkers = std::min<unsigned>(host_threads(b->out_bytes), HostPipe::kRing);
    std::mutex mu;
    std::condition_variable cv;
    size_t issued = 0, next_claim = 0, n_done = 0;
    std::vector<char> done(chunks.size(), 0);
    bool failed = false;
    auto worker = [&] {
        (void)hipSetDevice(ctx->device);
        for (;;) {
            size_t j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return next_claim < issued || next_claim >= chunks.size() || failed; });
                if (failed || next_claim >= chunks.size()) return;
                j = next_claim++;
            }
            const int slot = (int)(j % HostPipe::kRing);
            const bool ok = hipEventSynchronize(hp.ring_ev[slot]) == hipSuccess;
            if (ok) scatter(chunks[j], hp.ring[slot]);
            {
                std::lock_guard<std::mutex> lk(mu);
                done[j] = 1;
                n_done++;
                if (!ok) failed = true;
            }
            cv.notify_all();
        }
    };
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nworkers; t++) th.emplace_back(worker);
    int st = XLZ_OK;
    size_t piece_base = 0;
    uint32_t slice_begun = ~0u;
    for (size_t j = 0; j <= chunks.size() && st == XLZ_OK; j++) {
        // the launches behind the last chunk's one (and launches that finish no byte at all) still deliver their results
        const uint32_t upto = j < chunks.size() ? chunks[j].k : (uint32_t)(K - 1);
        while (slice_begun == ~0u || slice_begun < upto) {
            const uint32_t k = slice_begun + 1;
            if (hipStreamWaitEvent(hp.copy_stream, b->slice_ev[k], 0) != hipSuccess ||
                hipMemcpyAsync(b->pin_res + (size_t)k * nu, b->d_results, nu * sizeof(UnitResult), hipMemcpyDeviceToHost,
                               hp.copy_stream) != hipSuccess ||
                launch_gather(b->d_pieces + piece_base, (uint32_t)b->slice_pieces[k].size(), b->d_out, b->d_pack,
                              b->slice_pack_bytes[k], ctx->num_cus, hp.copy_stream) != 0)
                st = XLZ_ERR_DEVICE;
            piece_base += b->slice_pieces[k].size();
            slice_begun = k;
            if (st != XLZ_OK) break;
        }
        if (j == chunks.size() || st != XLZ_OK) break;
        const int slot = (int)(j % HostPipe::kRing);
        if (j >= (size_t)HostPipe::kRing) { // the slot's previous chunk must have been scattered
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return done[j - HostPipe::kRing] || failed; });
            if (failed) break;
        }
        if (hipMemcpyAsync(hp.ring[slot], b->d_pack + chunks[j].off, chunks[j].len, hipMemcpyDeviceToHost, hp.copy_stream) != hipSuccess ||
            hipEventRecord(hp.ring_ev[slot], hp.copy_stream) != hipSuccess) {
            st = XLZ_ERR_DEVICE;
            break;
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            issued = j + 1;
        }
        cv.notify_all();
    }
    {
        std::unique_lock<std::mutex> lk(mu);
        if (st != XLZ_OK) failed = true;
        if (!failed) cv.wait(lk, [&] { return n_done == chunks.size() || failed; });
        if (failed) st = XLZ_ERR_DEVICE;
        next_claim = std::max(next_claim, chunks.size()); // release idle workers
    }
    cv.notify_all();
    for (auto &x : th) x.join();
    if (hipStreamSynchronize(hp.copy_stream) != hipSuccess) st = XLZ_ERR_DEVICE;
    gaps.clear();
    for (size_t ui = 0; ui < nu && st == XLZ_OK && K > 1; ui++) {
        const Unit &u = b->units[ui];
        if (u.flags & UNIT_F_BIG_MODEL) { // decoded by the HBM-model launch behind the slices: nothing of it has gone out
            if (gaps.empty() || gaps.back() != u.stream) gaps.push_back(u.stream);
            continue;
        }
        const uint64_t fin = b->pin_res[(K - 1) * nu + ui].out_len;
        for (size_t k = 0; k + 1 < K; k++) {
            const uint64_t hi = std::min(slice_bound(u.out_cap, b->slice_fracs[k]), u.out_cap);
            if (b->pin_res[k * nu + ui].out_len < std::min<uint64_t>(hi, fin)) {
                if (gaps.empty() || gaps.back() != u.stream) gaps.push_back(u.stream); // (units are in stream order)
                break;
            }
        }
    }
    if (st == XLZ_OK && per_slice) { // slot occupancy of every launch, from the stamps of the units it ran
        per_slice->assign(K, 0.0);
        uint32_t slots = 0;
        (void)xlz_batch_launch_info(b, &slots, nullptr);
        for (size_t k = 0; k < K && slots; k++) {
            const UnitResult *res = b->pin_res + k * nu, *prev = k ? res - nu : nullptr;
            uint32_t t0 = 0, span = 0;
            uint64_t busy = 0;
            bool any = false;
            for (size_t ui = 0; ui < nu; ui++) {
                if (b->units[ui].flags & UNIT_F_BIG_MODEL) continue; // (no result yet)
                if (prev && prev[ui].t_end == res[ui].t_end && prev[ui].t_start == res[ui].t_start) continue; // did not run
                if (!any || (int32_t)(res[ui].t_start - t0) < 0) t0 = res[ui].t_start;
                any = true;
            }
            for (size_t ui = 0; ui < nu && any; ui++) {
                if (b->units[ui].flags & UNIT_F_BIG_MODEL) continue;
                if (prev && prev[ui].t_end == res[ui].t_end && prev[ui].t_start == res[ui].t_start) continue;
                busy += (uint32_t)(res[ui].t_end - res[ui].t_start);
                span = std::max(span, (uint32_t)(res[ui].t_end - t0));
            }
            if (span) (*per_slice)[k] = (double)busy / ((double)slots * span);
        }
    }
    return st;
}

} // namespace

static int decode_oversize(xlz_ctx *ctx, const xlz_stream_desc *streams, xlz_result *results, const std::vector<size_t> &idx);

// Wave-slot bookkeeping of a call from the units' s_memrealtime stamps (100 MHz, the same clock for every launch of the
// device): busy wave time, first start and last end over ALL sub-batches of the call -- their launches overlap on two
// streams, so a sub-batch's own span says little; the call's occupancy = busy / (wave slots x (last end - first start)).
struct SlotClock {
    bool any = false;
    uint32_t ref = 0;        // the first stamp seen: everything else relative to it (the 32-bit clock wraps every 43 s)
    int64_t first = 0, last = 0;
    uint64_t busy = 0;
    void add(const std::vector<UnitResult> &ur)
    {
        for (const UnitResult &u : ur) {
            if (!any) ref = u.t_start, first = 0, last = 0, any = true;
            const int64_t a = (int32_t)(u.t_start - ref), e = (int32_t)(u.t_end - ref);
            first = std::min(first, a);
            last = std::max(last, e);
            busy += (uint32_t)(u.t_end - u.t_start);
        }
    }
    double span_ms() const { return any ? (double)(last - first) / 1e5 : 0.0; }
    double occupancy(uint32_t slots) const { return any && slots && last > first ? (double)busy / ((double)slots * (double)(last - first)) : 0.0; }
};

extern "C" int xlz_ctx_set_slicing(xlz_ctx *ctx, uint64_t min_call_bytes, uint64_t slice_bytes, uint32_t max_slices)
{
    if (!ctx) return XLZ_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(ctx->mu);
    ctx->sliced_call_bytes = min_call_bytes ? min_call_bytes : kSlicedCallBytes;
    ctx->slice_bytes = slice_bytes ? slice_bytes : kSliceBytes;
    ctx->max_slices = max_slices ? std::min<uint32_t>(max_slices, 63) : (uint32_t)kMaxSlices;
    return XLZ_OK;
}

extern "C" int xlz_ctx_trim(xlz_ctx *ctx, uint64_t *released)
{
    if (!ctx) return XLZ_ERR_BAD_ARG;
    HIP_TRY(hipSetDevice(ctx->device));
    if (released) *released = ctx->pool.idle_bytes();
    ctx->pool.drop_idle();
    return XLZ_OK;
}

extern "C" int xlz_ctx_last_call_stats(xlz_ctx *ctx, xlz_call_stats *out)
{
    if (!ctx || !out) return XLZ_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lock(ctx->mu);
    if (!ctx->have_last_call) return XLZ_ERR_BAD_ARG;
    *out = ctx->last_call;
    return XLZ_OK;
}

// Where a call of several wave rounds is cut into sub-batches whose upload, decode and download overlap: up to eight
// equal shares of the output, each at least one wave round of streams and half a GiB -- and in front of and behind them a
// piece a quarter that size: what stays exposed is the first piece's upload and the last one's download, and the launches
// of consecutive pieces overlap on two streams (xlz_ctx: stream2), so a small piece costs no idle wave slots.  (Sixteen
// equal pieces of 4096 streams were measured on the 65 536 x 64 KiB batch: 6 % slower than eight -- while piece k drains
// only piece k + 1 can fill its slots, and 4096 units do not fill 5120; profiles/r05/pipeline_pieces.txt.)  One sub-batch =
// the call is one wave round (or too small to bother): it overlaps its copies with its own decode (slices).
//
// ROUNDS mode (*rounds = true): the streams are long -- a wave round of them (4096) is a GiB of output and more, i.e. hundreds
// of milliseconds -- and the call has few rounds.  Overlapping pieces do nothing for such a call (8192 x 2 MiB are two rounds
// whatever is done, and a quarter piece in front only misaligns them: measured 14.8 GiB/s host to host, then 12.0 with the
// pieces at 16 per CU); instead every piece is exactly ONE round of 16 per CU, the pieces run one behind the other on ONE
// stream, and each of them overlaps its copies with its own decode like a call of one round does (slices; piece k + 1's
// upload runs under piece k's launches anyway).
static void plan_sub_batches(const xlz_stream_desc *streams, size_t n, std::vector<size_t> &cuts, bool *rounds)
{
    cuts.assign(1, 0);
    *rounds = false;
    uint64_t total = 0;
    for (size_t i = 0; i < n; i++) total += streams[i].out_cap;
    const size_t k_round = 4096; // streams of one wave round (16 waves on each of 256 CUs)
    if (n > 6144 && total / n >= (256u << 10)) { // (up to 6144 streams are ONE round of 20 / 24 per CU: one sliced piece)
        *rounds = true;
        for (size_t at = k_round; at < n; at += k_round) cuts.push_back(at);
        cuts.push_back(n);
        return;
    }
    const size_t n_sub = std::min<size_t>(std::min<size_t>(8, n / k_round), (size_t)(total >> 29));
    if (n_sub >= 2) {
        // shares: 1/4, 1, ..., 1, 1/4 (n_sub - 1 whole ones)
        const double whole = (double)total / ((double)(n_sub - 1) + 0.5);
        std::vector<double> bounds; // cumulative output at which a piece ends
        double acc_b = whole / 4;
        bounds.push_back(acc_b);
        for (size_t k = 0; k + 1 < n_sub; k++) bounds.push_back(acc_b += whole);
        uint64_t acc = 0;
        size_t next = 0;
        for (size_t i = 0; i < n && next < bounds.size(); i++) {
            acc += streams[i].out_cap;
            const bool small = next == 0; // (the last piece is what is left)
            const size_t least = small ? k_round / 8 : k_round / 2;
            if ((double)acc >= bounds[next] && i + 1 - cuts.back() >= least && n - (i + 1) >= k_round / 8) {
                cuts.push_back(i + 1);
                next++;
            }
        }
    }
    cuts.push_back(n);
}

extern "C" int xlz_decode_batch_plan(const xlz_stream_desc *streams, size_t n, size_t *cuts, size_t max_cuts, size_t *n_cuts, int *mode)
{
    if ((!streams && n) || !n_cuts || (!cuts && max_cuts)) return XLZ_ERR_BAD_ARG;
    std::vector<size_t> c;
    bool rounds = false;
    plan_sub_batches(streams, n, c, &rounds);
    *n_cuts = c.size();
    for (size_t k = 0; k < c.size() && k < max_cuts; k++) cuts[k] = c[k];
    if (mode) *mode = c.size() <= 2 ? 0 : rounds ? 2 : 1;
    return max_cuts && c.size() > max_cuts ? XLZ_ERR_OUT_CAP : XLZ_OK;
}

extern "C" int xlz_decode_batch(xlz_ctx *ctx, const xlz_stream_desc *streams, size_t n, xlz_result *results)
{
    if (!ctx || (!streams && n) || (!results && n)) return XLZ_ERR_BAD_ARG;
    for (size_t i = 0; i < n; i++)
        if (!streams[i].out && streams[i].out_cap) return XLZ_ERR_BAD_ARG;
    // upload (pinned image, one copy) -> decode -> download (pinned ring, D2H overlapped with the scatter into the
    // callers' buffers).  A call of several wave rounds runs as a PIPELINE of sub-batches on three host threads:
    // sub-batch k+1 is parsed, packed and uploaded and sub-batch k-1 is downloaded and scattered while sub-batch k
    // decodes; the launches are queued back to back on the context's stream.  (The reference's pump interleaves
    // producing and consuming by construction, reader1.go:223-254.)  Copies on a second stream run at full PCIe speed
    // under the persistent decode grid and do not slow it (tools/overlap_probe.py: 57 GB/s either way).
    const char *dbg = getenv("XLZ_DEBUG"); // debugging aid: log HIP failures and the phase times of this call
    const auto t0 = std::chrono::steady_clock::now();
    auto now_ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
    std::vector<size_t> cuts;
    bool rounds_mode = false;
    plan_sub_batches(streams, n, cuts, &rounds_mode);
    const size_t S = cuts.size() - 1;
    xlz_call_stats cs;
    memset(&cs, 0, sizeof cs);
    cs.streams = n;
    cs.sub_batches = (uint32_t)S;

    // A call of ONE sub-batch overlaps its OWN copies with its decode: it runs as a sequence of launches (slices) that each
    // advance every unit by a share of its output; share k - 1 is downloaded while share k decodes, and the first launch
    // starts on the heads of the inputs.
    uint64_t sliced_call_bytes, slice_bytes, max_slices;
    {
        std::lock_guard<std::mutex> lock(ctx->mu); // (once: the launching thread holds this lock most of the call)
        sliced_call_bytes = ctx->sliced_call_bytes, slice_bytes = ctx->slice_bytes, max_slices = ctx->max_slices;
    }
    auto slices_for = [&](size_t k) -> uint32_t {
        // (in a pipeline the pieces overlap each other; slices of the first or last piece were measured there: the next
        //  piece's workgroups take the slots every slice boundary frees and the two pieces finish together)
        if (S > 1 && !rounds_mode) return 1;
        uint64_t total = 0;
        for (size_t i = cuts[k]; i < cuts[k + 1]; i++) total += streams[i].out_cap;
        return total >= sliced_call_bytes ? (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(max_slices, total / slice_bytes)) : 1u;
    };

    std::vector<xlz_batch *> sub(S, nullptr);
    std::mutex mu;
    std::condition_variable cv;
    std::vector<int> created(S, 0), launched(S, 0), decoded(S, 0); // 0 pending, 1 done, -1 failed
    int st_up = XLZ_OK, st_down = XLZ_OK;
    bool abort_all = false;
    const bool threaded = S > 1;
    double t_first_up = 0, t_decoded = 0, occ_busy = 0, occ_span = 0;
    SlotClock clock;

    auto uploader = [&] {
        for (size_t k = 0; k < S; k++) {
            {
                std::lock_guard<std::mutex> lk(mu);
                if (abort_all) return;
            }
            xlz_batch *b = nullptr;
            BatchOpts o;
            o.want_slices = slices_for(k);
            o.head_frac = k == 0 ? head_frac_for(o.want_slices) : 0; // (only the first sub-batch's upload is exposed)
            o.call_units = (S > 1 && !rounds_mode) ? (uint32_t)std::min<size_t>(n, 0xFFFFFFFFu) : 0u; // (streams: a lower bound of the call's units)
            o.pooled = true;
            if (S > 1 && !rounds_mode && k % 2) o.run_stream = ctx->stream2, o.queue = ctx->queue2; // (xlz_ctx: stream2)
            const int st = batch_create_ex(ctx, streams + cuts[k], cuts[k + 1] - cuts[k], &b, o);
            if (dbg) fprintf(stderr, "xlz_decode_batch: sub-batch %zu uploaded at %.1f ms\n", k, now_ms());
            {
                std::lock_guard<std::mutex> lk(mu);
                sub[k] = b;
                created[k] = st == XLZ_OK ? 1 : -1;
                if (st != XLZ_OK) st_up = st, abort_all = true;
                if (k == 0) t_first_up = now_ms();
            }
            cv.notify_all();
            if (st != XLZ_OK) return;
        }
    };
    // this thread: launch sub-batch k as soon as it is uploaded, then collect k - 1
    int st = XLZ_OK;
    auto collect_one = [&](size_t k) {
        int e = xlz_batch_results(sub[k], results + cuts[k]);
        if (e == XLZ_OK) {
            uint32_t slots = 0;
            (void)xlz_batch_launch_info(sub[k], &slots, nullptr);
            cs.units += sub[k]->unit_results.size();
            cs.wave_slots = std::max(cs.wave_slots, slots);
            if (sub[k]->slice_fracs.empty()) clock.add(sub[k]->unit_results); // (a sliced call: the mean over its launches, from download_sliced)
        }
        if (dbg) fprintf(stderr, "xlz_decode_batch: sub-batch %zu decoded at %.1f ms\n", k, now_ms());
        {
            std::lock_guard<std::mutex> lk(mu);
            decoded[k] = e == XLZ_OK ? 1 : -1;
            if (e != XLZ_OK) abort_all = true;
            t_decoded = now_ms();
        }
        cv.notify_all();
        return e;
    };
    // true when sub-batch k's results are on the host (threaded: the launching thread collects them; else: do it here)
    auto await_decoded = [&](size_t k) {
        if (!threaded) {
            if (decoded[k] == 0) {
                const int e = collect_one(k);
                if (e != XLZ_OK) st_down = e;
            }
            return decoded[k] == 1;
        }
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return decoded[k] != 0 || abort_all; });
        return decoded[k] == 1;
    };
    auto downloader = [&] {
        for (size_t k = 0; k < S; k++) {
            int e = XLZ_OK;
            if (threaded) {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return launched[k] != 0 || abort_all; });
                if (launched[k] != 1) return;
            }
            xlz_batch *b = sub[k];
            if (!b->slice_fracs.empty()) {
                // sliced: all launches are queued; fetch what each one finishes while the next one decodes
                std::vector<double> occ;
                std::vector<size_t> again; // streams to fetch once more: units that fell short of a bound, ...
                e = download_sliced(b, streams + cuts[k], &occ, again);
                if (dbg)
                    fprintf(stderr, "xlz_decode_batch: sub-batch %zu: %zu slices downloaded at %.1f ms, %zu streams with gaps\n", k,
                            b->slice_fracs.size(), now_ms(), again.size());
                if (e == XLZ_OK && !await_decoded(k)) return;
                if (e == XLZ_OK) { // ... and bytes a re-run wrote after their slices had gone out
                    again.insert(again.end(), b->rewritten.begin(), b->rewritten.end());
                    std::sort(again.begin(), again.end());
                    again.erase(std::unique(again.begin(), again.end()), again.end());
                    if (!again.empty()) e = download_all(b, streams + cuts[k], results + cuts[k], &again);
                }
                std::lock_guard<std::mutex> lk(mu);
                cs.slices = std::max<uint32_t>(cs.slices, (uint32_t)b->slice_fracs.size());
                for (double o : occ) occ_busy += o * 1.0, occ_span += 1.0; // (weight: one launch = 1 ms; good enough for a mean)
            } else {
                if (!await_decoded(k)) return;
                e = download_all(b, streams + cuts[k], results + cuts[k]);
            }
            if (dbg) fprintf(stderr, "xlz_decode_batch: sub-batch %zu downloaded at %.1f ms\n", k, now_ms());
            if (e != XLZ_OK) {
                std::lock_guard<std::mutex> lk(mu);
                st_down = e;
                abort_all = true;
                cv.notify_all();
                return;
            }
        }
    };
    std::thread th_up, th_down;
    if (threaded) {
        try {
            th_up = std::thread(uploader);
            th_down = std::thread(downloader);
        } catch (...) { // no thread to be had: stop what has started and fail the call, do not terminate the process
            {
                std::lock_guard<std::mutex> lk(mu);
                abort_all = true;
            }
            cv.notify_all();
            if (th_up.joinable()) th_up.join();
            for (xlz_batch *b : sub) xlz_batch_destroy(b);
            return XLZ_ERR_DEVICE;
        }
    }
    double sliced_kernel_ms = -1;
    if (!threaded) {
        uploader();
        st = st_up;
        if (st == XLZ_OK) st = xlz_batch_run(sub[0]);
        if (st == XLZ_OK) {
            launched[0] = 1;
            downloader(); // (collects the results when it needs them)
            st = st_down;
            float ms = 0;
            if (st == XLZ_OK && !sub[0]->slice_fracs.empty() && xlz_batch_last_kernel_ms(sub[0], &ms) == XLZ_OK) sliced_kernel_ms = ms;
        }
    } else {
        for (size_t k = 0; k < S && st == XLZ_OK; k++) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return created[k] != 0 || abort_all; });
                if (created[k] != 1) break;
            }
            st = xlz_batch_run(sub[k]);
            {
                std::lock_guard<std::mutex> lk(mu);
                launched[k] = st == XLZ_OK ? 1 : -1;
                if (st != XLZ_OK) abort_all = true;
            }
            cv.notify_all();
            if (st == XLZ_OK && k > 0) st = collect_one(k - 1);
        }
        bool all_launched;
        {
            std::lock_guard<std::mutex> lk(mu);
            all_launched = created[S - 1] == 1 && launched[S - 1] == 1 && !abort_all;
        }
        if (st == XLZ_OK && all_launched) st = collect_one(S - 1);
        {
            std::lock_guard<std::mutex> lk(mu);
            if (st != XLZ_OK) abort_all = true;
        }
        cv.notify_all();
        th_up.join();
        th_down.join();
        if (st == XLZ_OK) st = st_up != XLZ_OK ? st_up : st_down;
    }
    const double t_end = now_ms();
    cs.upload_ms = t_first_up;             // until the first sub-batch was on the device
    cs.decode_ms = t_decoded - t_first_up; // first launch to the last results (uploads and downloads of the others inside)
    cs.download_ms = t_end - t_decoded;    // what was left to download when the last sub-batch had decoded
    if (sliced_kernel_ms >= 0) {           // sliced: the launches by their HIP events, the rest of the call is copies
        cs.decode_ms = sliced_kernel_ms;
        cs.download_ms = std::max(0.0, t_end - t_first_up - sliced_kernel_ms);
    }
    cs.total_ms = t_end;
    cs.kernel_span_ms = sliced_kernel_ms >= 0 ? sliced_kernel_ms : clock.span_ms();
    cs.slot_occupancy = occ_span > 0 ? occ_busy / occ_span : clock.occupancy(cs.wave_slots);
    {
        std::lock_guard<std::mutex> lock(ctx->mu);
        ctx->last_call = cs;
        ctx->have_last_call = st == XLZ_OK;
    }
    std::vector<size_t> big; // streams of 4 GiB and more do not fit a unit's 32-bit counters: sessions
    for (size_t k = 0; k < S && st == XLZ_OK; k++)
        for (size_t i = 0; i < sub[k]->n; i++)
            if (sub[k]->plans[i].oversize) big.push_back(cuts[k] + i);
    for (xlz_batch *b : sub) {
        if (b && st == XLZ_OK) b->quiet = true; // every launch has been collected, every byte fetched
        xlz_batch_destroy(b);                   // (the blocks return to the context's pool)
    }
    ctx->pool.end_of_call();
    if (st == XLZ_OK && !big.empty()) st = decode_oversize(ctx, streams, results, big);
    return st;
}

// Multi-GPU form of xlz_decode_batch.  The streams are independent, and so are the units of an
// LZMA2 stream (a chunk that resets the dictionary and brings new properties depends on nothing before it,
// reader2.go:100-173; the output offset of every unit follows from the chunk headers): the call is dealt to the contexts
// (one per GPU, each on its own host thread) as ITEMS -- whole streams, and runs of units of the LZMA2 streams that are a
// large part of the call -- by compressed bytes, longest first (the key of the kernel's own work queue; round 3 dealt whole
// streams by out_cap, so ONE big LZMA2 stream, one .xz file of few blocks or one .7z folder landed on one GPU).  A run of
// units is a slice of the compressed input and a disjoint slice of the caller's output buffer: no device-to-device
// traffic, no collective.  A stream whose slices do not come back exactly as their headers announce (malformed streams:
// a unit that produces or consumes something else, a copy that reads behind its dictionary reset into another slice's
// bytes) is decoded again as a whole on one context: bytes, status and in_consumed are the single-GPU call's.
namespace {

struct MultiItem {
    size_t stream;
    uint64_t in_off, in_len;   // slice of the stream's input (whole stream: 0, in_len)
    uint64_t out_off, out_len; // slice of its output (the announced size of the run of units)
    bool whole, first, last;
    xlz_result res;
};

constexpr uint64_t kMultiMinSplitBytes = 64u << 10; // an LZMA2 stream shorter than this is dealt whole

// cut an LZMA2 stream into at most `pieces` runs of units of about equal compressed size
void multi_split(const xlz_stream_desc &s, size_t stream, size_t pieces, std::vector<MultiItem> &items)
{
    std::vector<Lz2Unit> lu;
    uint32_t mx = 0;
    // (a stream with more room than a unit can address is dealt whole: its last slice would inherit that room and run as a
    //  session, which knows no slices -- ADVICE r4)
    if (pieces >= 2 && s.in_len <= kMaxUnitBytes && s.out_cap <= kMaxUnitBytes) scan_lzma2(s.in, s.in_len, lu, mx);
    uint64_t announced = 0;
    for (const Lz2Unit &u : lu) announced += u.expect_out;
    if (lu.size() < 2 || announced > s.out_cap) { // one unit, or the caller's room is short of the headers: whole
        MultiItem it{stream, 0, s.in_len, 0, s.out_cap, true, true, true, xlz_result{}};
        items.push_back(it);
        return;
    }
    pieces = std::min(pieces, lu.size());
    size_t k = 0;
    for (size_t p = 0; p < pieces; p++) {
        const size_t k0 = k;
        const uint64_t until = (uint64_t)s.in_len * (p + 1) / pieces;
        k++;
        while (k < lu.size() && (p + 1 == pieces || lu[k].in_start + lu[k].in_len <= until) && lu.size() - k > pieces - p - 1) k++;
        if (p + 1 == pieces) k = lu.size();
        MultiItem it;
        it.stream = stream;
        it.in_off = lu[k0].in_start;
        it.in_len = (k == lu.size() ? (uint64_t)s.in_len : lu[k].in_start) - it.in_off;
        it.out_off = lu[k0].out_start;
        it.out_len = (k == lu.size() ? announced : lu[k].out_start) - it.out_off;
        it.whole = false;
        it.first = k0 == 0;
        it.last = k == lu.size();
        it.res = xlz_result{};
        items.push_back(it);
    }
}

} // namespace

namespace {

// items of a call and the context each goes to
void multi_plan(size_t n_ctx, const xlz_stream_desc *streams, size_t n, std::vector<MultiItem> &items,
                std::vector<std::vector<size_t>> &shard)
{
    // ---- items: an LZMA2 stream that holds more than a quarter of a context's share is cut into runs of units
    uint64_t total_in = 0;
    for (size_t i = 0; i < n; i++) total_in += streams[i].in_len;
    const uint64_t share = std::max<uint64_t>(1, total_in / n_ctx);
    items.clear();
    items.reserve(n + 4 * n_ctx);
    for (size_t i = 0; i < n; i++) {
        const xlz_stream_desc &s = streams[i];
        size_t pieces = 1;
        if (n_ctx > 1 && s.format == XLZ_FMT_LZMA2_RAW && !(s.flags & XLZ_STREAM_F_LZMA2_SLICE) && s.in_len >= kMultiMinSplitBytes &&
            s.in_len > share / 4)
            pieces = (size_t)std::min<uint64_t>(4 * n_ctx, (s.in_len + share / 4 - 1) / std::max<uint64_t>(1, share / 4));
        multi_split(s, i, pieces, items);
    }
    // ---- deal: longest compressed size first, to the context with the least so far
    std::vector<size_t> order(items.size());
    std::iota(order.begin(), order.end(), (size_t)0);
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return items[a].in_len > items[b].in_len; });
    shard.assign(n_ctx, {});
    std::vector<uint64_t> load(n_ctx, 0);
    for (size_t i : order) {
        size_t k = 0;
        for (size_t c = 1; c < n_ctx; c++)
            if (load[c] < load[k]) k = c;
        shard[k].push_back(i);
        load[k] += items[i].in_len + 1;
    }
    for (auto &idx : shard) std::sort(idx.begin(), idx.end());
}

} // namespace

// the plan alone (host only, no GPU): which slice of which stream xlz_decode_batch_multi would hand to which context
extern "C" int xlz_decode_batch_multi_plan(size_t n_ctx, const xlz_stream_desc *streams, size_t n, xlz_multi_item *out,
                                           size_t max_items, size_t *n_items)
{
    if (!n_ctx || (!streams && n) || !n_items || (!out && max_items)) return XLZ_ERR_BAD_ARG;
    for (size_t i = 0; i < n; i++)
        if (!streams[i].in && streams[i].in_len) return XLZ_ERR_BAD_ARG;
    std::vector<MultiItem> items;
    std::vector<std::vector<size_t>> shard;
    multi_plan(n_ctx, streams, n, items, shard);
    *n_items = items.size();
    for (size_t c = 0; c < n_ctx; c++)
        for (size_t k : shard[c]) {
            if (k >= max_items) continue;
            const MultiItem &it = items[k];
            out[k].stream = it.stream;
            out[k].context = (uint32_t)c;
            out[k].flags = (it.whole ? 1u : 0u) | (it.first ? 2u : 0u) | (it.last ? 4u : 0u);
            out[k].in_off = it.in_off;
            out[k].in_len = it.in_len;
            out[k].out_off = it.out_off;
            out[k].out_len = it.out_len;
        }
    return max_items && items.size() > max_items ? XLZ_ERR_OUT_CAP : XLZ_OK;
}

extern "C" int xlz_decode_batch_multi(xlz_ctx *const *ctxs, size_t n_ctx, const xlz_stream_desc *streams, size_t n,
                                      xlz_result *results)
{
    if (!ctxs || n_ctx == 0 || (!streams && n) || (!results && n)) return XLZ_ERR_BAD_ARG;
    for (size_t c = 0; c < n_ctx; c++)
        if (!ctxs[c]) return XLZ_ERR_BAD_ARG;
    for (size_t i = 0; i < n; i++)
        if ((!streams[i].in && streams[i].in_len) || (!streams[i].out && streams[i].out_cap)) return XLZ_ERR_BAD_ARG;
    if (n_ctx == 1 || n == 0) return xlz_decode_batch(ctxs[0], streams, n, results);
    std::vector<MultiItem> items;
    std::vector<std::vector<size_t>> shard;
    multi_plan(n_ctx, streams, n, items, shard);
    std::vector<int> st(n_ctx, XLZ_OK);
    auto work = [&](size_t c) {
        std::vector<size_t> &idx = shard[c];
        if (idx.empty()) return;
        std::vector<xlz_stream_desc> d(idx.size());
        std::vector<xlz_result> r(idx.size());
        for (size_t k = 0; k < idx.size(); k++) {
            const MultiItem &it = items[idx[k]];
            d[k] = streams[it.stream];
            if (!it.whole) {
                d[k].in += it.in_off;
                d[k].in_len = (size_t)it.in_len;
                d[k].out += it.out_off;
                d[k].out_cap = it.last ? (size_t)(streams[it.stream].out_cap - it.out_off) : (size_t)it.out_len;
                if (!it.first) d[k].flags |= XLZ_STREAM_F_LZMA2_SLICE;
            }
        }
        st[c] = xlz_decode_batch(ctxs[c], d.data(), d.size(), r.data());
        if (st[c] == XLZ_OK)
            for (size_t k = 0; k < idx.size(); k++) items[idx[k]].res = r[k];
    };
    std::vector<std::thread> th;
    for (size_t c = 1; c < n_ctx; c++) th.emplace_back(work, c);
    work(0);
    for (auto &x : th) x.join();
    for (size_t c = 0; c < n_ctx; c++)
        if (st[c] != XLZ_OK) return st[c];
    // ---- fold: a stream dealt in slices is done when every slice came back exactly as its headers announce
    std::vector<size_t> again;
    for (size_t k = 0; k < items.size();) {
        const size_t i = items[k].stream;
        if (items[k].whole) {
            results[i] = items[k].res;
            k++;
            continue;
        }
        bool clean = true;
        xlz_result sum{};
        size_t e = k;
        for (; e < items.size() && items[e].stream == i; e++) {
            const MultiItem &it = items[e];
            if (it.last)
                clean = clean && it.res.status == XLZ_OK;
            else
                clean = clean && it.res.status == XLZ_ERR_UNEXPECTED_EOF && it.res.out_len == it.out_len &&
                        it.res.in_consumed == it.in_len;
            sum.out_len += it.res.out_len;
            if (it.last) {
                sum.in_consumed = it.in_off + it.res.in_consumed;
                sum.status = it.res.status;
            }
        }
        if (clean)
            results[i] = sum;
        else
            again.push_back(i);
        k = e;
    }
    if (!again.empty()) { // (malformed streams only) as a whole, on the first context: what the single-GPU call returns
        std::vector<xlz_stream_desc> d(again.size());
        std::vector<xlz_result> r(again.size());
        for (size_t k = 0; k < again.size(); k++) d[k] = streams[again[k]];
        const int s2 = xlz_decode_batch(ctxs[0], d.data(), d.size(), r.data());
        if (s2 != XLZ_OK) return s2;
        for (size_t k = 0; k < again.size(); k++) results[again[k]] = r[k];
    }
    return XLZ_OK;
}

// ---------------------------------------------------------------- readers ----
// Pull-style mirror of Reader1 / Reader2 / readCloser (reader1.go:223-254, reader2.go:216-250,
// readcloser.go:9-41).
//
// A reader is a SESSION on the device: the unit's decoder state (range coder, reps, window
// position, the whole probability model) is saved in HBM when the wave has produced the next
// `kChunk` bytes and restored by the next launch (UnitState, xlz_format.h) -- the counterpart of
// the reference's decompress(need) returning once window.pending >= need (decompress.go:13).
// Memory is bounded like the reference's (O(dictSize), window.go:18-29): the device keeps the last
// dictSize bytes of output in front of the bytes being produced and slides them down when the
// buffer is full; the host holds one chunk.  Nothing is ever decoded twice and the input is read
// in one pass (it is fed to the device through a bounded window too).
struct Session {
    uint8_t *d_ctl = nullptr;   // Unit | order | UnitResult | UnitState | input window
    uint8_t *d_win = nullptr;   // output window: [history | bytes of this refill | slack]
    uint8_t *d_shadow = nullptr; // LZMA2: the reference's window buffer as of the last dictionary reset (dictSize bytes):
                                 // what a copy reads behind a reset once the epoch's bytes have slid out of d_win
                                 // (window.go:135-140 does not clear the buffer; the wave keeps the image, xlz_kernel.hip)
    size_t win_cap = 0, win_max = 0;
    size_t off_state = 0, off_in = 0, in_buf = 0;
    Unit unit;
    uint32_t model_lc_lp = 0;
    uint32_t pos = 0;           // bytes in d_win after the last launch
    uint32_t rebase = 0;        // the window was moved down by this much since the last launch
    uint64_t in_skip = 0;       // input bytes in front of the device's input window
    uint64_t in_loaded = 0;     // input bytes uploaded so far (absolute end of the window)
    uint64_t consumed = 0;      // input bytes the decoder has read
    bool started = false;
    size_t payload_off = 0;     // where the unit's input starts inside reader.in (header stripped)
    uint64_t left = kUnknownSize; // bytes the stream can still produce (known unpack size), or unknown
};

struct xlz_reader {
    xlz_ctx *ctx = nullptr;
    xlz_stream_desc desc;
    std::vector<uint8_t> in;     // the reader's own copy of the compressed stream ...
    const uint8_t *src = nullptr; // ... or a borrowed buffer (oversize streams of xlz_decode_batch): src/src_len is
    size_t src_len = 0;          // what the session reads
    std::vector<uint8_t> chunk;  // decoded bytes not yet handed to Read
    size_t rd = 0;
    bool finished = false;       // the stream's end (or error) has been reached; status is final
    bool closed = false;
    bool is_closer = false;      // built by a *ForSevenZip constructor: wraps errors like readCloser
    int32_t status = XLZ_OK;
    int call_status = XLZ_OK;    // a device failure (not a stream status)
    int step_status = XLZ_OK;    // of the launch that carried this reader in the last sessions_step (readers of the other launch keep theirs)
    uint64_t delivered = 0;      // decoded bytes handed to the chunk buffer so far
    Session *ss = nullptr;
    bool whole = false;          // fallback: decode the whole stream in one exact batch (see reader_whole)
    // batching: refills of concurrent readers are decoded by the context's batcher thread in one launch
    bool refill_pending = false;
    uint64_t n_refills = 0, n_whole = 0, n_shadow = 0;
    bool pending_reset = false, pending_reopen = false; // (*Reader1).Reset / Reopen before the next refill
    // streaming input (xlz_reader_expect_more / _feed / _feed_eof): `in` holds the bytes from stream
    // offset in_base on; what the decoder has consumed is dropped at every feed
    bool streaming = false, in_eof = false, need_input = false;
    bool input_fresh = true; // nothing of the current input (constructor's or Reopen's) has been decoded yet: expect_more is allowed
    uint64_t in_base = 0;
    // unit-parallel LZMA2 (reader_parallel): the stream's dictionary-reset units, the next one to decode
    std::vector<uint64_t> par_in, par_out; // prefix sums of the units' input / output bytes
    size_t par_next = 0;
    bool par = false, par_tried = false;
    uint64_t n_par = 0;
};

// Background coalescer of readers (one per context).
struct Batcher {
    xlz_ctx *ctx = nullptr;
    uint32_t window_us = 500, max_streams = 4096;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::deque<xlz_reader *> pending;
    bool stop = false;
    uint64_t n_batches = 0, n_streams = 0;
    std::thread th;
};

namespace {

constexpr size_t kChunk = 1u << 20;       // bytes a refill aims for (decompress(need), need = 1 MiB)
constexpr size_t kWinSlack = 65536 + 1024; // a refill overshoots by < one stored chunk / one match + a 64-lane row
constexpr size_t kInBuf = 4u << 20;       // device-side input window of a session
constexpr size_t kCtlHead = 256;          // Unit (80) | order (4) | UnitResult (32)

xlz_reader *reader_new(xlz_ctx *ctx, const uint8_t *in, size_t in_len)
{
    xlz_reader *r = new (std::nothrow) xlz_reader;
    if (!r) return nullptr;
    r->ctx = ctx;
    r->in.assign(in, in + in_len);
    r->src = r->in.data();
    r->src_len = r->in.size();
    memset(&r->desc, 0, sizeof r->desc);
    return r;
}

// rangeDec.Init as seen by a constructor (range_decoder.go:27-46, reader1.go:153-156)
int check_rc_init(const uint8_t *p, size_t n)
{
    if (n == 0) return XLZ_ERR_HEADER_EOF;
    if (p[0] != 0) return XLZ_ERR_RC_INIT;
    if (n < 5) return XLZ_ERR_HEADER_EOF;
    return XLZ_OK;
}

void session_free(Session *ss)
{
    if (!ss) return;
    if (ss->d_ctl) (void)hipFree(ss->d_ctl);
    if (ss->d_win) (void)hipFree(ss->d_win);
    if (ss->d_shadow) (void)hipFree(ss->d_shadow);
    delete ss;
}

// Device side of a reader.  XLZ_OK, XLZ_ERR_DEVICE, or XLZ_ERR_UNSUPPORTED (model too large for
// LDS: the whole-stream path decodes those).
int session_open(xlz_reader *r)
{
    Session *ss = new (std::nothrow) Session;
    if (!ss) return XLZ_ERR_DEVICE;
    Unit &u = ss->unit;
    memset(&u, 0, sizeof u);
    uint64_t known = kUnknownSize;
    if (r->desc.format == XLZ_FMT_LZMA2_RAW) {
        std::vector<Lz2Unit> lu;
        uint32_t mx = 0;
        scan_lzma2(r->src, r->src_len, lu, mx);
        u.kind = UNIT_LZMA2;
        u.dict_size = r->desc.dict_size < kLzmaDicMin ? 8u * 1024 * 1024 : r->desc.dict_size; // reader2.go:88-91
        u.unpack_size = kUnknownSize;
        if (r->streaming && mx < 4) mx = 4; // later chunks may bring other properties: room for all that lc+lp <= 4 allows
        u.lc = (uint8_t)mx;
        ss->model_lc_lp = mx;
    } else {
        xlz_stream_desc d = r->desc;
        d.in = r->src;
        d.in_len = r->src_len;
        StreamPlan pl;
        bool has = false;
        if (r->desc.format == XLZ_FMT_LZMA_ALONE)
            plan_lzma_alone(d, pl, u, has);
        else
            plan_lzma_raw(d, pl, u, has);
        if (!has) { // the constructors have already refused these
            delete ss;
            return XLZ_ERR_BAD_ARG;
        }
        ss->payload_off = pl.header_len;
        ss->model_lc_lp = (uint32_t)u.lc + u.lp;
        known = u.unpack_size;
    }
    if (ss->model_lc_lp > kMaxLcLp) { // (the constructors have refused these: DecodeProp)
        delete ss;
        return XLZ_ERR_UNSUPPORTED;
    }
    // lc+lp > 8 does not fit a CU's LDS: the unit runs in the HBM-model launch (sessions_step), its model saved and
    // restored like any other
    u.flags = UNIT_F_LAST;
    // output window: twice the dictionary (history + room to slide without overlap) + one refill;
    // a stream of known size never needs more than its size; start small, grow on demand
    ss->win_max = 2 * (size_t)u.dict_size + kChunk + kWinSlack;
    if (known != kUnknownSize && known + kWinSlack < ss->win_max) ss->win_max = (size_t)known + kWinSlack;
    // positions inside the window are 32-bit: a dictionary beyond ~2 GiB gets a window it cannot
    // slide in; such a stream decodes while it fits and is refused beyond (session_prepare)
    if (ss->win_max > kMaxUnitBytes) ss->win_max = (size_t)kMaxUnitBytes;
    ss->win_cap = std::min<size_t>(ss->win_max, 2 * kChunk + kWinSlack);
    ss->left = known;
    ss->in_buf = r->streaming ? kInBuf : std::min<size_t>(kInBuf, align_up(r->src_len - ss->payload_off + 16, 256) + kArenaTailPad);
    ss->off_state = kCtlHead;
    ss->off_in = align_up(ss->off_state + state_bytes(ss->model_lc_lp), 256);
    if (hipMalloc(&ss->d_ctl, ss->off_in + ss->in_buf) != hipSuccess || hipMalloc(&ss->d_win, ss->win_cap) != hipSuccess) {
        session_free(ss);
        return XLZ_ERR_DEVICE;
    }
    const uint32_t zero = 0;
    if (hipMemcpy(ss->d_ctl + 128, &zero, 4, hipMemcpyHostToDevice) != hipSuccess) { // order[0] = 0
        session_free(ss);
        return XLZ_ERR_DEVICE;
    }
    if (u.kind == UNIT_LZMA2) {
        // the window image for reads behind a dictionary reset is allocated when the stream first needs one
        // (session_make_shadow, asked for by the wave with AUX_SHADOW); until then the state header says "none"
        const uint64_t none = 0;
        if (hipMemcpy(ss->d_ctl + ss->off_state + (size_t)kStateShadowWord * 4, &none, 8, hipMemcpyHostToDevice) != hipSuccess) {
            session_free(ss);
            return XLZ_ERR_DEVICE;
        }
    }
    r->ss = ss;
    return XLZ_OK;
}

// Make room for the next refill (slide the history down, or grow the buffer), move the input window
// and fill in the unit.  Called with ctx->mu held, device set.
int session_prepare(xlz_reader *r, hipStream_t stream)
{
    Session *ss = r->ss;
    Unit &u = ss->unit;
    // ---- output window: room for what this refill can produce (a stream of known size that ends inside the
    // refill needs only its rest: its window was sized by that, session_open)
    const size_t need = ss->left == kUnknownSize ? kChunk : (size_t)std::min<uint64_t>(kChunk, ss->left);
    if ((size_t)ss->pos + need + kWinSlack > ss->win_cap) {
        const size_t keep = std::min<size_t>(ss->pos, u.dict_size);
        if (ss->win_cap < ss->win_max) { // grow (the history may still be shorter than the dictionary)
            size_t want = std::min(ss->win_max, std::max(ss->win_cap * 2, (size_t)ss->pos + need + kWinSlack));
            uint8_t *nw = nullptr;
            if (hipMalloc(&nw, want) != hipSuccess) return XLZ_ERR_DEVICE;
            if (hipMemcpyAsync(nw, ss->d_win, ss->pos, hipMemcpyDeviceToDevice, stream) != hipSuccess ||
                hipStreamSynchronize(stream) != hipSuccess) {
                (void)hipFree(nw);
                return XLZ_ERR_DEVICE;
            }
            (void)hipFree(ss->d_win);
            ss->d_win = nw;
            ss->win_cap = want;
        }
        if ((size_t)ss->pos + need + kWinSlack > ss->win_cap) {
            // full size reached: only the last dictSize bytes can still be referenced (window.go:18-29);
            // win_max = 2 * dictSize + ... guarantees source and destination do not overlap
            if ((size_t)ss->pos - keep < keep) return XLZ_ERR_UNSUPPORTED; // dictionary > 2 GiB on a stream > 4 GiB
            if (hipMemcpyAsync(ss->d_win, ss->d_win + (ss->pos - keep), keep, hipMemcpyDeviceToDevice, stream) != hipSuccess)
                return XLZ_ERR_DEVICE;
            ss->rebase += (uint32_t)(ss->pos - keep);
            ss->pos = (uint32_t)keep;
        }
    }
    // ---- input window: [in_skip, in_loaded) of the payload is on the device
    // (payload offsets; the host holds the stream from offset r->in_base on, all of it unless it is fed in pieces)
    const uint64_t total = r->in_base + r->src_len - ss->payload_off; // payload bytes the host has seen so far
    const bool more_coming = r->streaming && !r->in_eof;
    const uint64_t margin = u.kind == UNIT_LZMA2 ? 70000 : 4096;
    if (!ss->started || (ss->in_loaded < total && ss->in_loaded - ss->consumed < margin)) {
        ss->in_skip = ss->consumed & ~(uint64_t)255;
        const uint64_t end = std::min<uint64_t>(total, ss->in_skip + ss->in_buf - kArenaTailPad);
        if (hipMemcpyAsync(ss->d_ctl + ss->off_in, r->src + (ss->payload_off + ss->in_skip - r->in_base), (size_t)(end - ss->in_skip),
                           hipMemcpyHostToDevice, stream) != hipSuccess ||
            hipMemsetAsync(ss->d_ctl + ss->off_in + (end - ss->in_skip), 0, kArenaTailPad, stream) != hipSuccess ||
            hipStreamSynchronize(stream) != hipSuccess) // the source is pageable: the copy has been staged when this returns
            return XLZ_ERR_DEVICE;
        ss->in_loaded = end;
    }
    u.in_off = (uint64_t)(ss->d_ctl + ss->off_in);
    u.in_len = (uint32_t)(ss->in_loaded - ss->in_skip);
    u.in_skip = (uint32_t)ss->in_skip;
    u.out_off = (uint64_t)ss->d_win;
    u.out_cap = (uint32_t)(ss->win_cap - kOutTailPad);
    u.pause_at = ss->pos + (uint32_t)kChunk;
    u.rebase = ss->rebase;
    u.state = (uint64_t)(ss->d_ctl + ss->off_state);
    u.flags = UNIT_F_LAST | (ss->started ? UNIT_F_RESUME : 0u) |
              ((ss->in_loaded < total || more_coming) ? UNIT_F_MORE_INPUT : 0u);
    if (ss->started && r->pending_reset) u.flags |= UNIT_F_RESET_MODEL; // before the first launch the model is fresh anyway
    if (ss->started && r->pending_reopen) u.flags |= UNIT_F_REOPEN;
    // (pending_reset / pending_reopen are cleared by sessions_step once the launch that consumed them has succeeded)
    return XLZ_OK;
}

// Give a session's unit a state block for a larger model (the wave paused in front of an LZMA2 chunk whose properties
// need it, AUX_GROW): the 256-byte register header moves over, the model itself is re-initialised by that chunk
// (new properties always come with a state reset, reader2.go:155-165), the input window moves behind the new block.
int session_grow_model(xlz_reader *r, uint32_t need, hipStream_t stream)
{
    Session *ss = r->ss;
    if (need > kMaxLcLp) return XLZ_ERR_UNSUPPORTED;
    const size_t new_off_in = align_up(ss->off_state + state_bytes(need), 256);
    uint8_t *nc = nullptr;
    if (hipMalloc(&nc, new_off_in + ss->in_buf) != hipSuccess) return XLZ_ERR_DEVICE;
    if (hipMemcpyAsync(nc, ss->d_ctl, ss->off_state + kStateWords * 4, hipMemcpyDeviceToDevice, stream) != hipSuccess ||
        hipMemcpyAsync(nc + new_off_in, ss->d_ctl + ss->off_in, ss->in_buf, hipMemcpyDeviceToDevice, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess) {
        (void)hipFree(nc);
        return XLZ_ERR_DEVICE;
    }
    (void)hipFree(ss->d_ctl);
    ss->d_ctl = nc;
    ss->off_in = new_off_in;
    ss->model_lc_lp = need;
    ss->unit.lc = (uint8_t)need; // LZMA2 units: lc = the largest lc+lp the model storage holds, lp = pb = 0
    return XLZ_OK;
}

// The wave stands in front of a dictionary reset that ends a non-empty epoch (AUX_SHADOW): from here on a copy may read
// what earlier epochs left in the reference's uncleared window buffer (window.go:135-140).  The image is allocated now --
// dictSize bytes, zero-filled like the reference's window (window.go:18-29) -- and its address goes into the state
// header, where the resumed wave looks for it.  Streams whose only dictionary reset is their first chunk's (most of
// them) never get here: through round 3 every NewReader2 paid dictSize bytes of HBM and a memset up front.
int session_make_shadow(xlz_reader *r)
{
    Session *ss = r->ss;
    if (!ss->d_shadow) {
        const size_t n = ss->unit.dict_size;
        uint8_t *img = nullptr;
        if (hipMalloc(&img, n) != hipSuccess) return XLZ_ERR_DEVICE;
        if (hipMemset(img, 0, n) != hipSuccess) {
            (void)hipFree(img);
            return XLZ_ERR_DEVICE;
        }
        ss->d_shadow = img;
        r->n_shadow++;
    }
    // (written whenever the wave asks: a call that failed behind the allocation must not leave the wave asking for ever)
    const uint64_t addr = (uint64_t)ss->d_shadow;
    if (hipMemcpy(ss->d_ctl + ss->off_state + (size_t)kStateShadowWord * 4, &addr, 8, hipMemcpyHostToDevice) != hipSuccess)
        return XLZ_ERR_DEVICE;
    return XLZ_OK;
}

int sessions_launch(xlz_ctx *ctx, const std::vector<xlz_reader *> &rs, bool big);

// One refill of every reader in `all`: each reader's unit continues from its saved state and stops after about
// kChunk more bytes.  The readers whose model fits LDS share ONE launch, those with lc+lp > 8 another (HBM-model
// kernel).  Fills r->chunk / r->finished / r->status.
int sessions_step(xlz_ctx *ctx, const std::vector<xlz_reader *> &all)
{
    if (all.empty()) return XLZ_OK;
    std::lock_guard<std::mutex> lock(ctx->mu);
    // (every way out of this function leaves a status in every reader: the batcher looks at nothing else -- ADVICE r4: an
    //  early return in front of the first assignment left the readers at their previous XLZ_OK, reader_refill came back
    //  with an empty chunk and xlz_reader_read span)
    for (xlz_reader *r : all) r->step_status = XLZ_ERR_DEVICE;
    HIP_TRY(hipSetDevice(ctx->device));
    std::vector<xlz_reader *> normal, bigs; // one that cannot be prepared ends alone
    for (xlz_reader *r : all) {
        int st = session_prepare(r, ctx->stream);
        if (st == XLZ_ERR_UNSUPPORTED) { // this stream ends here; the others are not affected
            r->finished = true;
            r->status = XLZ_ERR_UNSUPPORTED;
            r->chunk.clear();
            r->rd = 0;
            continue;
        }
        if (st != XLZ_OK) { // the device failed: nothing was launched, pending Reset / Reopen flags are still set
            for (xlz_reader *q : all) q->step_status = st;
            return st;
        }
        (r->ss->model_lc_lp > kMaxLcLpLds ? bigs : normal).push_back(r);
    }
    // Two launches (LDS models, HBM models), two outcomes: the readers of a launch that succeeded have advanced
    // and keep their bytes whatever happened to the other launch; only the failing launch's readers see the failure.
    for (xlz_reader *r : all) r->step_status = XLZ_OK;
    const int st_normal = normal.empty() ? XLZ_OK : sessions_launch(ctx, normal, false);
    const int st_big = bigs.empty() ? XLZ_OK : sessions_launch(ctx, bigs, true);
    for (xlz_reader *r : normal) r->step_status = st_normal;
    for (xlz_reader *r : bigs) r->step_status = st_big;
    return st_normal != XLZ_OK ? st_normal : st_big;
}

// ONE launch over prepared sessions of one kind (ctx->mu held, device set)
int sessions_launch(xlz_ctx *ctx, const std::vector<xlz_reader *> &rs, bool big)
{
    const size_t n = rs.size();
    uint32_t max_lc_lp = 0;
    std::vector<Unit> units(n);
    for (size_t i = 0; i < n; i++) {
        units[i] = rs[i]->ss->unit;
        if (big) units[i].flags |= UNIT_F_BIG_MODEL;
        max_lc_lp = std::max(max_lc_lp, rs[i]->ss->model_lc_lp);
    }
    // units / order / results of this launch: in the first reader's control block when it is alone,
    // else in a scratch allocation
    Unit *d_units = nullptr;
    uint32_t *d_order = nullptr;
    UnitResult *d_res = nullptr;
    uint8_t *scratch = nullptr;
    if (n == 1) {
        d_units = reinterpret_cast<Unit *>(rs[0]->ss->d_ctl);
        d_order = reinterpret_cast<uint32_t *>(rs[0]->ss->d_ctl + 128);
        d_res = reinterpret_cast<UnitResult *>(rs[0]->ss->d_ctl + 160);
    } else {
        const size_t o_order = align_up(n * sizeof(Unit), 256), o_res = o_order + align_up(n * 4, 256);
        HIP_TRY(hipMalloc(&scratch, o_res + n * sizeof(UnitResult)));
        d_units = reinterpret_cast<Unit *>(scratch);
        d_order = reinterpret_cast<uint32_t *>(scratch + o_order);
        d_res = reinterpret_cast<UnitResult *>(scratch + o_res);
        std::vector<uint32_t> order(n);
        std::iota(order.begin(), order.end(), 0u);
        if (hipMemcpyAsync(d_order, order.data(), n * 4, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) {
            (void)hipFree(scratch);
            return XLZ_ERR_DEVICE;
        }
    }
    uint16_t *d_mlit = nullptr; // resumable units keep their matched-literal tables in their state blocks
    uint16_t *d_big = nullptr;  // HBM-model launch: one model slot per workgroup (restored from / saved to the unit's state)
    uint32_t big_stride = 0;
    if (big) {
        big_stride = num_probs(max_lc_lp) + num_matched_probs(max_lc_lp);
        const size_t grid = std::min<size_t>(n, big_model_grid(ctx->num_cus));
        if (hipMalloc(&d_big, grid * big_stride * sizeof(uint16_t)) != hipSuccess) {
            if (scratch) (void)hipFree(scratch);
            return XLZ_ERR_DEVICE;
        }
    }
    int st = XLZ_ERR_DEVICE;
    std::vector<UnitResult> res(n);
    if (hipMemcpyAsync(d_units, units.data(), n * sizeof(Unit), hipMemcpyHostToDevice, ctx->stream) == hipSuccess &&
        hipMemsetAsync(ctx->queue, 0, 256, ctx->stream) == hipSuccess) {
        LaunchParams p;
        memset(&p, 0, sizeof p); // (epochs, order_base, ...: off unless set below)
        p.in_arena = nullptr; // units carry absolute device addresses
        p.out_arena = nullptr;
        p.units = d_units;
        p.order = d_order;
        p.results = d_res;
        p.queue = ctx->queue;
        p.n_units = (uint32_t)n;
        p.max_lc_lp = max_lc_lp;
        p.mlit = d_mlit;
        p.mlit_stride = 0;
        p.scratch = d_big;
        p.scratch_stride = big_stride;
        p.prio_tab = ctx->prio_tab;
        if (launch_decode(p, ctx->num_cus, ctx->stream) == 0 &&
            hipMemcpyAsync(res.data(), d_res, n * sizeof(UnitResult), hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
            hipStreamSynchronize(ctx->stream) == hipSuccess)
            st = XLZ_OK;
    }
    if (st == XLZ_OK) {
        for (size_t i = 0; i < n; i++) {
            xlz_reader *r = rs[i];
            Session *ss = r->ss;
            const UnitResult &u = res[i];
            const uint32_t new_pos = (uint32_t)u.out_len;
            ss->started = true;
            ss->rebase = 0;
            r->pending_reset = r->pending_reopen = false; // this launch has applied them
            // (the device counts input modulo 2^32, relative arithmetic only; the window is < 4 GiB)
            ss->consumed = ss->in_skip + (uint32_t)((uint32_t)u.in_consumed - (uint32_t)ss->in_skip);
            r->n_refills++;
            if (u.status == ST_PAUSED && (u.aux & AUX_GROW)) {
                // the next chunk's properties need a larger model than the state block holds: nothing of it has been
                // decoded; bytes produced before it are delivered below, the next refill resumes on the larger block
                const int g = session_grow_model(r, (u.aux & AUX_GROW_MASK) >> AUX_GROW_SHIFT, ctx->stream);
                if (g == XLZ_ERR_DEVICE) st = XLZ_ERR_DEVICE;
                if (g == XLZ_ERR_UNSUPPORTED) {
                    r->finished = true;
                    r->status = XLZ_ERR_UNSUPPORTED;
                    r->chunk.clear();
                    r->rd = 0;
                    continue;
                }
            }
            if (u.status == ST_PAUSED && (u.aux & AUX_SHADOW) && session_make_shadow(r) != XLZ_OK) st = XLZ_ERR_DEVICE;
            if (((u.aux & AUX_STALE) || u.status == ST_ERR_UNSUPPORTED) && r->streaming) {
                // (no whole-stream fallback when the input is fed in pieces and dropped behind the decoder)
                r->finished = true;
                r->status = XLZ_ERR_UNSUPPORTED;
                r->chunk.clear();
                r->rd = 0;
                continue;
            }
            if ((u.aux & AUX_STALE) || u.status == ST_ERR_UNSUPPORTED) {
                // a copy reached across an LZMA2 dictionary reset (the bytes of this refill are not exact),
                // or the stream's real properties exceed what its headers announced: malformed streams
                // only.  The whole-stream path has the epoch table / the HBM model and skips what has
                // been delivered.
                r->whole = true;
                continue;
            }
            const size_t fresh = new_pos > ss->pos ? new_pos - ss->pos : 0;
            r->chunk.resize(fresh);
            r->rd = 0;
            if (fresh && hipMemcpyAsync(r->chunk.data(), ss->d_win + ss->pos, fresh, hipMemcpyDeviceToHost, ctx->stream) !=
                             hipSuccess)
                st = XLZ_ERR_DEVICE;
            ss->pos = new_pos;
            r->delivered += fresh;
            if (ss->left != kUnknownSize) ss->left -= std::min<uint64_t>(ss->left, fresh);
            if (u.status != ST_PAUSED) {
                r->finished = true;
                r->status = u.status;
            } else if ((u.aux & AUX_NEED_INPUT) && r->streaming && !r->in_eof &&
                       ss->in_loaded >= r->in_base + r->src_len - ss->payload_off) {
                r->need_input = true; // everything the host has is on the device and the wave wants more
            }
        }
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) st = XLZ_ERR_DEVICE;
    }
    if (scratch) (void)hipFree(scratch);
    if (d_big) (void)hipFree(d_big);
    return st;
}

// Fallback for what a session cannot do (models beyond LDS; malformed LZMA2 streams that read
// across a dictionary reset): the whole stream through the batch path, which has the HBM-model
// launch and the exact (epoch table) launch; the bytes already delivered are skipped.
int reader_whole(xlz_reader *r)
{
    r->n_whole++;
    if (r->src_len > kMaxUnitBytes || r->delivered > kMaxUnitBytes) { // the batch path holds a stream in one unit
        r->finished = true;
        r->status = XLZ_ERR_UNSUPPORTED;
        return XLZ_OK;
    }
    uint64_t cap = 0;
    bool known = false;
    if (r->desc.format == XLZ_FMT_LZMA_ALONE) {
        uint64_t u = xlz_decode_unpack_size(r->src + 5);
        if (u != kUnknownSize) known = true, cap = u;
    } else if (r->desc.format == XLZ_FMT_LZMA_RAW && r->desc.unpack_size != kUnknownSize) {
        known = true;
        cap = r->desc.unpack_size;
    }
    if (!known) cap = std::max<uint64_t>(1u << 16, std::max<uint64_t>((uint64_t)r->src_len * 6, r->delivered * 2));
    std::vector<uint8_t> out;
    for (;;) {
        if (cap > kMaxUnitBytes) cap = kMaxUnitBytes;
        out.resize((size_t)cap);
        xlz_stream_desc d = r->desc;
        d.in = r->src;
        d.in_len = r->src_len;
        d.out = out.data();
        d.out_cap = out.size();
        xlz_result res;
        int st = xlz_decode_batch(r->ctx, &d, 1, &res);
        if (st != XLZ_OK) return st;
        if (res.status == XLZ_ERR_OUT_CAP && !known && cap < kMaxUnitBytes) {
            cap *= 4;
            continue;
        }
        const size_t skip = (size_t)std::min<uint64_t>(r->delivered, res.out_len);
        r->chunk.assign(out.begin() + skip, out.begin() + (size_t)res.out_len);
        r->rd = 0;
        r->delivered = res.out_len;
        r->finished = true;
        r->status = res.status;
        return XLZ_OK;
    }
}

void batcher_loop(Batcher *bt)
{
    std::unique_lock<std::mutex> lk(bt->mu);
    for (;;) {
        bt->cv_work.wait(lk, [&] { return bt->stop || !bt->pending.empty(); });
        if (bt->stop && bt->pending.empty()) return;
        // give concurrent readers a moment to join this launch
        bt->cv_work.wait_for(lk, std::chrono::microseconds(bt->window_us),
                             [&] { return bt->stop || bt->pending.size() >= bt->max_streams; });
        std::vector<xlz_reader *> work;
        while (!bt->pending.empty() && work.size() < bt->max_streams) {
            work.push_back(bt->pending.front());
            bt->pending.pop_front();
        }
        lk.unlock();
        (void)sessions_step(bt->ctx, work); // (every reader carries the status of its own launch: step_status)
        lk.lock();
        bt->n_batches++;
        bt->n_streams += work.size();
        for (xlz_reader *r : work) {
            if (r->step_status != XLZ_OK) r->call_status = r->step_status;
            r->refill_pending = false;
        }
        bt->cv_done.notify_all();
    }
}

// An LZMA2 stream made of many dictionary-reset units (multi-threaded encoders write those; BASELINE
// config 4) read through NewReader2: a session would walk it with ONE wave.  When the whole stream is
// at hand and its headers announce at least eight units, the reader instead decodes runs of whole
// units -- up to kParBytes of output per refill -- through the batch path, where every unit has a
// wave of its own, and serves them in order.  The slice of a run is a raw LZMA2 stream without its
// end byte: "input ended" at exactly the announced sizes is its clean outcome.  Anything else (the
// real decode leaves the headers: malformed streams) hands the reader to the whole-stream path.
constexpr uint64_t kParBytes = 64u << 20;

bool reader_parallel_plan(xlz_reader *r)
{
    r->par_tried = true;
    std::vector<Lz2Unit> lu;
    uint32_t mx = 0;
    scan_lzma2(r->src, r->src_len, lu, mx);
    if (lu.size() < 8 || mx > kMaxLcLpLds) return false;
    r->par_in.assign(1, 0);
    r->par_out.assign(1, 0);
    for (const Lz2Unit &u : lu) {
        if (u.in_start != r->par_in.back() || u.out_start != r->par_out.back()) return false;
        r->par_in.push_back(u.in_start + (uint64_t)u.in_len);
        r->par_out.push_back(u.out_start + u.expect_out);
    }
    return r->par_in.back() == r->src_len;
}

int reader_parallel_step(xlz_reader *r)
{
    const size_t nunits = r->par_in.size() - 1;
    const size_t k0 = r->par_next;
    size_t k1 = k0 + 1;
    while (k1 < nunits && r->par_out[k1 + 1] - r->par_out[k0] <= kParBytes) k1++;
    const bool last = k1 == nunits;
    const uint64_t want_out = r->par_out[k1] - r->par_out[k0], in_len = r->par_in[k1] - r->par_in[k0];
    if (want_out > kMaxUnitBytes || in_len > kMaxUnitBytes) { // (one enormous unit: a session's job)
        r->par = false;
        return XLZ_OK;
    }
    std::vector<uint8_t> out((size_t)want_out + (last ? 65536 : 0) + 1);
    xlz_stream_desc d = r->desc;
    d.in = r->src + r->par_in[k0];
    d.in_len = (size_t)in_len;
    d.out = out.data();
    d.out_cap = out.size() - 1;
    xlz_result res;
    int st = xlz_decode_batch(r->ctx, &d, 1, &res);
    if (st != XLZ_OK) return st;
    r->n_par++;
    const bool clean = last ? (res.status >= 0 && res.out_len == want_out)
                            : (res.status == XLZ_ERR_UNEXPECTED_EOF && res.out_len == want_out && res.in_consumed == in_len);
    if (!clean && !(last && k0 == 0)) { // off the headers: the whole-stream path settles bytes and status
        r->par = false;
        r->whole = true;
        return XLZ_OK;
    }
    out.resize((size_t)res.out_len);
    r->chunk.swap(out);
    r->rd = 0;
    r->delivered += res.out_len;
    r->par_next = k1;
    if (last) {
        r->finished = true;
        r->status = res.status;
    }
    return XLZ_OK;
}

// more decoded bytes into r->chunk (or the end of the stream into r->finished / r->status)
int reader_refill(xlz_reader *r)
{
    r->input_fresh = false;
    if (r->desc.format == XLZ_FMT_LZMA2_RAW && !r->streaming && !r->whole && !r->ss && !r->par_tried)
        r->par = reader_parallel_plan(r);
    if (r->par) {
        int st = reader_parallel_step(r);
        if (st != XLZ_OK || r->par) return st;
        if (r->whole && !r->finished) return reader_whole(r);
        // (a single enormous unit: fall through to a session, which starts at the stream's start -- nothing was delivered yet
        //  unless earlier runs were; then the whole-stream path is the one that can skip)
        if (r->delivered) {
            r->whole = true;
            return reader_whole(r);
        }
    }
    if (!r->ss && !r->whole) {
        int st = XLZ_OK;
        {
            std::lock_guard<std::mutex> lock(r->ctx->mu);
            HIP_TRY(hipSetDevice(r->ctx->device));
            st = session_open(r);
        }
        if (st == XLZ_ERR_UNSUPPORTED)
            r->whole = true;
        else if (st != XLZ_OK)
            return st;
    }
    if (r->whole && r->streaming && !r->in_eof && !r->finished) {
        // a model that does not fit LDS on a FED reader: the whole-stream path needs the whole stream.  Nothing has been
        // dropped yet (xlz_reader_feed only drops behind a session), so ask for the rest and decode once the end is
        // declared -- never decode a partial input as if it were the stream (silent truncation)
        r->need_input = true;
        return XLZ_OK;
    }
    if (!r->whole) {
        Batcher *bt = r->ctx->batcher;
        if (bt) { // the batcher thread steps it together with its contemporaries
            std::unique_lock<std::mutex> lk(bt->mu);
            r->refill_pending = true;
            bt->pending.push_back(r);
            bt->cv_work.notify_one();
            bt->cv_done.wait(lk, [&] { return !r->refill_pending; });
            if (r->call_status != XLZ_OK) return r->call_status;
        } else {
            std::vector<xlz_reader *> one{r};
            int st = sessions_step(r->ctx, one);
            if (st != XLZ_OK) return st;
        }
    }
    if (r->whole && !r->finished) return reader_whole(r);
    return XLZ_OK;
}

void reader_release_device(xlz_reader *r)
{
    if (!r->ss) return;
    std::lock_guard<std::mutex> lock(r->ctx->mu);
    (void)hipSetDevice(r->ctx->device);
    session_free(r->ss);
    r->ss = nullptr;
}

} // namespace

// Streams of >= 4 GiB (input or output) inside xlz_decode_batch: each is a session (the window slides,
// the device counts positions relative to it, state.go:123-129's 64-bit bytesLeft is kept), all of
// them stepped together, every refill copied straight into the caller's buffer.
static int decode_oversize(xlz_ctx *ctx, const xlz_stream_desc *streams, xlz_result *results, const std::vector<size_t> &idx)
{
    std::vector<xlz_reader> rs(idx.size());
    std::vector<uint64_t> produced(idx.size(), 0);
    std::vector<xlz_reader *> active;
    int st = XLZ_OK;
    for (size_t k = 0; k < idx.size(); k++) {
        xlz_reader &r = rs[k];
        r.ctx = ctx;
        r.desc = streams[idx[k]];
        r.src = streams[idx[k]].in;
        r.src_len = streams[idx[k]].in_len;
        int so;
        {
            std::lock_guard<std::mutex> lock(ctx->mu);
            HIP_TRY(hipSetDevice(ctx->device));
            so = session_open(&r);
        }
        if (so == XLZ_OK)
            active.push_back(&r);
        else if (so == XLZ_ERR_DEVICE)
            st = so;
        else
            results[idx[k]].status = XLZ_ERR_UNSUPPORTED; // model too large for LDS AND >= 4 GiB
    }
    while (st == XLZ_OK && !active.empty()) {
        st = sessions_step(ctx, active);
        if (st != XLZ_OK) break;
        std::vector<xlz_reader *> next;
        for (xlz_reader *r : active) {
            const size_t k = (size_t)(r - rs.data());
            const xlz_stream_desc &d = streams[idx[k]];
            xlz_result &res = results[idx[k]];
            if (r->whole) { // (malformed LZMA2 reading across a dictionary reset: needs the whole-stream path)
                res.status = XLZ_ERR_UNSUPPORTED;
                res.out_len = produced[k];
                continue;
            }
            const size_t room = d.out_cap > produced[k] ? (size_t)(d.out_cap - produced[k]) : 0;
            const size_t take = std::min(room, r->chunk.size());
            if (take) memcpy(d.out + produced[k], r->chunk.data(), take);
            const bool overflow = take < r->chunk.size();
            produced[k] += take;
            r->chunk.clear();
            if (overflow || r->finished) {
                res.status = overflow ? XLZ_ERR_OUT_CAP : r->status;
                res.out_len = produced[k];
                res.in_consumed = r->ss->payload_off + r->ss->consumed;
                res.reserved = 0;
                continue;
            }
            next.push_back(r);
        }
        active.swap(next);
    }
    for (xlz_reader &r : rs) reader_release_device(&r);
    return st;
}

// NewReader1, reader1.go:18-24
extern "C" xlz_reader *xlz_new_reader1(xlz_ctx *ctx, const uint8_t *in, size_t in_len, int *err)
{
    int e = XLZ_OK;
    xlz_reader *r = nullptr;
    if (!ctx || (!in && in_len)) {
        e = XLZ_ERR_BAD_ARG;
    } else if (in_len == 0) {
        e = XLZ_ERR_HEADER_EOF;
    } else if (xlz_decode_prop(in[0], nullptr, nullptr, nullptr) != XLZ_OK) {
        e = XLZ_ERR_PROPS;
    } else if (in_len < 13) {
        e = XLZ_ERR_HEADER_EOF;
    } else if ((e = check_rc_init(in + 13, in_len - 13)) == XLZ_OK) {
        r = reader_new(ctx, in, in_len);
        if (!r)
            e = XLZ_ERR_BAD_ARG;
        else
            r->desc.format = XLZ_FMT_LZMA_ALONE;
    }
    if (err) *err = e;
    return r;
}

// NewReader2, reader2.go:26-41: the constructor already runs startChunk for the
// first chunk, so its framing errors are constructor errors.
extern "C" xlz_reader *xlz_new_reader2(xlz_ctx *ctx, const uint8_t *in, size_t in_len, int dict_size, int *err)
{
    int e = XLZ_OK;
    xlz_reader *r = nullptr;
    if (!ctx || (!in && in_len)) {
        e = XLZ_ERR_BAD_ARG;
    } else if (in_len == 0) {
        e = XLZ_ERR_UNEXPECTED_EOF; // reader2.go:104-110
    } else {
        const uint8_t c = in[0];
        const bool lzma_chunk = c >= 0x80;
        const size_t hl = c == 0 || (c >= 3 && c < 0x80) ? 1 : (c < 3 ? 3 : ((c >> 5) >= 6 ? 6 : 5));
        if (in_len < hl) {
            e = XLZ_ERR_UNEXPECTED_EOF; // reader2.go:121-128
        } else if (lzma_chunk) {
            // first LZMA chunk: props come from header[5], which is still 0 when the
            // chunk carries none (reader2.go:146-153)
            const uint8_t props = hl == 6 ? in[5] : 0;
            if (xlz_decode_prop(props, nullptr, nullptr, nullptr) != XLZ_OK) {
                e = XLZ_ERR_PROPS;
            } else {
                size_t comp = (((size_t)in[3] << 8) | in[4]) + 1;
                size_t avail = std::min(comp, in_len - hl);
                e = check_rc_init(in + hl, avail);
            }
        }
        if (e == XLZ_OK) {
            r = reader_new(ctx, in, in_len);
            if (!r) {
                e = XLZ_ERR_BAD_ARG;
            } else {
                r->desc.format = XLZ_FMT_LZMA2_RAW;
                r->desc.dict_size = (uint32_t)dict_size;
            }
        }
    }
    if (err) *err = e;
    return r;
}

// NewLZMADecompressorForSevenZip, reader1.go:32-61
extern "C" xlz_reader *xlz_new_lzma_decompressor_for_sevenzip(xlz_ctx *ctx, const uint8_t *props, size_t props_len,
                                                              uint64_t unpack_size, const uint8_t *const *readers,
                                                              const size_t *reader_lens, size_t n_readers, int *err)
{
    int e = XLZ_OK;
    xlz_reader *r = nullptr;
    if (n_readers != 1) {
        e = XLZ_ERR_NEED_ONE_READER; // :33-35
    } else if (!ctx || !props || props_len < 5 || !readers || !reader_lens || (!readers[0] && reader_lens[0])) {
        e = XLZ_ERR_BAD_ARG; // the reference would panic indexing props[1:5]
    } else if (xlz_decode_prop(props[0], nullptr, nullptr, nullptr) != XLZ_OK) {
        e = XLZ_ERR_PROPS; // :37-40
    } else {
        // the reference hands back the readCloser AND initialize()'s error (:57-60)
        e = check_rc_init(readers[0], reader_lens[0]);
        if (e == XLZ_OK) {
            r = reader_new(ctx, readers[0], reader_lens[0]);
            if (!r) {
                e = XLZ_ERR_BAD_ARG;
            } else {
                r->desc.format = XLZ_FMT_LZMA_RAW;
                r->desc.props = props[0];
                r->desc.dict_size = xlz_decode_dict_size(props + 1);
                r->desc.unpack_size = unpack_size;
                r->is_closer = true;
            }
        }
    }
    if (err) *err = e;
    return r;
}

// NewLZMA2DecompressorForSevenZip, reader2.go:49-75
extern "C" xlz_reader *xlz_new_lzma2_decompressor_for_sevenzip(xlz_ctx *ctx, const uint8_t *props, size_t props_len,
                                                               uint64_t unpack_size, const uint8_t *const *readers,
                                                               const size_t *reader_lens, size_t n_readers, int *err)
{
    (void)unpack_size; // ignored by the reference too (reader2.go:49)
    int e = XLZ_OK;
    xlz_reader *r = nullptr;
    if (n_readers != 1) {
        e = XLZ_ERR_NEED_ONE_READER; // :50-52
    } else if (props_len != 1) {
        e = XLZ_ERR_INSUFFICIENT_PROPS; // :54-56
    } else if (!ctx || !props || !readers || !reader_lens) {
        e = XLZ_ERR_BAD_ARG;
    } else {
        r = xlz_new_reader2(ctx, readers[0], reader_lens[0], (int)xlz_decode_dict_size2(props[0]), &e);
        if (r) r->is_closer = true;
    }
    if (err) *err = e;
    return r;
}

// Reader1.Read / Reader2.Read / readCloser.Read (reader1.go:223-254): drain what is pending, decode
// more when the caller's buffer is not full yet, io.EOF only once everything has been delivered.
extern "C" long xlz_reader_read(xlz_reader *r, uint8_t *p, size_t n, int *err)
{
    int e = XLZ_OK;
    long got = 0;
    if (!r || (!p && n)) {
        e = XLZ_ERR_BAD_ARG;
    } else if (r->closed) {
        e = XLZ_ERR_CLOSED; // readcloser.go:31-33
    } else {
        for (;;) {
            const size_t left = r->chunk.size() - r->rd;
            const size_t k = std::min(left, n - (size_t)got);
            if (k) memcpy(p + got, r->chunk.data() + r->rd, k);
            r->rd += k;
            got += (long)k;
            if ((size_t)got == n) break; // p is full (the reference's Read(p) with len(p) == 0 never returns)
            // the chunk is drained here
            if (r->finished) { // everything produced has been handed over: io.EOF, or the decode error
                e = r->status >= 0 ? XLZ_EOF : r->status;
                break;
            }
            if (r->need_input) { // streaming input: the caller feeds (or declares the end) and reads again
                e = XLZ_NEED_INPUT;
                break;
            }
            if ((e = reader_refill(r)) != XLZ_OK) break;
        }
        if (n == 0 && r->finished && r->rd == r->chunk.size()) e = r->status >= 0 ? XLZ_EOF : r->status;
    }
    if (err) *err = e;
    return got;
}

// readCloser.Close, readcloser.go:16-28
extern "C" int xlz_reader_close(xlz_reader *r)
{
    if (!r) return XLZ_ERR_BAD_ARG;
    if (r->closed) return XLZ_ERR_CLOSED;
    r->closed = true;
    reader_release_device(r);
    r->in.clear();
    r->in.shrink_to_fit();
    r->chunk.clear();
    r->chunk.shrink_to_fit();
    return XLZ_OK;
}

extern "C" void xlz_reader_free(xlz_reader *r)
{
    if (!r) return;
    reader_release_device(r);
    delete r;
}

// (*Reader1).Reset, reader1.go:161-164: state.Reset (every probability back to 1024, state and reps
// to 0) and isEndOfStream = false; window, range coder and input stay.
extern "C" int xlz_reader_reset(xlz_reader *r)
{
    if (!r) return XLZ_ERR_BAD_ARG;
    if (r->closed) return XLZ_ERR_CLOSED;
    if (r->desc.format == XLZ_FMT_LZMA2_RAW || r->whole) return XLZ_ERR_UNSUPPORTED; // a Reader1 method
    r->pending_reset = true;
    r->finished = false;
    return XLZ_OK;
}

// (*Reader1).Reopen, reader1.go:166-176: a NEW compressed stream (raw: no header) continues on the
// same window and model: SetUnpackSize, then rangeDec.Reopen -> Init on the new source, whose
// error is returned as is (io.EOF -> XLZ_ERR_HEADER_EOF, first byte != 0 -> XLZ_ERR_RESULT).
extern "C" int xlz_reader_reopen(xlz_reader *r, const uint8_t *in, size_t in_len, uint64_t unpack_size)
{
    if (!r || (!in && in_len)) return XLZ_ERR_BAD_ARG;
    if (r->closed) return XLZ_ERR_CLOSED;
    if (r->desc.format == XLZ_FMT_LZMA2_RAW || r->whole) return XLZ_ERR_UNSUPPORTED;
    // (a stream that is still being fed is simply abandoned, like the reference's old inStream)
    r->streaming = false; // the new stream is given whole -- unless xlz_reader_expect_more follows: `in` is its first piece
    r->in_eof = false;
    r->input_fresh = true;
    r->in_base = 0;
    r->need_input = false;
    if (!r->ss) { // not started yet: open the session on the current stream's parameters first
        std::lock_guard<std::mutex> lock(r->ctx->mu);
        HIP_TRY(hipSetDevice(r->ctx->device));
        int st = session_open(r);
        if (st != XLZ_OK) return st;
    }
    const int e = check_rc_init(in, in_len);
    r->in.assign(in, in + in_len);
    r->src = r->in.data();
    r->src_len = r->in.size();
    Session *ss = r->ss;
    ss->payload_off = 0;
    ss->in_skip = ss->in_loaded = ss->consumed = 0;
    ss->unit.unpack_size = unpack_size;
    // a known size no longer bounds the window: the new stream appends to it
    ss->win_max = 2 * (size_t)ss->unit.dict_size + kChunk + kWinSlack;
    ss->left = kUnknownSize; // (every refill gets room for a full kChunk again)
    if (ss->in_buf < kInBuf) { // the input window was sized for the first stream
        uint8_t *nc = nullptr;
        std::lock_guard<std::mutex> lock(r->ctx->mu);
        HIP_TRY(hipSetDevice(r->ctx->device));
        HIP_TRY(hipMalloc(&nc, ss->off_in + kInBuf));
        if (hipMemcpy(nc, ss->d_ctl, ss->off_in, hipMemcpyDeviceToDevice) != hipSuccess) {
            (void)hipFree(nc);
            return XLZ_ERR_DEVICE;
        }
        (void)hipFree(ss->d_ctl);
        ss->d_ctl = nc;
        ss->in_buf = kInBuf;
    }
    if (ss->started) {
        r->pending_reopen = true;
    } else { // nothing decoded yet: the first launch simply starts on the new stream
        ss->unit.kind = UNIT_LZMA1;
    }
    r->finished = false;
    r->status = XLZ_OK;
    return e == XLZ_ERR_RC_INIT ? XLZ_ERR_RESULT : e;
}

// Streaming input.  A reader made from the FIRST piece of a stream (at least the header and the five
// range-coder bytes: the constructors check those) is told that more follows; xlz_reader_read then
// returns XLZ_NEED_INPUT whenever the decoder has used up what it was given, and the caller feeds the
// next piece or declares the end.  The host keeps only what the decoder has not consumed yet.
extern "C" int xlz_reader_expect_more(xlz_reader *r)
{
    if (!r) return XLZ_ERR_BAD_ARG;
    if (r->closed) return XLZ_ERR_CLOSED;
    if (!r->input_fresh || r->whole) return XLZ_ERR_BAD_ARG; // before the first read of this input (constructor or Reopen)
    r->streaming = true;
    r->in_eof = false;
    return XLZ_OK;
}

extern "C" int xlz_reader_feed(xlz_reader *r, const uint8_t *data, size_t n)
{
    if (!r || (!data && n)) return XLZ_ERR_BAD_ARG;
    if (r->closed) return XLZ_ERR_CLOSED;
    if (!r->streaming || r->in_eof) return XLZ_ERR_BAD_ARG;
    if (r->ss) { // drop what the decoder is done with (the device window restarts at consumed & ~255)
        const uint64_t keep_from = r->ss->payload_off + (r->ss->consumed & ~(uint64_t)255);
        if (keep_from > r->in_base) {
            const size_t drop = (size_t)std::min<uint64_t>(keep_from - r->in_base, r->in.size());
            r->in.erase(r->in.begin(), r->in.begin() + drop);
            r->in_base += drop;
        }
    }
    r->in.insert(r->in.end(), data, data + n);
    r->src = r->in.data();
    r->src_len = r->in.size();
    r->need_input = false;
    return XLZ_OK;
}

extern "C" int xlz_reader_feed_eof(xlz_reader *r)
{
    if (!r) return XLZ_ERR_BAD_ARG;
    if (r->closed) return XLZ_ERR_CLOSED;
    if (!r->streaming) return XLZ_ERR_BAD_ARG;
    r->in_eof = true;
    r->need_input = false;
    return XLZ_OK;
}

extern "C" int xlz_reader_memory(const xlz_reader *r, uint64_t *window_bytes, uint64_t *image_bytes)
{
    if (!r) return XLZ_ERR_BAD_ARG;
    if (window_bytes) *window_bytes = r->ss ? r->ss->win_cap : 0;
    if (image_bytes) *image_bytes = (r->ss && r->ss->d_shadow) ? r->ss->unit.dict_size : 0;
    return XLZ_OK;
}

extern "C" int xlz_reader_stats(const xlz_reader *r, uint64_t *refills, uint64_t *whole_decodes, uint64_t *in_uploaded)
{
    if (!r) return XLZ_ERR_BAD_ARG;
    if (refills) *refills = r->n_refills + r->n_par;
    if (whole_decodes) *whole_decodes = r->n_whole;
    if (in_uploaded) *in_uploaded = r->ss ? r->ss->in_loaded : 0;
    return XLZ_OK;
}

static void batcher_shutdown(Batcher *bt)
{
    {
        std::lock_guard<std::mutex> lk(bt->mu);
        bt->stop = true;
        bt->cv_work.notify_all();
    }
    bt->th.join();
    delete bt;
}

extern "C" int xlz_ctx_enable_batching(xlz_ctx *ctx, uint32_t window_us, uint32_t max_streams)
{
    if (!ctx || ctx->batcher) return XLZ_ERR_BAD_ARG;
    Batcher *bt = new (std::nothrow) Batcher;
    if (!bt) return XLZ_ERR_BAD_ARG;
    bt->ctx = ctx;
    bt->window_us = window_us;
    bt->max_streams = max_streams ? max_streams : 1;
    bt->th = std::thread(batcher_loop, bt);
    ctx->batcher = bt;
    return XLZ_OK;
}

extern "C" int xlz_ctx_batching_stats(xlz_ctx *ctx, uint64_t *batches, uint64_t *streams)
{
    if (!ctx || !ctx->batcher) return XLZ_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(ctx->batcher->mu);
    if (batches) *batches = ctx->batcher->n_batches;
    if (streams) *streams = ctx->batcher->n_streams;
    return XLZ_OK;
}
