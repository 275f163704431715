// xlz_kernel.hip -- gfx950 (MI355X) decode kernel for batched LZMA / LZMA2.
//
// Mapping: ONE WAVE PER UNIT.  A unit is one independent LZMA1
// stream or one run of LZMA2 chunks.  The adaptive range decoder of
// decompress.go:8-1136 is a strictly serial dependent chain, so a unit cannot
// use lanes for its decisions; instead every lane of the wave executes the same
// wave-uniform code (range/code/state live in SGPRs, branches are scalar, no
// divergence), the unit's whole probability model (state.go:3-27) lives in LDS,
// and the 64 lanes are used where the work IS parallel: model initialisation,
// match copies, the compressed-input window (64 lanes x 4 bytes in one VGPR),
// and the look-ahead of a bit-tree node's children (a per-lane LDS gather).
// One single-wave workgroup owns one LDS model; 160 KiB / 15.7 KiB = 10 units
// per CU run concurrently and a persistent grid pulls units from an atomic
// queue (heaviest first).
//
// Two decoders of one packet exist (lzma_packet_checked / lzma_packet_fast).
// The fast one is hand-scheduled GCN assembly and assumes what lzma_run has
// verified for it: >= 32 input bytes, >= 336 bytes of output room and bytesLeft.
// The checked one is plain C++ with every end-of-input / capacity test of the
// reference and runs only within a few bytes of a stream's or chunk's end.
//
// The sliding window (window.go) is the unit's own flat output range in HBM:
// distances are bounded by dictSize, so `out[pos - dist]` is the circular
// window's byte, and bytes "before the start" read as 0 exactly like the
// reference's zero-filled, not-yet-full window.
//
// Measured costs on MI355X that shaped this file (tools/ubench): dependent SALU
// or VALU op 4 cycles; LDS read -> use 60; scalar branch 15 not taken / 20
// taken; a VCC branch 45; one SIMD issues <= 1 SALU op per 4 cycles.
//
// file:line citations are into the reference repository (kulaginds/lzma).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>

#include "xlz_format.h"

namespace xlz {

#define RFL(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
__device__ __forceinline__ uint64_t rfl64(uint64_t v)
{
    return ((uint64_t)RFL((uint32_t)(v >> 32)) << 32) | RFL((uint32_t)v);
}

constexpr uint32_t kTop = 1u << 24;             // types.go:27
constexpr uint32_t kBitModelBits = 11;          // types.go:12
constexpr uint32_t kMoveBits = 5;               // types.go:13
constexpr uint32_t kProbInitPair = 0x04000400u; // two probs of 1024 (types.go:14)
constexpr uint32_t kEndPosModelIndex = 14;      // types.go:22
constexpr uint32_t kNumAlignBits = 4;           // types.go:20
constexpr uint32_t kMatchMinLen = 2;            // types.go:24

constexpr uint32_t kInWindow = 256;   // bytes of compressed input held in one VGPR (64 lanes x 4)
constexpr uint32_t kFastInput = 32;   // >= lzmaRequiredInputMax = 20 (types.go:38), with slack
// The fast loop runs while at least kFastOutput bytes of output room AND of bytesLeft remain: the copies it does itself are
// shorter than 64 bytes and store one whole 64-lane row (pos + 64 + 63 stays inside the unit's own output range).  A copy it
// hands back (FX_COPY: 64 bytes and more, overlapping, ...) is done with whole rows only while kFastCopyRoom = maxMatchLen 273
// (types.go:46) + 63 bytes remain, and otherwise the way the checked path does it (truncated to bytesLeft, clamped to the room:
// decompress.go:657-668).  Round 3 measured the smaller margin (+0.6 % on 64 KiB streams: 336 bytes of checked path cost such a
// stream 0.9 % of its time) and fuzzed it; round 4 adopted it together with the other instruction-count changes.
constexpr uint32_t kFastOutput = 128;
constexpr uint32_t kFastCopyRoom = 336;

constexpr uint32_t kLzma1InputMargin = 64;     // UNIT_F_MORE_INPUT: ask for more input below this (a packet needs <= 20)
constexpr uint32_t kLzma2InputMargin = 67584;  // ... LZMA2: a whole chunk (6-byte header + 64 KiB) must be in the window

enum : int { RUN_END = 0, RUN_INPUT_EOF = 1, RUN_ERR_RESULT = 2, RUN_OUT_CAP = 3, RUN_CONTINUE = 4, RUN_PAUSE = 5 };

// Everything a unit carries between packets / chunks.  Except `vin` all members are
// wave-uniform; after inlining they live in SGPRs.
struct Dec {
    // range decoder (range_decoder.go:7-13)
    uint32_t range, code;
    // compressed input.  Positions are relative to the unit's payload rounded down to 4
    // bytes.  `vin` holds the 256 bytes at [win0, win0 + 256): lane i = dword i.
    const uint32_t *inw; // payload base, 4-byte aligned
    uint32_t vin;        // per lane
    uint32_t win0;       // window start (multiple of 4)
    uint32_t arel;       // next byte to read, relative to win0
    uint32_t cur;        // unread bytes of dword arel >> 2, next byte in bits 7:0
    uint32_t abase;      // position of the unit's first payload byte (0..3)
    uint32_t aend;       // first position that may NOT be read (limitedByteReader, bytereader.go:7-28)
    // LZMA state (state.go:28-45)
    uint32_t state, rep0, rep1, rep2, rep3;
    uint32_t lc, lp_mask, pos_mask;
    uint32_t size_defined; // 0 / 1 (kept as a word: it is saved and restored)
    uint64_t bytes_left; // state.go:123-129 keeps 64 bits
    // window (window.go:8-16) over the flat output
    uint32_t pos;        // bytes of output produced by this unit
    uint32_t wbase;      // output offset of the last dictionary reset (0 for LZMA1)
    uint32_t wpos;       // the reference's wrapped window.pos
    uint32_t dict_size;  // window.size
    uint32_t out_cap;
    uint32_t prev_byte;  // byte at distance 1 (0 while the window is empty)
    uint32_t match_byte; // byte at distance rep0+1, valid right after a match / rep
    uint32_t stale;      // a copy reached in front of the current dictionary epoch and no epoch table is there
    // pull readers: the unit stops at the first packet / chunk boundary with pos >= pause_at
    uint32_t pause_at;
    uint32_t in_margin;  // LZMA1 + UNIT_F_MORE_INPUT: pause for more input below this many bytes, else 0
    uint32_t need_input; // that pause happened
    uint32_t in_base;    // input bytes consumed before the unit's input window (UNIT_F_RESUME)
    bool epoch0_clean;   // nothing precedes the unit's first dictionary epoch: bytes in front of it are 0
    // exact launches: earlier dictionary epochs a copy may still read (window.Reset keeps the buffer)
    Epoch *epochs;
    uint32_t n_epochs;
    // pull readers (sessions): the reference's window BUFFER as the last dictionary reset left it (dict_size bytes,
    // circular index -> the last byte any earlier epoch wrote there, 0 if none).  A session's output window slides, so
    // the bytes of earlier epochs are not in `out` any more; the walker copies every epoch that ends into this image
    // (shadow_update).  nullptr for batch units (their whole output is in the arena: epoch table).
    uint8_t *shadow;
    uint8_t *dump;       // 64 bytes per lane-row that predicated-off lanes store to
    uint32_t work_end;   // input position where the unit ends (remaining work = work_end - in_pos)
    uint32_t *prio_slot; // this wave's word in LaunchParams.prio_tab (rank_priority)
};

__device__ __forceinline__ uint32_t umod_small(uint32_t i, uint32_t d)
{
    // i < 1024, 1 <= d < 1024: one float step plus a correction is exact
    uint32_t q = (uint32_t)((float)i * __frcp_rn((float)d));
    int32_t r = (int32_t)(i - q * d);
    if (r < 0) r += (int32_t)d;
    if ((uint32_t)r >= d) r -= (int32_t)d;
    return (uint32_t)r;
}

// The byte the reference's window holds at circular index c (0 <= c < dict_size) while the
// current dictionary epoch has not written there yet: window.Reset (window.go:135-140) keeps the
// buffer, so it is the last byte any EARLIER epoch of the stream wrote at c, and 0 if none did
// (zero-filled at allocation, window.go:18-29).  The epochs still visible are kept as a stack
// with strictly decreasing lengths (an epoch hides every earlier one that was not longer), so
// the first entry from the top that is long enough holds the byte.  Exact launches only.
__device__ __forceinline__ uint32_t stale_byte(const uint8_t *__restrict__ out, const Dec &d, uint32_t c, bool want)
{
    if (d.shadow) { // sessions: the uncleared window buffer itself
        const bool ok = want && c < d.dict_size;
        const uint32_t v = d.shadow[ok ? c : 0u];
        return ok ? v : 0u;
    }
    uint32_t b = 0;
    bool found = !want;
    for (uint32_t k = d.n_epochs; k-- > 0;) { // wave-uniform trip count, selects instead of branches
        const uint32_t start = RFL(d.epochs[k].start), len = RFL(d.epochs[k].len);
        const bool hit = !found && len > c;
        uint32_t o = start + c;
        if (len > d.dict_size) // the epoch wrapped: the LAST write to index c
            o += ((len - 1 - c) / d.dict_size) * d.dict_size;
        const uint32_t v = out[hit ? o : start];
        b = hit ? v : b;
        found = found || hit;
    }
    return b;
}

// sessions: the dictionary epoch that ends at a reset goes into the window image.  The byte k positions behind the
// write position has circular index window.pos - 1 - k (mod size; d.wpos IS the reference's wrapped window.pos, so this
// holds even when the host has slid the epoch's start out of the buffer); the last min(epoch, dictSize) bytes are always
// still in `out` (the host keeps dictSize bytes of history).
// The bytes are contiguous in `out` and land on the circular range that ends at window.pos: two plain copies at most,
// moved 16 bytes per lane like a stored chunk (stored_copy; round 3 moved one byte per lane and step).
__device__ __forceinline__ void stored_copy(const uint8_t *__restrict__ src, uint8_t *__restrict__ out, uint32_t pos,
                                            uint32_t n, uint32_t lane);
__device__ __forceinline__ void shadow_update(Dec &d, const uint8_t *__restrict__ out, uint32_t lane)
{
    const uint32_t n = min(d.pos - d.wbase, d.dict_size);
    const uint32_t start = d.wpos >= n ? d.wpos - n : d.wpos + d.dict_size - n; // circular index of out[pos - n]
    const uint32_t first = min(n, d.dict_size - start);
    const uint8_t *src = out + (d.pos - n);
    stored_copy(src, d.shadow, start, first, lane);
    if (n > first) stored_copy(src + first, d.shadow, 0, n - first, lane);
}

// dictionary reset (window.Reset): the epoch [wbase, pos) ends; remember it for stale reads
__device__ __forceinline__ bool epoch_push(Dec &d)
{
    if (!d.epochs) return true;
    const uint32_t len = d.pos - d.wbase;
    if (len == 0) return true;
    while (d.n_epochs && RFL(d.epochs[d.n_epochs - 1].len) <= len) d.n_epochs--;
    if (d.n_epochs == kMaxEpochs) return false;
    Epoch e;
    e.start = d.wbase;
    e.len = len;
    d.epochs[d.n_epochs++] = e; // every lane stores the same 8 bytes
    return true;
}

// window.CopyMatch when the copy reaches in front of the current dictionary epoch and the launch
// has an epoch table: every byte is the reference's (window.go:55-87 over the uncleared circular
// buffer).  Rare (malformed LZMA2 streams), so plain and slow; also yields prevByte / matchByte.
__device__ __forceinline__ void wave_copy_exact(uint8_t *__restrict__ out, Dec &d, uint32_t dist, uint32_t len,
                                                uint32_t lane)
{
    const uint32_t pos = d.pos;
    const uint64_t lo = (uint64_t)dist + d.wbase;
    const uint32_t fill = pos - d.wbase; // bytes of the current epoch; the window is not full here
    const bool wrap = dist <= len;
    for (uint32_t base = 0; base <= len; base += kWave) { // i == len too: the next packet's matchByte
        const uint32_t i = min(base + lane, len);
        uint32_t j = i;
        if (wrap) j = umod_small(i, dist); // byte i repeats byte i mod dist (i < 1024, dist <= 273 here)
        const uint64_t vs = (uint64_t)pos + j;
        const bool ok = vs >= lo;
        const uint32_t raw = out[ok ? vs - dist : 0];
        // circular index of the source: window.pos + j - dist + size (window.go:57-61), window.pos = fill
        const uint32_t sb = stale_byte(out, d, fill + j + d.dict_size - dist, !ok);
        const uint32_t b = ok ? raw : sb;
        uint8_t *dst = (base + lane) < len ? out + ((uint64_t)pos + i) : d.dump + lane;
        *dst = (uint8_t)b;
        if (len - 1 >= base && len - 1 < base + kWave) d.prev_byte = (uint32_t)__builtin_amdgcn_readlane((int)b, len - 1 - base);
        if (len < base + kWave) d.match_byte = (uint32_t)__builtin_amdgcn_readlane((int)b, len - base);
    }
}

// window.CopyMatch (window.go:55-87) over the flat output, all 64 lanes.
// Output byte i of the copy is the byte at virtual index  pos - dist + (i mod dist):
// for i >= dist that is the replication the reference's byte loop produces.  The
// same pass also fetches the byte at i == len, which is the next packet's
// matchByte (GetByte(rep0+1), decompress.go:60), and byte len-1, the next
// prevByte (decompress.go:52) -- so literals never read the window from memory.
//
// NO lane-dependent branch anywhere: hipcc structurizes every enclosing loop as
// divergent as soon as one divergent branch sits inside it, which drags the whole
// range-decoder state into VGPRs under exec masks.  Predication is done with
// selects instead: a lane without a valid source loads out[0] and selects 0; ALL
// 64 lanes store, so bytes [pos+len, pos+64) receive scratch values -- they are
// not-yet-produced output (or the 64-byte pad behind the unit's region) and are
// overwritten by later packets before anything can read them.
//
// PRECISE = true (checked path, i.e. near the end of a unit's output range, where the next
// bytes may belong to a neighbouring unit of the same LZMA2 stream): lanes >= len store to the
// launch's dump row instead of touching bytes past the copy.
//
// A source in front of the current dictionary epoch reads as 0 -- exact for the first epoch of a
// stream (zero-filled window, window.go:18-29); later epochs can see the previous epoch's bytes
// in the reference (window.go:135-140): with an epoch table (exact launch) wave_copy_exact
// reproduces them, without one the unit is flagged (AUX_STALE) and the host runs such a launch.
template <bool PRECISE>
__device__ __forceinline__ void wave_copy(uint8_t *__restrict__ out, Dec &d, uint32_t dist, uint32_t len,
                                          uint32_t lane)
{
    const uint32_t pos = d.pos;
    const uint64_t lo = (uint64_t)dist + d.wbase; // virtual index >= wbase  <=>  pos + j >= lo
    const bool wrap = dist <= len;                // some i in [0, len] needs i mod dist
    if ((uint64_t)pos < lo && !(d.wbase == 0 && d.epoch0_clean)) {
        if (d.epochs || d.shadow) {
            wave_copy_exact(out, d, dist, len, lane);
            return;
        }
        d.stale = 1;
    }
    if (len < kWave) {
        uint32_t j = lane;
        if (wrap) j = umod_small(lane, dist);
        if (PRECISE) j = lane <= len ? j : 0u; // never look past the copy
        const uint64_t vs = (uint64_t)pos + j;
        const bool ok = vs >= lo;
        const uint64_t sa = ok ? vs - dist : 0;
        const uint32_t raw = out[sa];
        const uint32_t b = ok ? raw : 0u;
        if (PRECISE) {
            uint8_t *dst = lane < len ? out + ((uint64_t)pos + lane) : d.dump + lane;
            *dst = (uint8_t)b;
        } else {
            out[pos + lane] = (uint8_t)b;
        }
        d.prev_byte = (uint32_t)__builtin_amdgcn_readlane((int)b, len - 1);
        d.match_byte = (uint32_t)__builtin_amdgcn_readlane((int)b, len);
    } else {
        for (uint32_t base = 0; base < len; base += kWave) {
            const uint32_t i = base + lane;
            uint32_t j = i;
            if (wrap) j = umod_small(i, dist);
            if (PRECISE) j = i < len ? j : 0u;
            const uint64_t vs = (uint64_t)pos + j;
            const bool ok = vs >= lo;
            const uint64_t sa = ok ? vs - dist : 0;
            const uint32_t raw = out[sa];
            const uint32_t b = ok ? raw : 0u;
            if (PRECISE) {
                uint8_t *dst = i < len ? out + ((uint64_t)pos + i) : d.dump + lane;
                *dst = (uint8_t)b;
            } else {
                out[pos + i] = (uint8_t)b;
            }
        }
        // bytes len-1 and len of the copy, re-read (long matches are rare)
        const uint32_t i = len - 1 + (lane & 1); // lanes 0 and 1 matter
        uint32_t j = i;
        if (wrap) j = umod_small(i, dist);
        const uint64_t vs = (uint64_t)pos + j;
        const bool ok = vs >= lo;
        uint32_t b = out[ok ? vs - dist : 0];
        b = ok ? b : 0u;
        d.prev_byte = (uint32_t)__builtin_amdgcn_readlane((int)b, 0);
        d.match_byte = (uint32_t)__builtin_amdgcn_readlane((int)b, 1);
    }
}

// ---- compressed input ---------------------------------------------------------
// (re)load the 256-byte window so that it starts at the dword holding position `p`
__device__ __forceinline__ void in_window(Dec &d, uint32_t p, uint32_t lane)
{
    d.win0 = p & ~3u;
    d.arel = p & 3u;
    d.vin = d.inw[(d.win0 >> 2) + lane]; // one coalesced 256-byte load
    d.cur = (uint32_t)__builtin_amdgcn_readlane((int)d.vin, 0) >> (8 * d.arel);
}

__device__ __forceinline__ void in_open(Dec &d, const uint8_t *arena, uint64_t off, uint32_t avail, uint32_t lane)
{
    const uint64_t a = (uint64_t)arena + off;
    d.inw = (const uint32_t *)(a & ~(uint64_t)3);
    d.abase = (uint32_t)(a & 3);
    d.aend = d.abase + avail;
    in_window(d, d.abase, lane);
}

__device__ __forceinline__ uint32_t in_pos(const Dec &d) { return d.win0 + d.arel; }

// one byte from the window.  The caller has checked in_pos < aend where that matters and
// lzma_run keeps >= kFastInput bytes of window ahead of every packet.
#define IN_BYTE(B)                                                                                \
    do {                                                                                          \
        (B) = d.cur & 0xFFu;                                                                      \
        d.cur >>= 8;                                                                              \
        d.arel++;                                                                                 \
        if ((d.arel & 3u) == 0) d.cur = (uint32_t)__builtin_amdgcn_readlane((int)d.vin, d.arel >> 2); \
    } while (0)

// ================================================================================
//  CHECKED path: plain C++, every test of the reference.  Runs near stream ends.
// ================================================================================
// The binary decision (decompress.go:26-43,176-190; range_decoder.go:57-98).  The
// probability update  p - ((p - k) >>a 5)  with k = 2017 for bit 0 and k = 0 for bit 1
// equals the reference's  p + ((2048 - p) >> 5)  /  p - (p >> 5)  for every 11-bit p.
__device__ __forceinline__ uint32_t rc_core(uint32_t &range, uint32_t &code, uint32_t &p)
{
    const uint32_t bound = (range >> kBitModelBits) * p;
    const bool z = code < bound;
    range = z ? bound : range - bound;
    code = z ? code : code - bound;
    p = p - (uint32_t)((int32_t)(p - (z ? 2017u : 0u)) >> kMoveBits);
    return z ? 0u : 1u;
}

#define NORMALIZE()                                                                               \
    do {                                                                                          \
        if (d.range < kTop) {                                                                     \
            uint32_t nb_;                                                                         \
            if (in_pos(d) == d.aend) return RUN_INPUT_EOF; /* decompress.go:35-38 */              \
            IN_BYTE(nb_);                                                                         \
            d.range <<= 8;                                                                        \
            d.code = (d.code << 8) | nb_;                                                         \
        }                                                                                         \
    } while (0)

#define BIT_NN_IN(ARR, IDX, BITV)                                                                 \
    do {                                                                                          \
        const uint32_t i_ = (IDX);                                                                \
        uint32_t p_ = RFL((ARR)[i_]);                                                             \
        (BITV) = rc_core(d.range, d.code, p_);                                                    \
        (ARR)[i_] = (uint16_t)p_; /* all 64 lanes, same address, same value */                    \
    } while (0)
#define BIT_NN(IDX, BITV) BIT_NN_IN(probs, IDX, BITV)

#define BIT_IN(ARR, IDX, B)                                                                       \
    do {                                                                                          \
        BIT_NN_IN(ARR, IDX, B);                                                                   \
        NORMALIZE();                                                                              \
    } while (0)
#define BIT(IDX, B) BIT_IN(probs, IDX, B)

// Where a tree node lives.  The reference indexes a bit tree by m = 1 b0 b1 .. (decided bits after a
// leading 1).  This build stores node m at tree_slot(m): the same leading 1 followed by the
// COMPLEMENTED bits, because the fast path's index step is `s_addc m, m, m` with SCC = (code < bound)
// = !bit (xlz_fastpath.inc).  A permutation inside every tree level; every table starts out as
// kProbInit everywhere, so nothing else changes.
__device__ __forceinline__ uint32_t tree_slot(uint32_t m, uint32_t level) { return m ^ ((1u << level) - 1); }

// forward bit tree (bit_tree_decoder.go:18-40): M keeps the leading 1
#define TREE_IN(ARR, BASE, NB, M)                                                                 \
    do {                                                                                          \
        (M) = 1;                                                                                  \
        _Pragma("unroll 1") for (uint32_t k_ = 0; k_ < (NB); k_++)                                \
        {                                                                                         \
            uint32_t tb_;                                                                         \
            BIT_IN(ARR, (BASE) + tree_slot((M), k_), tb_);                                        \
            (M) = ((M) << 1) | tb_;                                                               \
        }                                                                                         \
    } while (0)
#define TREE(BASE, NB, M) TREE_IN(probs, BASE, NB, M)

// reverse bit tree (bit_tree_decoder.go:42-70)
#define RTREE(BASE, NB, SYM)                                                                      \
    do {                                                                                          \
        uint32_t m_ = 1;                                                                          \
        (SYM) = 0;                                                                                \
        _Pragma("unroll 1") for (uint32_t k_ = 0; k_ < (NB); k_++)                                \
        {                                                                                         \
            uint32_t tb_;                                                                         \
            BIT((BASE) + tree_slot(m_, k_), tb_);                                                 \
            m_ = (m_ << 1) | tb_;                                                                 \
            (SYM) |= tb_ << k_;                                                                   \
        }                                                                                         \
    } while (0)

// lenDecoder.Decode (len_decoder.go:34-60; decompress.go:218-429,870-1123)
// HIGH_ARR / HIGH_BASE: where the coder's high tree lives (len: LDS model; rep-len: the model's HBM part, xlz_format.h)
#define LEN_DECODE(LBASE, HIGH_ARR, HIGH_BASE, LEN)                                               \
    do {                                                                                          \
        uint32_t c_, m__;                                                                         \
        BIT((LBASE) + L::LEN_CHOICE, c_);                                                         \
        if (c_ == 0) {                                                                            \
            TREE((LBASE) + L::LEN_LOW + (pos_state << 3), 3, m__);                                \
            (LEN) = m__ - 8;                                                                      \
        } else {                                                                                  \
            BIT((LBASE) + L::LEN_CHOICE2, c_);                                                    \
            if (c_ == 0) {                                                                        \
                TREE((LBASE) + L::LEN_MID + (pos_state << 3), 3, m__);                            \
                (LEN) = m__;                                                                      \
            } else {                                                                              \
                TREE_IN(HIGH_ARR, HIGH_BASE, 8, m__);                                             \
                (LEN) = 16 + m__ - 256;                                                           \
            }                                                                                     \
        }                                                                                         \
    } while (0)

// state.go:153-187
__device__ __forceinline__ uint32_t upd_literal(uint32_t s) { return s < 4 ? 0 : (s < 10 ? s - 3 : s - 6); }

// distance validity, decompress.go:651-653 + window.CheckDistance (window.go:89-91)
__device__ __forceinline__ bool bad_distance(const Dec &d)
{
    const bool is_full = (d.pos - d.wbase) >= d.dict_size;
    return d.rep0 >= d.dict_size || !(is_full || d.rep0 <= d.wpos);
}

// ONE packet of (*Reader1).decompress (one iteration of the loop at decompress.go:13),
// every mutation in the reference's order, every test of the reference present.
// L = the model's layout (xlz_format.h: ModelLayout): the table bases and the room for posStates
template <class L>
__device__ __forceinline__ int lzma_packet_checked(Dec &d, uint16_t *probs, uint16_t *__restrict__ mprobs,
                                                   uint8_t *__restrict__ out, uint32_t lane)
{
    uint32_t bit, length;

    const uint32_t pos_state = d.wpos & d.pos_mask;              // :22
    const uint32_t state2 = (d.state << L::kPosBits) + pos_state; // :23 (the table has room for 2^kPosBits posStates)

    BIT(L::P_IS_MATCH + state2, bit); // :25-43,176-190
    if (bit == 0) {
        // ---- literal, decompress.go:44-175 ----
        if (d.size_defined && d.bytes_left == 0) return RUN_ERR_RESULT; // :45-47
        const uint32_t lit_state = ((d.wpos & d.lp_mask) << d.lc) + (d.prev_byte >> (8 - d.lc)); // :56
        const uint32_t lbase = L::P_LIT + kLitPlain * lit_state;                                  // :57
        uint32_t symbol = 1;
        if (d.state >= 7) { // matched literal :59-114: probs[((1 + matchBit) << 8) + symbol] of the
                            // reference's table = mprobs[(matchBit << 8) + symbol] of this state
            uint16_t *mp = mprobs + kRepHigh + kLitMatched * lit_state;
            uint32_t mb = d.match_byte;
            uint32_t level = 0;
            do {
                const uint32_t match_bit = (mb >> 7) & 1;
                mb <<= 1;
                uint16_t *pp = mp + (match_bit << 8) + tree_slot(symbol, level++);
                uint32_t p_ = RFL(*pp);
                bit = rc_core(d.range, d.code, p_);
                *pp = (uint16_t)p_;
                NORMALIZE();
                symbol = (symbol << 1) | bit;
                if (match_bit != bit) break;
            } while (symbol < 0x100);
        }
        for (uint32_t level = 31 - __clz(symbol); symbol < 0x100; level++) { // :127-166
            BIT(lbase + tree_slot(symbol, level), bit);
            symbol = (symbol << 1) | bit;
        }
        symbol &= 0xFF;
        if (d.pos >= d.out_cap) return RUN_OUT_CAP;
        out[d.pos] = (uint8_t)symbol; // window.PutByte :168 (all lanes, same byte, same address)
        d.pos++;
        if (++d.wpos >= d.dict_size) d.wpos -= d.dict_size; // window.go:38-41
        d.prev_byte = RFL(symbol); // (the readfirstlane keeps prev_byte's phi web scalar, see sv_store)
        d.state = upd_literal(d.state); // :171
        d.bytes_left--;                 // :172 (wraps harmlessly when the size is undefined)
        return RUN_CONTINUE;
    }

    BIT(L::P_IS_REP + d.state, bit); // :195-213,669-683
    if (bit == 0) {
        // ---- simple match, :215-668 ----
        d.rep3 = d.rep2;
        d.rep2 = d.rep1;
        d.rep1 = d.rep0; // :216
        LEN_DECODE(L::P_LEN, probs, L::P_LEN + L::LEN_HIGH, length);
        d.state = d.state < 7 ? 7 : 10; // stateUpdateMatch :431
        const uint32_t len_state = length > 3 ? 3 : length;
        uint32_t pos_slot;
        TREE(L::P_POS_SLOT + (len_state << 6), 6, pos_slot); // :441-486
        pos_slot -= 64;
        if (pos_slot < 4) {
            d.rep0 = pos_slot; // :488-489
        } else {
            const uint32_t nbits = (pos_slot >> 1) - 1;
            uint32_t dist = (2 | (pos_slot & 1)) << nbits; // :491-492
            uint32_t sym;
            if (pos_slot < kEndPosModelIndex) {
                RTREE(L::P_POS_DEC + dist - pos_slot, nbits, sym); // :495-546
                d.rep0 = dist + sym;
            } else {
                uint32_t res = 0; // DecodeDirectBits :549-577
                _Pragma("unroll 1") for (uint32_t n = nbits - kNumAlignBits; n > 0; n--)
                {
                    d.range >>= 1;
                    d.code -= d.range;
                    const uint32_t t = 0u - (d.code >> 31);
                    d.code += d.range & t;
                    res = (res << 1) + (t + 1);
                    NORMALIZE();
                }
                dist += res << kNumAlignBits;
                RTREE(L::P_ALIGN, kNumAlignBits, sym); // :579-625
                d.rep0 = dist + sym;                // :627-628
            }
        }
        if (d.rep0 == 0xFFFFFFFFu) { // end marker :633-645
            if (d.code == 0) {
                if (d.size_defined && d.bytes_left > 0) return RUN_ERR_RESULT;
                return RUN_END;
            }
            return RUN_ERR_RESULT;
        }
        if (d.size_defined && d.bytes_left == 0) return RUN_ERR_RESULT; // :647-649
        if (bad_distance(d)) return RUN_ERR_RESULT;                     // :651-653
        length += kMatchMinLen;                                         // :656
    } else {
        // ---- rep match, :685-1123 ----
        if (d.size_defined && d.bytes_left == 0) return RUN_ERR_RESULT; // :686-688
        if (d.pos == d.wbase) return RUN_ERR_RESULT;                    // window.IsEmpty :690-692
        BIT(L::P_IS_REP_G0 + d.state, bit);                                // :694-772
        if (bit == 0) {
            BIT(L::P_IS_REP0_LONG + state2, bit); // :715-756
            if (bit == 0) {                    // short rep :735-739
                d.state = d.state < 7 ? 9 : 11;
                if (d.pos >= d.out_cap) return RUN_OUT_CAP;
                uint32_t dist = d.rep0 + 1;
                if (dist == 0) dist = d.dict_size;
                wave_copy<true>(out, d, dist, 1, lane);
                d.pos++;
                if (++d.wpos >= d.dict_size) d.wpos -= d.dict_size;
                d.bytes_left--;
                return RUN_CONTINUE;
            }
        } else {
            uint32_t dist;
            BIT_NN(L::P_IS_REP_G1 + d.state, bit); // :777-813
            if (bit == 0) {
                dist = d.rep1;
                d.rep1 = d.rep0;
                d.rep0 = dist; // rotated before the normalise (:785-798)
                NORMALIZE();
            } else {
                NORMALIZE();
                BIT_NN(L::P_IS_REP_G2 + d.state, bit); // :816-861
                if (bit == 0) {
                    dist = d.rep2;
                    d.rep2 = d.rep1;
                } else {
                    dist = d.rep3;
                    d.rep3 = d.rep2;
                    d.rep2 = d.rep1;
                }
                d.rep1 = d.rep0;
                d.rep0 = dist;
                NORMALIZE();
            }
        }
        LEN_DECODE(L::P_REP_LEN, mprobs, 0u, length);
        d.state = d.state < 7 ? 8 : 11; // stateUpdateRep :933,1027,1103
        length += kMatchMinLen;
    }

    // window.CopyMatch + size bookkeeping, :657-668, 936-947, 1030-1041, 1106-1117
    {
        bool truncated = false, overflow = false;
        if (d.size_defined && (uint32_t)d.bytes_left < length) { // uint32(s.bytesLeft) < length, decompress.go:657
            length = (uint32_t)d.bytes_left;
            truncated = true;
        }
        if (length > d.out_cap - d.pos) {
            length = d.out_cap - d.pos;
            overflow = true;
        }
        uint32_t dist = d.rep0 + 1;
        if (dist == 0) dist = d.dict_size; // CopyMatch(0, n) re-reads the slot being written
        if (length > 0) wave_copy<true>(out, d, dist, length, lane);
        d.pos += length;
        d.wpos += length;
        if (d.wpos >= d.dict_size) d.wpos -= d.dict_size; // window.go:67-71
        d.bytes_left -= length;
        if (overflow) return RUN_OUT_CAP;
        if (truncated) return RUN_ERR_RESULT;
    }
    return RUN_CONTINUE;
}

#undef NORMALIZE
#undef BIT_NN_IN
#undef BIT_NN
#undef BIT_IN
#undef BIT
#undef TREE_IN
#undef TREE
#undef RTREE
#undef LEN_DECODE

// ================================================================================
//  FAST path: the whole packet loop as ONE hand-scheduled asm statement, generated by
//  tools/gen_fastpath.py into xlz_fastpath.inc (register conventions and exit codes are
//  documented there).  Same arithmetic and the same order of model updates as the checked
//  path; the end-of-input / capacity / truncation tests are hoisted into lzma_run, which
//  passes them in as two limits (arel_lim, pos_lim).
// ================================================================================
// (tools/gen_fastpath.py computes the same table bases from the same rule: model_layout / xlz_format.h: ModelLayout;
//  tests/test_fastpath_gen.py keeps both committed loops in sync with the generator)
enum : uint32_t { FX_LIMIT = 0, FX_ERR = 1, FX_MARKER = 2, FX_COPY = 3 };

#ifndef XLZ_FASTPATH_INC // A/B builds of generator variants (tools/gen_fastpath.py --variant ... --out ...)
#define XLZ_FASTPATH_INC "xlz_fastpath.inc"
#endif
#ifndef XLZ_FASTPATH_PB2_INC // the same loop over the compact layout (gen_fastpath.py --variant compact)
#define XLZ_FASTPATH_PB2_INC "xlz_fastpath_pb2.inc"
#endif
// ... and with BRANCHY decisions (gen_fastpath.py --variant compact,dbr,dbrs): every decision branches on its outcome and
// each outcome writes range and tree slot itself -- five scalar instructions per tree level instead of seven, one short
// forward branch per outcome.  Which of the two is faster depends on how many waves share a CU (profiles/r05/
// ab_dbr_at_24_per_cu.txt, ab_dbrs_combos.txt): with 16 per CU a wave mostly waits for its own previous instruction and
// the branches' bubbles cost 3-5 % (rounds 3 and 4 measured exactly that and left the switch off); with 24 per CU the
// CU's one scalar unit is what binds (0.86 scalar instructions per CU cycle of 0.97 possible) and the branchy loop is
// 4.5-5 % FASTER; with 20 per CU it is still 5 % slower.  So launches that get 24 workgroups per CU run this loop.
#ifndef XLZ_FASTPATH_PB2_BR_INC
#define XLZ_FASTPATH_PB2_BR_INC "xlz_fastpath_pb2_br.inc"
#endif
constexpr uint32_t kBranchyPerCu = 24; // workgroups per CU from which on a launch runs the branchy loop (compact layout only)

// Per-lane constants of the head gather: lane j fetches the j-th context-selected probability a
// packet can start with; its LDS byte address is hc + state * hms + state2 * hm2.
struct HeadVec {
    uint32_t hc, hms, hm2, lit_next;
};
template <class L> __device__ __forceinline__ HeadVec head_vectors(uint32_t lane)
{
    HeadVec h;
#ifdef XLZ_HEAD_PLAIN // A/B build (tools/gen_fastpath.py --without hdpp): head probability j at lane j
    const uint32_t hj = lane;
#else // head probability j alone in DPP cell (row j / 4, bank j % 4): lane 16 (j / 4) + 4 (j % 4); its update is
      // written with row_mask / bank_mask instead of a lane compare and a select
    const uint32_t hj = (lane % 4 == 0 && lane < 40) ? (lane / 16) * 4 + (lane % 16) / 4 : 10u;
#endif
    const uint32_t base = hj == 0 ? L::P_IS_MATCH
                          : hj == 1 ? L::P_IS_REP
                          : hj == 2 ? L::P_IS_REP_G0
                          : hj == 3 ? L::P_IS_REP_G1
                          : hj == 4 ? L::P_IS_REP_G2
                          : hj == 5 ? L::P_IS_REP0_LONG
                          : hj == 6 ? L::P_LEN + L::LEN_CHOICE
                          : hj == 7 ? L::P_LEN + L::LEN_CHOICE2
                          : hj == 8 ? L::P_REP_LEN + L::LEN_CHOICE
                          : hj == 9 ? L::P_REP_LEN + L::LEN_CHOICE2
                                    : L::P_LEN + 2; // the other lanes: an unused slot (their v40 is stored too)
    h.hc = base * 2;
    h.lit_next = upd_literal(lane < 12 ? lane : 0); // stateUpdateLiteral as a table: lane = old state
    h.hms = (hj >= 1 && hj <= 4) ? 2u : 0u; // indexed by state
    h.hm2 = (hj == 0 || hj == 5) ? 2u : 0u; // indexed by state2 = (state << kPosBits) + posState
#ifndef XLZ_NO_HISS // hiss: address = hc + state * (hms + 2^kPosBits hm2) + posState * hm2 (three instructions); A/B builds --without hiss: -DXLZ_NO_HISS (full layout only)
    h.hms += L::kPosStates * h.hm2;
#endif
    return h;
}

#define XLZ_FAST_OPERANDS                                                                                                          \
        : [range] "+s"(d.range), [code] "+s"(d.code), [cur] "+s"(d.cur), [arel] "+s"(d.arel), [state] "+s"(d.state),                \
          [rep0] "+s"(d.rep0), [rep1] "+s"(d.rep1), [rep2] "+s"(d.rep2), [rep3] "+s"(d.rep3), [pos] "+s"(d.pos),                    \
          [wpos] "+s"(d.wpos), [prev] "+s"(d.prev_byte), [mb] "+s"(d.match_byte), [exitc] "=&s"(exitc),                             \
          [lenout] "=&s"(lenout)                                                                                                    \
        : [arel_lim] "s"(arel_lim), [pos_lim] "s"(pos_lim), [dict] "s"(d.dict_size), [dictm1] "s"(d.dict_size - 1), [pos_mask] "s"(d.pos_mask), \
          [lc] "s"(d.lc), [lc8] "s"(8u - d.lc), [wbase] "s"(d.wbase), [outp] "s"(out), [mptr] "s"(mprobs),                          \
          [vin] "v"(d.vin), [vlane] "v"(lane), [vhc] "v"(hv.hc), [vhms] "v"(hv.hms), [vhm2] "v"(hv.hm2),                            \
          [vlitnext] "v"(hv.lit_next), [vlpm] "v"(vlpm), [vpm] "v"(vpm)                                                             \
        : "scc", "vcc", "memory", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "s88", "s89", "s90", "s91", "s92", \
          "s93", "s94", "s95", "s96", "s97", "s98", "s99", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", \
          "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", \
          "v63"

// COMPACT: the loop rendered over the compact layout (only table bases differ; hv comes from head_vectors<ModelLayout<COMPACT>>);
// BRANCHY (compact only): the loop with branchy decisions, for launches of 24 workgroups per CU
template <bool COMPACT, bool BRANCHY>
__device__ __forceinline__ uint32_t lzma_fast_loop(Dec &d, uint8_t *out, uint16_t *mprobs, uint32_t lane,
                                                   const HeadVec &hv, uint32_t arel_lim, uint32_t pos_lim,
                                                   uint32_t &lenout)
{
    uint32_t exitc;
    uint32_t vlpm; // lc + lp in a VGPR (width of the literal state's bit field): the literal context is computed on the VALU
    uint32_t vpm;  // pos_mask likewise (head gather addresses)
#ifndef XLZ_NO_LCTX // lctx: the literal state is ONE bit field of lc + lp bits of (window.pos << 8 | prevByte); [vlpm] = its width
    asm volatile("v_mov_b32 %0, %1" : "=v"(vlpm) : "s"(d.lc + (uint32_t)__builtin_popcount(d.lp_mask)));
#else // A/B build of tools/gen_fastpath.py --without lctx: [vlpm] = lp_mask
    asm volatile("v_mov_b32 %0, %1" : "=v"(vlpm) : "s"(d.lp_mask));
#endif
    asm volatile("v_mov_b32 %0, %1" : "=v"(vpm) : "s"(d.pos_mask));
    static_assert(COMPACT || !BRANCHY, "the branchy loop exists over the compact layout only");
    if constexpr (COMPACT && BRANCHY) {
        asm volatile(
#include XLZ_FASTPATH_PB2_BR_INC
            XLZ_FAST_OPERANDS);
    } else if constexpr (COMPACT) {
        asm volatile(
#include XLZ_FASTPATH_PB2_INC
            XLZ_FAST_OPERANDS);
    } else {
        asm volatile(
#include XLZ_FASTPATH_INC
            XLZ_FAST_OPERANDS);
    }
    return exitc;
}
#undef XLZ_FAST_OPERANDS

// The SIMD's instruction arbiter serves its OLDEST wave first: of the four waves that share a
// SIMD the first one dispatched runs ~1.45x faster than the last (measured: the unit durations of
// a 4096-stream launch fell into four classes of 1024 waves: 187 / 215 / 238 / 272 ms).  Total
// throughput does not care, but a launch of ONE wave round (4096 streams on 4096 slots) ends
// when the slowest class ends, with three quarters of the slots idle by then (slot occupancy
// 0.82).  So the waves of a SIMD schedule themselves longest-remaining-first: each time a wave
// refills its input window (every 224 compressed bytes) it publishes the compressed bytes its
// unit still has to decode in a table with one word per hardware wave slot, reads the 16 slots of
// its SIMD (one lane each), and sets its user priority (s_setprio, 0..3, considered before age)
// to 3 - (number of waves that have more left).  The waves of a SIMD then finish together: slot
// occupancy 0.99, +17 % on the 4096-stream configs.  (Rotating the priority blindly gave +14 %;
// ranking over the whole CU instead of the SIMD: no further gain.)
__device__ __forceinline__ void set_priority(uint32_t p)
{
#ifdef XLZ_NO_SETPRIO // A/B builds (tools/overlap_probe.py: do prioritised decode waves starve a copy kernel's waves?)
    return;
#endif
    if (p == 0)
        __builtin_amdgcn_s_setprio(0);
    else if (p == 1)
        __builtin_amdgcn_s_setprio(1);
    else if (p == 2)
        __builtin_amdgcn_s_setprio(2);
    else
        __builtin_amdgcn_s_setprio(3);
}

__device__ __forceinline__ void rank_priority(Dec &d, uint32_t lane)
{
    const uint32_t mine = d.work_end - (d.win0 + d.arel);
    __hip_atomic_store(d.prio_slot, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint32_t *base = reinterpret_cast<uint32_t *>(reinterpret_cast<uint64_t>(d.prio_slot) & ~(uint64_t)63);
    const uint32_t other = __hip_atomic_load(base + (lane & 15), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint64_t more = __builtin_amdgcn_ballot_w64(other > mine && lane < 16);
    const uint32_t rank = (uint32_t)__builtin_popcountll(more);
    set_priority(rank >= 3 ? 0u : 3u - rank);
}

// (*Reader1).decompress run to the end of the current LZMA chunk (decompress.go:8-1136)
template <bool COMPACT, bool BRANCHY>
__device__ __forceinline__ int lzma_run(Dec &d, uint16_t *probs, uint16_t *__restrict__ mprobs,
                                        uint8_t *__restrict__ out, uint32_t lane, const HeadVec &hv, bool allow_fast)
{
    for (;;) {
        // decompress.go:14-20
        if (d.size_defined && d.bytes_left == 0 && d.code == 0) return RUN_END;
        // pull readers: decompress(need) returns once enough bytes are pending (decompress.go:13)
        if (d.pos >= d.pause_at) return RUN_PAUSE;
        // keep kFastInput bytes of window ahead of the packet
        if (d.arel > kInWindow - kFastInput) {
            in_window(d, in_pos(d), lane);
            rank_priority(d, lane);
        }
        const uint32_t in_left = d.aend - in_pos(d);
        if (in_left < d.in_margin) { // the input window of a longer stream runs low: not an EOF
            d.need_input = 1;
            return RUN_PAUSE;
        }
        const bool fast = allow_fast && in_left >= kFastInput && (d.out_cap - d.pos) >= kFastOutput &&
                          (!d.size_defined || d.bytes_left >= kFastOutput);
        if (!fast) {
            const int r = lzma_packet_checked<ModelLayout<COMPACT>>(d, probs, mprobs, out, lane);
            if (r != RUN_CONTINUE) return r;
            continue;
        }
        // a packet may START while arel <= arel_lim and pos < pos_lim
        const uint32_t arel_lim = min(kInWindow - kFastInput, d.arel + (in_left - kFastInput));
        uint32_t room = d.out_cap - d.pos - kFastOutput;
        if (d.size_defined) room = (uint32_t)min((uint64_t)room, d.bytes_left - kFastOutput);
        const uint32_t pos0 = d.pos;
        uint32_t len = 0;
        const uint32_t ec = lzma_fast_loop<COMPACT, BRANCHY>(d, out, mprobs, lane, hv, arel_lim, min(pos0 + room + 1, d.pause_at), len);
        d.bytes_left -= d.pos - pos0; // :172,660,665 ... (wraps harmlessly when the size is undefined)
        if (ec == FX_ERR) return RUN_ERR_RESULT; // :651-653, :690-692
        if (ec == FX_MARKER) {                    // end marker :633-645 (bytesLeft > 0 here if defined)
            if (d.code == 0 && !d.size_defined) return RUN_END;
            return RUN_ERR_RESULT;
        }
        if (ec == FX_COPY) { // the packet is decoded, its window.CopyMatch is still to do
            uint32_t dist = d.rep0 + 1;
            if (dist == 0) dist = d.dict_size;
            if ((d.out_cap - d.pos) >= kFastCopyRoom && (!d.size_defined || d.bytes_left >= kFastCopyRoom)) {
                wave_copy<false>(out, d, dist, len, lane); // whole 64-byte rows: room for the longest match + a row
                d.pos += len;
                d.wpos += len;
                if (d.wpos >= d.dict_size) d.wpos -= d.dict_size; // window.go:67-71
                d.bytes_left -= len;
            } else {
                // near the end of the unit's output or of the announced size: the copy as the checked path does it
                // (decompress.go:657-668: truncate to bytesLeft, copy, then the error)
                bool truncated = false, overflow = false;
                if (d.size_defined && (uint32_t)d.bytes_left < len) {
                    len = (uint32_t)d.bytes_left;
                    truncated = true;
                }
                if (len > d.out_cap - d.pos) {
                    len = d.out_cap - d.pos;
                    overflow = true;
                }
                if (len > 0) wave_copy<true>(out, d, dist, len, lane);
                d.pos += len;
                d.wpos += len;
                if (d.wpos >= d.dict_size) d.wpos -= d.dict_size;
                d.bytes_left -= len;
                if (overflow) return RUN_OUT_CAP;
                if (truncated) return RUN_ERR_RESULT;
            }
        }
    }
}

// range_decoder.go:27-46.  0 ok, 1 io.EOF, 2 first byte != 0
__device__ __forceinline__ int rc_init(Dec &d)
{
    uint32_t b;
    d.range = 0xFFFFFFFFu;
    d.code = 0;
    if (in_pos(d) == d.aend) return 1;
    IN_BYTE(b);
    if (b != 0) return 2;
    for (int i = 0; i < 4; i++) {
        if (in_pos(d) == d.aend) return 1;
        IN_BYTE(b);
        d.code = (d.code << 8) | b;
    }
    return 0;
}

// state.Reset (state.go:79-121): every prob to 1024, all 64 lanes, 4 bytes each
__device__ __forceinline__ void probs_reset(uint16_t *probs, uint32_t nprobs, uint32_t lane)
{
    uint32_t *w = reinterpret_cast<uint32_t *>(probs);
    const uint32_t nw = nprobs / 2;
    // uniform trip count; lanes past the end re-write the last word (no divergent exit)
    for (uint32_t base = 0; base < nw; base += kWave) w[min(base + lane, nw - 1)] = kProbInitPair;
}

__device__ __forceinline__ void set_unpack_size(Dec &d, uint64_t u)
{
    // state.go:123-151: defined unless all eight bytes are 0xFF
    d.size_defined = u != ~(uint64_t)0 ? 1u : 0u;
    d.bytes_left = u;
}

// ---- LZMA2 framing (reader2.go:100-298), one wave walking the chunks of a unit ----
// stored chunk body: window.ReadFrom + ReadPending (reader2.go:252-294, window.go:142-155)
// This is the one sub-path that is a plain copy, i.e. bound by HBM and not by a dependent chain (the shape of
// randomfile.dat.lzma2, reader2_test.go:31-36): 16 bytes per lane and kStoredGroups groups per lane in flight, 4 KiB per wave
// and step.  The destination is brought to a 16-byte boundary first; the source then sits at any byte offset, so a group is
// five ALIGNED dwords funnelled together with v_alignbyte (the shift is wave-uniform) -- no unaligned access, at most 4 bytes
// read behind the chunk (the input window's own 256-byte loads read further, kArenaTailPad).  Lanes past the end repeat the
// last byte / group: same address, same value, no divergent branch.
struct __attribute__((aligned(4))) StoredQuad { uint32_t x, y, z, w; };
#ifndef XLZ_STORED_GROUPS
#define XLZ_STORED_GROUPS 4
#endif
constexpr uint32_t kStoredGroups = XLZ_STORED_GROUPS;

__device__ __forceinline__ void stored_copy(const uint8_t *__restrict__ src, uint8_t *__restrict__ out, uint32_t pos,
                                            uint32_t n, uint32_t lane)
{
    uint8_t *__restrict__ dst = out + pos;
    const uint32_t head = min((uint32_t)(0 - (uint32_t)(uintptr_t)dst) & 15u, n);
    if (head) {
        const uint32_t i = min(lane, head - 1);
        dst[i] = src[i];
    }
    const uint32_t groups = (n - head) >> 4;
    if (groups) {
        const uint8_t *s0 = src + head;
        const uint32_t sh = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(uintptr_t)s0 & 3u));
        const uint8_t *sa = s0 - sh; // dword-aligned
        uint8_t *d0 = dst + head;    // 16-byte-aligned
        for (uint32_t base = 0; base < groups; base += kWave * kStoredGroups) {
            StoredQuad q[kStoredGroups];
            uint32_t e[kStoredGroups], g[kStoredGroups];
#pragma unroll
            for (uint32_t u = 0; u < kStoredGroups; u++) {
                g[u] = min(base + u * kWave + lane, groups - 1);
                const uint8_t *p = sa + (size_t)g[u] * 16;
                q[u] = *(const StoredQuad *)p;
                e[u] = *(const uint32_t *)(p + 16);
            }
#pragma unroll
            for (uint32_t u = 0; u < kStoredGroups; u++) {
                uint4 o;
                o.x = __builtin_amdgcn_alignbyte(q[u].y, q[u].x, sh);
                o.y = __builtin_amdgcn_alignbyte(q[u].z, q[u].y, sh);
                o.z = __builtin_amdgcn_alignbyte(q[u].w, q[u].z, sh);
                o.w = __builtin_amdgcn_alignbyte(e[u], q[u].w, sh);
#ifdef XLZ_STORED_NT
                __builtin_nontemporal_store(o.x, (uint32_t *)(d0 + (size_t)g[u] * 16));
                __builtin_nontemporal_store(o.y, (uint32_t *)(d0 + (size_t)g[u] * 16) + 1);
                __builtin_nontemporal_store(o.z, (uint32_t *)(d0 + (size_t)g[u] * 16) + 2);
                __builtin_nontemporal_store(o.w, (uint32_t *)(d0 + (size_t)g[u] * 16) + 3);
#else
                *(uint4 *)(d0 + (size_t)g[u] * 16) = o;
#endif
            }
        }
    }
    const uint32_t done = head + (groups << 4);
    if (n > done) {
        const uint32_t i = done + min(lane, n - done - 1);
        dst[i] = src[i];
    }
}

// prevByte / matchByte straight from the window (after a stored chunk they are not in registers)
__device__ __forceinline__ void reload_context(Dec &d, const uint8_t *__restrict__ out, uint32_t lane)
{
    // lane 0: byte at distance 1, lane 1: byte at distance rep0 + 1 (window.GetByte, window.go:44-53)
    uint32_t dist = (lane & 1) ? d.rep0 + 1 : 1u;
    if (dist == 0) dist = d.dict_size;
    const uint64_t lo = (uint64_t)dist + d.wbase;
    const bool ok = (uint64_t)d.pos >= lo;
    uint32_t b = out[ok ? d.pos - dist : 0];
    b = ok ? b : 0u;
    if (!(d.wbase == 0 && d.epoch0_clean)) { // bytes in front of this epoch may be an earlier epoch's
        const uint32_t fill = d.pos - d.wbase;
        const bool full = fill >= d.dict_size;
        const bool valid = dist <= d.dict_size; // GetByte with a larger distance indexes out of the reference's buffer
        if (d.epochs || d.shadow) {
            const uint32_t sb = stale_byte(out, d, fill + d.dict_size - dist, !ok && !full && valid);
            b = (!ok && !full && valid) ? sb : b;
        } else if (d.state >= 7 && __builtin_amdgcn_readlane((int)(uint32_t)(!ok && !full), 1)) {
            // matchByte would come from an earlier epoch (prevByte is 0 on an empty window).  Only a literal in a
            // match state reads it (decompress.go:59), and every copy on the way to such a state loads its own.
            d.stale = 1;
        }
    }
    d.prev_byte = (uint32_t)__builtin_amdgcn_readlane((int)b, 0);
    d.match_byte = (uint32_t)__builtin_amdgcn_readlane((int)b, 1);
    if (d.pos == d.wbase) d.prev_byte = 0; // window.IsEmpty (decompress.go:50-53)
}

// the matched-literal part of the model (HBM): 16 bytes per lane and step
__device__ __forceinline__ void mprobs_reset(uint16_t *__restrict__ mprobs, uint32_t n, uint32_t lane)
{
    uint4 *w = reinterpret_cast<uint4 *>(mprobs);
    const uint32_t nw = n / 8; // 8 probs per 16 bytes; n is a multiple of 256
    const uint4 v = make_uint4(kProbInitPair, kProbInitPair, kProbInitPair, kProbInitPair);
    for (uint32_t base = 0; base < nw; base += kWave) w[min(base + lane, nw - 1)] = v;
}

__device__ __forceinline__ void state_reset(Dec &d, uint16_t *probs, uint16_t *__restrict__ mprobs, uint32_t lc_lp,
                                            uint32_t lane, bool compact)
{
    probs_reset(probs, num_probs(lc_lp, compact), lane); // state.Reset, state.go:79-121
    mprobs_reset(mprobs, num_matched_probs(lc_lp), lane);
    d.state = 0;
    d.rep0 = d.rep1 = d.rep2 = d.rep3 = 0;
}

// LZMA2 walker state that outlives a chunk
struct Walk {
    uint32_t unit_end; // first position after the unit's input
    uint32_t lc_lp;
    uint32_t h5;       // header[5] persists across chunks (reader2.go:37,147)
    bool last_unit, have_reader, first_chunk;
    bool more_input;   // UNIT_F_MORE_INPUT: the unit's input is a window of a longer stream
    bool resumable;    // the unit has a state block (pull reader): it can pause for a larger model
    uint32_t model_lc_lp; // largest lc+lp this unit's model storage was sized for
    bool compact;         // the launch's model layout has room for 2^kCompactPosBits posStates only (xlz_format.h: ModelLayout)
};

enum : int32_t { WALK_RUN_CHUNK = 1000 }; // lzma2_next: a compressed chunk is set up, run it

// Reader2.startChunk (reader2.go:100-173) plus the stored-chunk body.  Returns
// WALK_RUN_CHUNK when an LZMA chunk is ready for lzma_run, otherwise the unit's final status.
__device__ __forceinline__ int32_t lzma2_next(Dec &d, Walk &w, uint16_t *probs, uint16_t *__restrict__ mprobs,
                                              const uint8_t *__restrict__ in_bytes,
                                              uint8_t *__restrict__ out, uint32_t max_lc_lp, uint32_t lane,
                                              uint32_t &aux)
{
    for (;;) {
        d.aend = w.unit_end;
        if (d.pos >= d.pause_at) return ST_PAUSED; // pull readers: enough bytes are pending
        if (w.more_input && w.unit_end - in_pos(d) < kLzma2InputMargin) { // a whole chunk must be in the window
            d.need_input = 1;
            return ST_PAUSED;
        }
        if (in_pos(d) == w.unit_end) // ReadByte fails (reader2.go:103-110)
            return w.last_unit ? ST_ERR_UNEXPECTED_EOF : ST_OK;
        if (d.arel > kInWindow - kFastInput) in_window(d, in_pos(d), lane);
        const uint32_t chunk_start = in_pos(d);
        uint32_t c, h1 = 0, h2 = 0, h3 = 0, h4 = 0;
        IN_BYTE(c);
        // decodeChunkType (reader2.go:175-199): 0x03..0x7F fall through to end-of-stream
        if (c == 0 || (c >= 3 && c < 0x80)) {
            aux |= AUX_END_MARK;
            return ST_OK;
        }
        const bool stored = c < 3;
        const uint32_t sub = c >> 5;                            // 4 no reset, 5 state, 6 +props, 7 +dict
        const uint32_t hl = stored ? 3u : (sub >= 6 ? 6u : 5u); // chunkLength (reader2.go:201-214)
        if (w.unit_end - in_pos(d) < hl - 1) {                  // io.ReadFull comes up short (:121-128)
            in_window(d, w.unit_end, lane);
            return ST_ERR_UNEXPECTED_EOF;
        }
        IN_BYTE(h1);
        IN_BYTE(h2);
        if (!stored) {
            IN_BYTE(h3);
            IN_BYTE(h4);
            if (hl == 6) IN_BYTE(w.h5);
        }
        uint32_t unc = (h1 << 8) | h2; // :130
        if (!stored && w.resumable && (!w.have_reader || sub >= 6) && w.h5 < 225) {
            // pull readers: the chunk brings properties whose model is larger than the unit's state block (the host
            // sized it by the headers it had seen when the session was opened; reader2.go:159-165 accepts any lc <= 8,
            // lp <= 4 at any chunk).  Nothing of this chunk has had an effect yet: step back in front of it and pause --
            // the host gives the unit a larger block (and the HBM-model launch beyond lc+lp = 8) and resumes.
            const uint32_t need = w.h5 % 9 + (w.h5 / 9) % 5;
            if (need > w.model_lc_lp || need > max_lc_lp) {
                in_window(d, chunk_start, lane);
                aux = (aux & ~AUX_GROW_MASK) | AUX_GROW | (need << AUX_GROW_SHIFT);
                return ST_PAUSED;
            }
        }
        if (c == 1 || sub == 7) {      // dictionary reset: window.Reset (:132-134, window.go:135-140)
            if (w.resumable && !d.shadow && d.pos > d.wbase) {
                // pull readers: the first dictionary reset behind a non-empty epoch.  The session has no window image yet
                // (it is allocated when a stream first needs one, not at NewReader2): step back in front of
                // the chunk and pause; the host allocates the image and the next refill copies the epoch into it.
                in_window(d, chunk_start, lane);
                aux |= AUX_SHADOW;
                return ST_PAUSED;
            }
            if (d.shadow && d.pos > d.wbase) shadow_update(d, out, lane);
            if (!epoch_push(d)) return ST_ERR_UNSUPPORTED; // more visible epochs than the table holds
            d.wbase = d.pos;
            d.wpos = 0;
            d.prev_byte = 0; // window.IsEmpty again (decompress.go:50-53)
        }
        if (stored) {
            unc++; // :137
            uint32_t n = min(unc, w.unit_end - in_pos(d)); // a short source delivers what is there
            bool overflow = false;
            if (n > d.out_cap - d.pos) {
                n = d.out_cap - d.pos;
                overflow = true;
            }
            if (n) stored_copy(in_bytes + (in_pos(d) - d.abase), out, d.pos, n, lane);
            d.pos += n;
            d.wpos += n; // window.ReadFrom (window.go:146-153); the dictionary may be smaller than n
            while (d.wpos >= d.dict_size) d.wpos -= d.dict_size;
            // out of room: window.ReadFrom still drains the chunk from the source (window.go:142-155;
            // the oracle's byte loop does the same), only then does the missing room surface
            if (overflow) n = min(unc, w.unit_end - in_pos(d));
            in_window(d, in_pos(d) + n, lane);
            if (overflow) return ST_ERR_OUT_CAP;
            reload_context(d, out, lane);
            continue;
        }
        unc |= (c & 0x1Fu) << 16; // :141-142
        unc++;
        const uint32_t comp = ((h3 << 8) | h4) + 1; // :143-144 (32-bit: SURVEY parity note 7)
        if (!w.have_reader || sub >= 6) {
            // NewReader1ForReader2 / Renew: props from header[5] (reader2.go:146-165)
            if (w.h5 >= 225) return ST_ERR_PROPS; // DecodeProp, reader1.go:211-213
            const uint32_t lc = w.h5 % 9, r = w.h5 / 9, lp = r % 5, pb = r / 5;
            if (lc + lp > max_lc_lp || lc + lp > w.model_lc_lp || (w.compact && pb > kCompactPosBits)) {
                // LDS / the saved state are sized (and laid out: pb) by the host's header scan; AUX_GROW tells the host that
                // a launch with room for this model would go on (and not, say, a full epoch table: ADVICE r4)
                aux = (aux & ~AUX_GROW_MASK) | AUX_GROW | (min(lc + lp, 15u) << AUX_GROW_SHIFT);
                return ST_ERR_UNSUPPORTED;
            }
            d.lc = lc;
            d.lp_mask = (1u << lp) - 1;
            d.pos_mask = (1u << pb) - 1;
            w.lc_lp = lc + lp;
            state_reset(d, probs, mprobs, w.lc_lp, lane, w.compact);
        } else if (sub == 5) {
            state_reset(d, probs, mprobs, w.lc_lp, lane, w.compact); // :156-157
        }
        w.first_chunk = !w.have_reader;
        w.have_reader = true;
        // Reopen: SetUnpackSize + limitByteReader + rangeDec.Init (reader1.go:166-176)
        d.size_defined = 1;
        d.bytes_left = unc;
        d.aend = min(in_pos(d) + comp, w.unit_end);
        return WALK_RUN_CHUNK;
    }
}

// ---- pull readers: decoder state saved to / restored from HBM between launches ----------
// Word indices inside a UnitState (xlz_format.h): everything a unit carries across a pause.
enum : uint32_t {
    SV_RANGE, SV_CODE, SV_STATE, SV_REP0, SV_REP1, SV_REP2, SV_REP3, SV_LC, SV_LP_MASK, SV_POS_MASK, SV_SIZE_DEFINED,
    SV_BYTES_LEFT_LO, SV_BYTES_LEFT_HI, SV_POS, SV_WBASE, SV_WPOS, SV_PREV, SV_MATCH, SV_STALE, SV_CONSUMED, SV_CHUNK_END,
    SV_LC_LP, SV_H5, SV_HAVE_READER, SV_FIRST_CHUNK, SV_PHASE, SV_AUX,
    SV_SHADOW_LO = 32, SV_SHADOW_HI = 33, // written by the HOST when the session is opened (xlz_format.h: kStateShadowWord)
    SV_COUNT
};
static_assert(SV_AUX < SV_SHADOW_LO && SV_SHADOW_LO == kStateShadowWord, "state header layout");
static_assert(SV_COUNT <= kStateWords, "UnitState header too small");

enum : uint32_t { PH_NEXT = 0, PH_CHUNK = 1 }; // between chunks (or not started) / inside lzma_run of a chunk

// One word of saved state.  The wave-uniform value takes a scalar add and an explicit v_mov on its way
// to the store: with a plain store LLVM's SGPR-copy fixing turns loop-carried values that are only
// ever moved (the four reps) into VGPR phis, which the fast loop's "s" asm constraints cannot take
// ("illegal VGPR to SGPR copy").
__device__ __forceinline__ void sv_store(uint32_t *st, uint32_t idx, uint32_t value)
{
    uint32_t v, z;
    asm volatile("s_mov_b32 %0, 0" : "=s"(z));
    value += z; // a scalar ALU op on the value (z is opaque to the compiler)
    asm volatile("v_mov_b32 %0, %1" : "=v"(v) : "s"(value));
    st[idx] = v;
}

__device__ __forceinline__ void model_copy(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, uint32_t nprobs,
                                           uint32_t lane)
{
    const uint32_t nw = nprobs / 2;
    for (uint32_t base = 0; base < nw; base += kWave) {
        const uint32_t i = min(base + lane, nw - 1);
        dst[i] = src[i];
    }
}

// BIG = false: the model is this workgroup's LDS (the asm fast loop addresses it from LDS offset
// 0).  BIG = true (lc+lp > 6): the model is a slot of HBM scratch and only the checked C++
// packet decoder runs -- slow, but the reference's whole parameter range decodes.
template <bool BIG, bool COMPACT, bool BRANCHY = false>
__device__ __forceinline__ void decode_units(const LaunchParams &p, uint16_t *probs, uint16_t *__restrict__ wg_mprobs)
{
    static_assert(!(BIG && COMPACT), "the HBM-model launch uses the full layout");
    const uint32_t lane = threadIdx.x;
    const HeadVec hv = head_vectors<ModelLayout<COMPACT>>(lane);
    constexpr bool big = BIG;

    for (;;) {
        // dequeue: lane 0 bumps the head; lanes 1..63 add 0 to pad words of the same
        // 256-byte block (a branch here would make the whole loop divergent, see wave_copy)
        const uint32_t q = RFL(atomicAdd(p.queue + lane, lane == 0 ? 1u : 0u));
        if (q >= p.n_units) break; // every wave reaches this once the queue is drained
        const uint32_t t_start = (uint32_t)wall_clock64();

        const uint32_t ui = RFL(p.order[p.order_base + q]);
        const Unit *up = p.units + ui;
        // a later launch of a sliced sequence: only the units the launch before left paused go on (the others are settled:
        // their results stay as they are)
        const bool slice_resume = p.slice_frac != 0 && p.slice_k != 0;
        if (slice_resume && RFL((uint32_t)p.results[ui].status) != (uint32_t)ST_PAUSED) continue;
        Dec d;
        Walk w;
        const uint64_t in_off = rfl64(up->in_off);
        const uint64_t out_off = rfl64(up->out_off);
        const uint64_t unpack = rfl64(up->unpack_size);
        const uint64_t state_addr = rfl64(up->state);
        const uint32_t in_all = RFL(up->in_len);
        const bool lzma2 = RFL(up->kind) == UNIT_LZMA2;
        const bool sliced = p.slice_frac != 0; // (the host gives every unit of such a launch a state block)
        // the first launch of a sequence may start while the tails of the inputs are still on their way (head_frac)
        const uint32_t in_len = (sliced && p.slice_k == 0) ? slice_head(in_all, p.head_frac, lzma2) : in_all;
        const uint32_t flags = RFL(up->flags) | (slice_resume ? (uint32_t)UNIT_F_RESUME : 0u) | (in_len < in_all ? (uint32_t)UNIT_F_MORE_INPUT : 0u);
        d.out_cap = RFL(up->out_cap);
        d.dict_size = RFL(up->dict_size);
        const uint32_t lc = RFL(up->lc), lp = RFL(up->lp), pb = RFL(up->pb);
        uint8_t *__restrict__ out = p.out_arena + out_off;
        // a resumable unit keeps the matched-literal half of its model in its own state block
        uint32_t *const st = reinterpret_cast<uint32_t *>(state_addr);
        // (the state blocks of a sliced launch are all sized for the launch's largest model, like the LDS: a chunk of a
        //  damaged LZMA2 stream may bring larger properties than the unit's own headers announced)
        uint16_t *__restrict__ mprobs =
            st ? reinterpret_cast<uint16_t *>(state_addr + state_mprobs_off(sliced ? p.max_lc_lp : lc + lp)) : wg_mprobs;
        const bool resume = (flags & UNIT_F_RESUME) != 0;

        d.dump = reinterpret_cast<uint8_t *>(p.queue + 64);
        {   // this wave's word of the priority table: XCC_ID[3:0] : HW_ID[15:0] (wave slot, SIMD, pipe, CU, SH, SE)
            const uint32_t hw = __builtin_amdgcn_s_getreg((16 - 1) << 11 | 0 << 6 | 4);
            const uint32_t xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);
            d.prio_slot = p.prio_tab + ((xcc << 16) | hw);
        }
        d.epochs = p.epochs ? p.epochs + (size_t)blockIdx.x * kMaxEpochs : nullptr;
        d.n_epochs = 0;
        d.epoch0_clean = !lzma2 || !(flags & UNIT_F_NOT_FIRST);
        d.pause_at = sliced ? slice_bound(d.out_cap, p.slice_frac) : st ? RFL(up->pause_at) : 0xFFFFFFFFu;
        d.in_margin = (!lzma2 && (flags & UNIT_F_MORE_INPUT)) ? kLzma1InputMargin : 0u;
        d.need_input = 0;
        d.in_base = resume ? RFL(up->in_skip) : 0u;
        in_open(d, p.in_arena, in_off, in_len, lane);
        w.unit_end = d.abase + in_len;
        d.work_end = w.unit_end;
        w.last_unit = (flags & UNIT_F_LAST) != 0;
        w.more_input = lzma2 && (flags & UNIT_F_MORE_INPUT);
        // only a pull reader's state block is sized per unit (LDS, and the blocks of a sliced launch: per launch)
        w.compact = COMPACT;
        w.model_lc_lp = (st && !sliced) ? lc + lp : 0xFFu;
        w.resumable = st != nullptr && !sliced; // (a unit of a sliced launch asks the host for nothing)
        d.shadow = (st && lzma2 && !sliced)
                       ? reinterpret_cast<uint8_t *>(rfl64(*reinterpret_cast<const uint64_t *>(st + SV_SHADOW_LO)))
                       : nullptr;

        uint32_t phase = PH_NEXT;
        uint32_t aux = 0;
        if (resume) {
            d.range = RFL(st[SV_RANGE]);
            d.code = RFL(st[SV_CODE]);
            d.state = RFL(st[SV_STATE]);
            d.rep0 = RFL(st[SV_REP0]);
            d.rep1 = RFL(st[SV_REP1]);
            d.rep2 = RFL(st[SV_REP2]);
            d.rep3 = RFL(st[SV_REP3]);
            d.lc = RFL(st[SV_LC]);
            d.lp_mask = RFL(st[SV_LP_MASK]);
            d.pos_mask = RFL(st[SV_POS_MASK]);
            d.size_defined = RFL(st[SV_SIZE_DEFINED]);
            d.bytes_left = ((uint64_t)RFL(st[SV_BYTES_LEFT_HI]) << 32) | RFL(st[SV_BYTES_LEFT_LO]);
            // the host may have moved the window down (the last dict_size bytes stay reachable)
            const uint32_t rebase = RFL(up->rebase);
            const uint32_t wb = RFL(st[SV_WBASE]);
            d.pos = RFL(st[SV_POS]) - rebase;
            d.wbase = wb > rebase ? wb - rebase : 0u;
            if (wb < rebase) d.epoch0_clean = false; // the epoch's start left the buffer (only if it is full anyway)
            d.wpos = RFL(st[SV_WPOS]);
            d.prev_byte = RFL(st[SV_PREV]);
            d.match_byte = RFL(st[SV_MATCH]);
            d.stale = RFL(st[SV_STALE]);
            w.lc_lp = RFL(st[SV_LC_LP]);
            w.h5 = RFL(st[SV_H5]);
            w.have_reader = RFL(st[SV_HAVE_READER]) != 0;
            w.first_chunk = RFL(st[SV_FIRST_CHUNK]) != 0;
            phase = RFL(st[SV_PHASE]);
            aux = RFL(st[SV_AUX]) & ~(AUX_GROW | AUX_GROW_MASK | AUX_SHADOW); // (requests to the host are answered by now)
            // input: continue at the saved position inside the (possibly moved) input window
            // ((*Reader1).Reopen: the input is a new stream, read from its first byte)
            const uint32_t consumed = (flags & UNIT_F_REOPEN) ? d.in_base : RFL(st[SV_CONSUMED]);
            in_window(d, d.abase + (consumed - d.in_base), lane);
            // inside an LZMA2 chunk the limitedByteReader's end stays where the chunk header put it
            d.aend = (lzma2 && phase == PH_CHUNK) ? min(d.abase + (RFL(st[SV_CHUNK_END]) - d.in_base), w.unit_end) : w.unit_end;
            model_copy(reinterpret_cast<uint32_t *>(probs), st + kStateWords, num_probs(w.lc_lp, COMPACT), lane);
            if (flags & UNIT_F_RESET_MODEL) state_reset(d, probs, mprobs, w.lc_lp, lane, COMPACT); // (*Reader1).Reset
            if (flags & UNIT_F_REOPEN) { // (*Reader1).Reopen: SetUnpackSize, then rangeDec.Reopen on the new input
                set_unpack_size(d, unpack);
                d.aend = w.unit_end;
                phase = PH_NEXT;
            }
        } else {
            d.lc = lc;
            d.lp_mask = (1u << lp) - 1;
            d.pos_mask = (1u << pb) - 1;
            d.state = 0;
            d.rep0 = d.rep1 = d.rep2 = d.rep3 = 0;
            d.pos = 0;
            d.wbase = 0;
            d.wpos = 0;
            d.prev_byte = 0;
            d.match_byte = 0;
            d.stale = 0;
            d.range = 0;
            d.code = 0;
            set_unpack_size(d, unpack);
            w.lc_lp = lc + lp;
            w.h5 = 0;
            w.have_reader = (flags & UNIT_F_HAVE_READER) != 0;
            w.first_chunk = true;
            if (!lzma2) { // newState -> Reset (state.go:47-61)
                probs_reset(probs, num_probs(lc + lp, COMPACT), lane);
                mprobs_reset(mprobs, num_matched_probs(lc + lp), lane);
            }
        }

        int32_t status;
        // LZMA1: exactly one "chunk" (the whole stream).  LZMA2: one per compressed chunk.
        for (;;) {
            if (phase != PH_CHUNK) {
                if (lzma2) {
                    status = lzma2_next(d, w, probs, mprobs, p.in_arena + in_off, out, p.max_lc_lp, lane, aux);
                    if (status != WALK_RUN_CHUNK) break;
                }
                const int ir = rc_init(d); // Reader1.initialize / Reopen (reader1.go:149-176)
                if (ir == 1) { // io.EOF: a constructor error, or -- raw, from a later LZMA2 chunk -- a clean EOF
                    status = (!lzma2 || w.first_chunk) ? ST_ERR_HEADER_EOF : ST_OK_INPUT_EOF;
                    break;
                }
                if (ir == 2) {
                    status = ST_ERR_RC_INIT;
                    break;
                }
                phase = PH_CHUNK;
                // The previous chunk (or the stream before a Reopen) may have ended inside a packet, after
                // state / rep0 moved on (decompress.go:216,431,785-798 mutate before the next ReadByte can
                // fail): the first literal then takes its matchByte at the rep0 of NOW, not the one loaded
                // after the last copy.
                if (d.state >= 7) reload_context(d, out, lane);
            }
            const int r = lzma_run<COMPACT, BRANCHY>(d, probs, mprobs, out, lane, hv, !big);
            if (r == RUN_PAUSE) {
                status = ST_PAUSED;
                break;
            }
            status = r == RUN_END ? ST_OK
                                  : r == RUN_INPUT_EOF ? ST_OK_INPUT_EOF
                                                       : r == RUN_OUT_CAP ? ST_ERR_OUT_CAP : ST_ERR_RESULT;
            // LZMA2: io.EOF from the chunk -> next startChunk (reader2.go:234-241); the unread
            // rest of a chunk is NOT skipped by the reference
            if (status < 0 || !lzma2) break;
            phase = PH_NEXT;
        }
        const uint32_t consumed = d.in_base + (in_pos(d) - d.abase);
        if (st) { // a resumable unit keeps its state whatever ended the launch (Reopen continues after an end too)
            sv_store(st, SV_RANGE, d.range);
            sv_store(st, SV_CODE, d.code);
            sv_store(st, SV_STATE, d.state);
            sv_store(st, SV_REP0, d.rep0);
            sv_store(st, SV_REP1, d.rep1);
            sv_store(st, SV_REP2, d.rep2);
            sv_store(st, SV_REP3, d.rep3);
            sv_store(st, SV_LC, d.lc);
            sv_store(st, SV_LP_MASK, d.lp_mask);
            sv_store(st, SV_POS_MASK, d.pos_mask);
            sv_store(st, SV_SIZE_DEFINED, d.size_defined);
            sv_store(st, SV_BYTES_LEFT_LO, (uint32_t)d.bytes_left);
            sv_store(st, SV_BYTES_LEFT_HI, (uint32_t)(d.bytes_left >> 32));
            sv_store(st, SV_POS, d.pos);
            sv_store(st, SV_WBASE, d.wbase);
            sv_store(st, SV_WPOS, d.wpos);
            sv_store(st, SV_PREV, d.prev_byte);
            sv_store(st, SV_MATCH, d.match_byte);
            sv_store(st, SV_STALE, d.stale);
            sv_store(st, SV_CONSUMED, consumed);
            sv_store(st, SV_CHUNK_END, d.in_base + (d.aend - d.abase));
            sv_store(st, SV_LC_LP, w.lc_lp);
            sv_store(st, SV_H5, w.h5);
            sv_store(st, SV_HAVE_READER, w.have_reader ? 1u : 0u);
            sv_store(st, SV_FIRST_CHUNK, w.first_chunk ? 1u : 0u);
            sv_store(st, SV_PHASE, phase);
            sv_store(st, SV_AUX, aux);
            model_copy(st + kStateWords, reinterpret_cast<const uint32_t *>(probs), num_probs(w.lc_lp, COMPACT), lane);
        }
        {
            UnitResult res; // every lane stores the same 32 bytes
            res.out_len = d.pos;
            res.in_consumed = consumed;
            res.status = status;
            res.aux = aux | (d.stale ? AUX_STALE : 0u) | (d.need_input ? AUX_NEED_INPUT : 0u);
            res.t_start = t_start;
            res.t_end = (uint32_t)wall_clock64();
            p.results[ui] = res;
        }
        __hip_atomic_store(d.prio_slot, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // idle: no claim on priority
    }
}

// Resident waves per SIMD.  The compiler's default register allocation took 108 VGPRs -- FOUR waves per SIMD (512 / 112),
// so a grid of 20 single-wave workgroups per CU was never resident (round 2's "20 waves per CU: +-0.5 %" measured nothing).
// Round 3 capped the allocation at 96 VGPRs (five waves per SIMD).  Round 5: SIX -- 80 VGPRs (the fast loop's fixed v13..v63
// stay; the C++ around it spills in cold code) cost nothing measurable (profiles/r05/ab_waves6.txt: the same 20 workgroups
// per CU +0.1 % / 0.0 %), and every workgroup beyond 20 that LDS admits is worth about 1.7 %: 21 per CU (the 7416-byte model
// of lc+lp = 3: 163840 / 7680) +3.1 % on the 65 536 x 64 KiB batch, 24 per CU (a 5240-byte model, lc+lp = 2) +7.0 %.
#ifndef XLZ_WAVES_PER_EU
#define XLZ_WAVES_PER_EU 6
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(XLZ_WAVES_PER_EU, XLZ_WAVES_PER_EU)))
void xlz_decode_kernel(LaunchParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint16_t lds_probs[];
    decode_units<false, false>(p, lds_probs, p.mlit + (size_t)blockIdx.x * p.mlit_stride);
}

// the same over the COMPACT model layout (every unit's pb <= 2: xlz_format.h: ModelLayout): five LDS granules instead of six
// for lc+lp = 3, 24 workgroups per CU
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(XLZ_WAVES_PER_EU, XLZ_WAVES_PER_EU)))
void xlz_decode_kernel_pb2(LaunchParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint16_t lds_probs[];
    decode_units<false, true>(p, lds_probs, p.mlit + (size_t)blockIdx.x * p.mlit_stride);
}

// ... with the branchy loop: launches of kBranchyPerCu workgroups per CU
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(XLZ_WAVES_PER_EU, XLZ_WAVES_PER_EU)))
void xlz_decode_kernel_pb2_br(LaunchParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint16_t lds_probs[];
    decode_units<false, true, true>(p, lds_probs, p.mlit + (size_t)blockIdx.x * p.mlit_stride);
}

__global__ __launch_bounds__(64) void xlz_decode_kernel_hbm_model(LaunchParams p)
{
    uint16_t *slot = p.scratch + (size_t)blockIdx.x * p.scratch_stride; // model, then its matched part
    decode_units<true, false>(p, slot, slot + num_probs(p.max_lc_lp));
}

// Pieces of an arena <-> one packed image (xlz_format.h: SlicePiece; SCATTER = false: arena -> image, the download of a
// sliced batch; true: image -> arena, the heads / tails of its upload).  A workgroup takes 16 KiB tiles of the packed
// image: it finds the piece its tile starts in (the table is sorted by pack_off) and copies what the tile holds of that
// piece and the following ones -- 16 bytes per lane once both sides are on a 16-byte boundary (a piece sits in the image at
// its arena offset modulo 256, so they get there together), the bytes in front and behind one by one.  Runs on a copy
// stream NEXT TO the persistent decode grid (which leaves wave slots, registers and LDS free: 16 of 32 waves per CU).
constexpr uint64_t kGatherTile = 16384;
template <bool SCATTER>
__global__ __launch_bounds__(256) void xlz_gather_kernel(const SlicePiece *__restrict__ pc, uint32_t n, uint8_t *__restrict__ arena,
                                                          uint8_t *__restrict__ pack, uint64_t pack_bytes)
{
    const uint64_t tiles = (pack_bytes + kGatherTile - 1) / kGatherTile;
    for (uint64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const uint64_t t0 = t * kGatherTile, t1 = min(t0 + kGatherTile, pack_bytes);
        uint32_t lo = 0, hi = n; // the last piece that starts at or in front of t0 (piece 0 if none does)
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (pc[mid].pack_off <= t0)
                lo = mid;
            else
                hi = mid;
        }
        for (uint32_t i = lo; i < n; i++) {
            const SlicePiece q = pc[i];
            if (q.pack_off >= t1) break;
            const uint64_t a = max(q.pack_off, t0), e = min(q.pack_off + q.len, t1);
            if (e <= a) continue;
            uint8_t *__restrict__ in_arena = arena + q.src_off + (a - q.pack_off);
            uint8_t *__restrict__ in_pack = pack + a;
            const uint8_t *__restrict__ src = SCATTER ? in_pack : in_arena;
            uint8_t *__restrict__ dst = SCATTER ? in_arena : in_pack;
            const uint32_t nb = (uint32_t)(e - a);
            const uint32_t peel = min((16u - (uint32_t)(a & 15u)) & 15u, nb); // a is the image offset: the arena's is congruent
            const uint32_t nv = (nb - peel) >> 4;
            if (threadIdx.x < peel) dst[threadIdx.x] = src[threadIdx.x];
            for (uint32_t j = threadIdx.x; j < nv; j += 256)
                reinterpret_cast<uint4 *>(dst + peel)[j] = reinterpret_cast<const uint4 *>(src + peel)[j];
            for (uint32_t j = peel + (nv << 4) + threadIdx.x; j < nb; j += 256) dst[j] = src[j];
        }
    }
}

int launch_gather(const SlicePiece *pieces, uint32_t n_pieces, uint8_t *arena, uint8_t *pack, uint64_t pack_bytes, int num_cus,
                  void *stream, bool scatter)
{
    if (!n_pieces || !pack_bytes) return 0;
    const uint64_t tiles = (pack_bytes + kGatherTile - 1) / kGatherTile;
    const uint32_t grid = (uint32_t)std::min<uint64_t>(tiles, (uint64_t)num_cus * 4);
    if (scatter)
        hipLaunchKernelGGL(xlz_gather_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, pieces, n_pieces, arena, pack, pack_bytes);
    else
        hipLaunchKernelGGL(xlz_gather_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, pieces, n_pieces, arena, pack, pack_bytes);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

uint32_t decode_lds_bytes(uint32_t max_lc_lp, bool compact) { return num_probs(max_lc_lp, compact) * 2u + 128u; }

uint32_t big_model_grid(int num_cus) { return (uint32_t)num_cus; } // one workgroup per CU
// Resident single-wave workgroups per CU of the LDS-model launch.  gfx950 hands out LDS in granules of 1280 bytes (160 KiB /
// 128): the 7416-byte model of lc+lp = 3 (+ 128 bytes of dump rows) takes six of them and TWENTY-ONE fit a CU (round 2's
// 7928-byte model took seven: 18; profiles/r03/ab_occupancy.txt); the register allocation (80 VGPRs) allows six waves per
// SIMD.  Per-wave speed at 16 waves per CU is already 78 % of a lone wave's (bench.py roofline.issue.latency_bound): every
// further wave buys less than its share, and nothing at all on launches of few rounds, where it only stretches every round.
constexpr uint32_t kLdsGranule = 1280;
static uint32_t decode_per_cu(uint32_t max_lc_lp, uint32_t n_units, int num_cus, bool compact)
{
    const uint32_t alloc = (decode_lds_bytes(max_lc_lp, compact) + kLdsGranule - 1) / kLdsGranule * kLdsGranule;
    uint32_t per_cu = kMaxLdsBytes / alloc;
#ifdef XLZ_PER_CU_MAX // A/B builds
    if (per_cu > XLZ_PER_CU_MAX) per_cu = XLZ_PER_CU_MAX;
#else
    if (per_cu > 4 * XLZ_WAVES_PER_EU) per_cu = 4 * XLZ_WAVES_PER_EU; // what the register allocation lets be resident
    // More waves per CU raise the CU's rate but stretch every round, so they only pay when the launch has rounds to spare:
    // the most a CU holds from 3.2 rounds of that many on, else 20, else 16 (four per SIMD).  Measured on the headline's
    // data (profiles/r05/ab_shard_sizes.txt): 16 384 units -- 3.2 rounds of 5120 -- run 6.7 % faster with 20 per CU than as
    // four rounds of 16 per CU (round 3's rule asked for four rounds of 5120 and chose 16); 8192 units 4.1 % slower.
    // ... and a launch that fits ONE round at a higher occupancy takes it: 5000 units are one round of 20 per CU (every
    // wave a little slower) instead of a full round of 16 per CU and a second one a fifth full -- and a call of one round
    // can overlap its copies with its decode (xlz_host.hip: slices)
    const uint32_t one_round[3] = {16u, 20u, per_cu};
    for (uint32_t c : one_round)
        if (c <= per_cu && n_units <= c * (uint32_t)num_cus) return c > 4 && c < 16 ? (c & ~3u) : c;
    const uint32_t steps[3] = {per_cu, 20u, 16u};
    for (uint32_t c : steps)
        if (c <= per_cu && (c <= 16 || (uint64_t)n_units * 5u >= 16ull * c * (uint32_t)num_cus)) { // (3.2 rounds: not "just over three")
            per_cu = c;
            break;
        }
    if (per_cu > 4 && per_cu < 16) per_cu &= ~3u; // equal load on the four SIMDs
#endif
    return per_cu;
}

// resident workgroups of the LDS-model launch over n_units units (n_units = ~0u: the most any launch uses)
uint32_t decode_grid(uint32_t max_lc_lp, int num_cus, uint32_t n_units, bool compact)
{
    return decode_per_cu(max_lc_lp, n_units, num_cus, compact) * (uint32_t)num_cus;
}

// does a launch over n_units units (~0u: of many rounds) run the branchy loop?  (compact layout at kBranchyPerCu per CU)
bool decode_branchy(uint32_t max_lc_lp, int num_cus, uint32_t n_units, bool compact)
{
#ifdef XLZ_NO_BRANCHY // A/B builds
    return false;
#endif
    if (const char *e = getenv("XLZ_BRANCHY")) // development aid (tools/fuzz_gpu.py): "1" runs every compact launch through the
        return compact && e[0] == '1';         // branchy loop, "0" none -- the results must not depend on it
    return compact && decode_per_cu(max_lc_lp, n_units, num_cus, compact) >= kBranchyPerCu;
}

int launch_decode(const LaunchParams &p, int num_cus, void *stream, uint32_t max_grid)
{
    if (p.scratch) { // HBM-resident model
        uint32_t grid = big_model_grid(num_cus);
        if (grid > p.n_units) grid = p.n_units;
        if (max_grid && grid > max_grid) grid = max_grid; // (the caller allocated scratch slots for that many workgroups)
        if (grid == 0) return 0;
        hipLaunchKernelGGL(xlz_decode_kernel_hbm_model, dim3(grid), dim3(kWave), 0, (hipStream_t)stream, p);
        return hipGetLastError() == hipSuccess ? 0 : -3;
    }
    const bool compact = p.compact != 0;
    const uint32_t lds = decode_lds_bytes(p.max_lc_lp, compact);
    if (lds > kMaxLdsBytes) return -1;
    uint32_t grid = decode_grid(p.max_lc_lp, num_cus, p.call_units ? p.call_units : p.n_units, compact);
    if (grid > p.n_units) grid = p.n_units;
    if (grid == 0) return 0;
    const bool branchy = decode_branchy(p.max_lc_lp, num_cus, p.call_units ? p.call_units : p.n_units, compact);
    const void *fn = branchy ? reinterpret_cast<const void *>(xlz_decode_kernel_pb2_br)
                     : compact ? reinterpret_cast<const void *>(xlz_decode_kernel_pb2) : reinterpret_cast<const void *>(xlz_decode_kernel);
    if (lds > 64u * 1024u && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes) != hipSuccess) return -2;
    if (branchy)
        hipLaunchKernelGGL(xlz_decode_kernel_pb2_br, dim3(grid), dim3(kWave), lds, (hipStream_t)stream, p);
    else if (compact)
        hipLaunchKernelGGL(xlz_decode_kernel_pb2, dim3(grid), dim3(kWave), lds, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(xlz_decode_kernel, dim3(grid), dim3(kWave), lds, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

} // namespace xlz
