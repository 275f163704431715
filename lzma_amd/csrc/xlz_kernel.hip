// xlz_kernel.hip -- gfx950 (MI355X) decode kernel for batched LZMA / LZMA2.
//
// Mapping: ONE WAVE PER UNIT.  A unit is one independent LZMA1
// stream or one run of LZMA2 chunks.  The adaptive range decoder of
// decompress.go:8-1136 is a strictly serial dependent chain, so a unit cannot
// use lanes for its decisions; instead every lane of the wave executes the same
// wave-uniform code (range/code/state live in SGPRs, branches are scalar, no
// divergence), the unit's whole probability model (state.go:3-27) lives in LDS,
// and the 64 lanes are used where the work IS parallel: model initialisation,
// match copies, stored-chunk copies.  One single-wave workgroup owns one LDS
// model; 160 KiB / 15.6 KiB = 10 units per CU run concurrently and a persistent
// grid pulls units from an atomic queue (heaviest first).
//
// The sliding window (window.go) is the unit's own flat output range in HBM:
// distances are bounded by dictSize, so `out[pos - dist]` is the circular
// window's byte, and bytes "before the start" read as 0 exactly like the
// reference's zero-filled, not-yet-full window.
//
// file:line citations are into the reference repository (kulaginds/lzma).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "xlz_format.h"

namespace xlz {

typedef const __attribute__((address_space(4))) uint64_t *const_q_ptr; // scalar (SMEM) loads

#define RFL(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
__device__ __forceinline__ uint64_t rfl64(uint64_t v)
{
    return ((uint64_t)RFL((uint32_t)(v >> 32)) << 32) | RFL((uint32_t)v);
}

constexpr uint32_t kTop = 1u << 24;          // types.go:27
constexpr uint32_t kBitModelBits = 11;       // types.go:12
constexpr uint32_t kMoveBits = 5;            // types.go:13
constexpr uint32_t kProbInitPair = 0x04000400u; // two probs of 1024 (types.go:14)
constexpr uint32_t kEndPosModelIndex = 14;   // types.go:22
constexpr uint32_t kNumAlignBits = 4;        // types.go:20
constexpr uint32_t kMatchMinLen = 2;         // types.go:24

enum : int { RUN_END = 0, RUN_INPUT_EOF = 1, RUN_ERR_RESULT = 2, RUN_OUT_CAP = 3 };

// Everything a unit carries between packets / chunks.  All members are
// wave-uniform; after inlining they live in SGPRs.
struct Dec {
    // range decoder (range_decoder.go:7-13)
    uint32_t range, code;
    // compressed input: 8-byte scalar loads, one qword ahead
    const_q_ptr inq;
    uint32_t qidx;     // index of the next qword to fetch
    uint64_t cur, nxt; // cur holds `navail` unread bytes (low byte first)
    uint32_t navail;
    uint32_t in_remain; // bytes that may still be read (limitedByteReader, bytereader.go:7-28)
    // LZMA state (state.go:28-45)
    uint32_t state, rep0, rep1, rep2, rep3;
    uint32_t lc, lp_mask, pos_mask;
    bool size_defined;
    uint32_t bytes_left;
    // window (window.go:8-16) over the flat output
    uint32_t pos;       // bytes of output produced by this unit
    uint32_t wbase;     // output offset of the last dictionary reset (0 for LZMA1)
    uint32_t wpos;      // the reference's wrapped window.pos
    uint32_t dict_size; // window.size
    uint32_t out_cap;
    uint32_t prev_byte;  // byte at distance 1 (0 while the window is empty)
    uint32_t match_byte; // byte at distance rep0+1, valid right after a match / rep
    uint32_t stale;      // a copy reached in front of the current dictionary epoch
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
__device__ __forceinline__ void wave_copy(uint8_t *__restrict__ out, Dec &d, uint32_t dist, uint32_t len,
                                          uint32_t lane)
{
    const uint32_t pos = d.pos;
    const uint64_t lo = (uint64_t)dist + d.wbase; // virtual index >= wbase  <=>  pos + j >= lo
    const bool wrap = dist <= len;                // some i in [0, len] needs i mod dist
    if ((uint64_t)pos < lo) d.stale = 1;
    if (len < kWave) {
        uint32_t j = lane;
        if (wrap) j = umod_small(lane, dist);
        const uint64_t vs = (uint64_t)pos + j;
        const bool ok = vs >= lo;
        uint32_t b = out[ok ? vs - dist : 0];
        b = ok ? b : 0u;
        out[pos + lane] = (uint8_t)b;
        d.prev_byte = (uint32_t)__builtin_amdgcn_readlane((int)b, len - 1);
        d.match_byte = (uint32_t)__builtin_amdgcn_readlane((int)b, len);
    } else {
        for (uint32_t base = 0; base < len; base += kWave) {
            const uint32_t i = base + lane;
            uint32_t j = i;
            if (wrap) j = umod_small(i, dist);
            const uint64_t vs = (uint64_t)pos + j;
            const bool ok = vs >= lo;
            uint32_t b = out[ok ? vs - dist : 0];
            b = ok ? b : 0u;
            out[pos + i] = (uint8_t)b;
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

// ---- input ------------------------------------------------------------------
__device__ __forceinline__ void in_open(Dec &d, const uint8_t *arena, uint64_t off, uint32_t avail)
{
    const uint64_t a = (uint64_t)arena + off;
    const uint32_t sh = (uint32_t)(a & 7);
    d.inq = (const_q_ptr)(a & ~(uint64_t)7);
    d.cur = d.inq[0] >> (8 * sh);
    d.nxt = d.inq[1];
    d.qidx = 2;
    d.navail = 8 - sh;
    d.in_remain = avail;
}

// one byte from the source; false = io.EOF
#define IN_BYTE(D, B, EOF_STMT)                                                                   \
    do {                                                                                          \
        if ((D).in_remain == 0) { EOF_STMT; }                                                     \
        (B) = (uint32_t)(D).cur & 0xFFu;                                                          \
        (D).cur >>= 8;                                                                            \
        (D).in_remain--;                                                                          \
        if (--(D).navail == 0) {                                                                  \
            (D).cur = (D).nxt;                                                                    \
            (D).navail = 8;                                                                       \
            (D).nxt = (D).inq[(D).qidx++];                                                        \
        }                                                                                         \
    } while (0)

// ---- the binary decision (decompress.go:26-43,176-190; range_decoder.go:57-98) ----
#define NORMALIZE()                                                                               \
    do {                                                                                          \
        if (d.range < kTop) {                                                                     \
            uint32_t nb_;                                                                         \
            IN_BYTE(d, nb_, return RUN_INPUT_EOF);                                                \
            d.range <<= 8;                                                                        \
            d.code = (d.code << 8) | nb_;                                                         \
        }                                                                                         \
    } while (0)

#define BIT_NN(IDX, BIT)                                                                          \
    do {                                                                                          \
        const uint32_t i_ = (IDX);                                                                \
        uint32_t p_ = RFL(probs[i_]);                                                             \
        const uint32_t bound_ = (d.range >> kBitModelBits) * p_;                                  \
        if (d.code < bound_) {                                                                    \
            d.range = bound_;                                                                     \
            p_ += ((1u << kBitModelBits) - p_) >> kMoveBits;                                      \
            (BIT) = 0;                                                                            \
        } else {                                                                                  \
            d.range -= bound_;                                                                    \
            d.code -= bound_;                                                                     \
            p_ -= p_ >> kMoveBits;                                                                \
            (BIT) = 1;                                                                            \
        }                                                                                         \
        probs[i_] = (uint16_t)p_;                                                                 \
    } while (0)

#define BIT(IDX, B)                                                                               \
    do {                                                                                          \
        BIT_NN(IDX, B);                                                                           \
        NORMALIZE();                                                                              \
    } while (0)

// forward bit tree (bit_tree_decoder.go:18-40): M keeps the leading 1
#define TREE(BASE, NB, M)                                                                         \
    do {                                                                                          \
        (M) = 1;                                                                                  \
        _Pragma("unroll 1") for (uint32_t k_ = 0; k_ < (NB); k_++)                                \
        {                                                                                         \
            uint32_t tb_;                                                                         \
            BIT((BASE) + (M), tb_);                                                               \
            (M) = ((M) << 1) | tb_;                                                               \
        }                                                                                         \
    } while (0)

// reverse bit tree (bit_tree_decoder.go:42-70)
#define RTREE(BASE, NB, SYM)                                                                      \
    do {                                                                                          \
        uint32_t m_ = 1;                                                                          \
        (SYM) = 0;                                                                                \
        _Pragma("unroll 1") for (uint32_t k_ = 0; k_ < (NB); k_++)                                \
        {                                                                                         \
            uint32_t tb_;                                                                         \
            BIT((BASE) + m_, tb_);                                                                \
            m_ = (m_ << 1) | tb_;                                                                 \
            (SYM) |= tb_ << k_;                                                                   \
        }                                                                                         \
    } while (0)

// lenDecoder.Decode (len_decoder.go:34-60; decompress.go:218-429,870-1123)
#define LEN_DECODE(LBASE, LEN)                                                                    \
    do {                                                                                          \
        uint32_t c_, m__;                                                                         \
        BIT((LBASE) + LEN_CHOICE, c_);                                                            \
        if (c_ == 0) {                                                                            \
            TREE((LBASE) + LEN_LOW + (pos_state << 3), 3, m__);                                   \
            (LEN) = m__ - 8;                                                                      \
        } else {                                                                                  \
            BIT((LBASE) + LEN_CHOICE2, c_);                                                       \
            if (c_ == 0) {                                                                        \
                TREE((LBASE) + LEN_MID + (pos_state << 3), 3, m__);                               \
                (LEN) = m__;                                                                      \
            } else {                                                                              \
                TREE((LBASE) + LEN_HIGH, 8, m__);                                                 \
                (LEN) = 16 + m__ - 256;                                                           \
            }                                                                                     \
        }                                                                                         \
    } while (0)

// state.go:153-187
__device__ __forceinline__ uint32_t upd_literal(uint32_t s) { return s < 4 ? 0 : (s < 10 ? s - 3 : s - 6); }

// (*Reader1).decompress run to the end of the current LZMA chunk
// (decompress.go:8-1136).  Every mutation happens in the reference's order.
__device__ __forceinline__ int lzma_run(Dec &d, uint16_t *probs, uint8_t *__restrict__ out, uint32_t lane)
{
    for (;;) {
        uint32_t bit, length;

        // decompress.go:14-20
        if (d.size_defined && d.bytes_left == 0 && d.code == 0) return RUN_END;

        const uint32_t pos_state = d.wpos & d.pos_mask;              // :22
        const uint32_t state2 = (d.state << kPosBitsMax) + pos_state; // :23

        BIT(P_IS_MATCH + state2, bit); // :25-43,176-190
        if (bit == 0) {
            // ---- literal, decompress.go:44-175 ----
            if (d.size_defined && d.bytes_left == 0) return RUN_ERR_RESULT; // :45-47
            const uint32_t lit_state = ((d.wpos & d.lp_mask) << d.lc) + (d.prev_byte >> (8 - d.lc)); // :56
            const uint32_t lbase = P_LIT + kLitCoderSize * lit_state;                              // :57
            uint32_t symbol = 1;
            if (d.state >= 7) { // matched literal :59-114
                uint32_t mb = d.match_byte;
                do {
                    const uint32_t match_bit = (mb >> 7) & 1;
                    mb <<= 1;
                    BIT(lbase + ((1 + match_bit) << 8) + symbol, bit);
                    symbol = (symbol << 1) | bit;
                    if (match_bit != bit) break;
                } while (symbol < 0x100);
            }
            while (symbol < 0x100) { // :127-166
                BIT(lbase + symbol, bit);
                symbol = (symbol << 1) | bit;
            }
            symbol &= 0xFF;
            if (d.pos >= d.out_cap) return RUN_OUT_CAP;
            out[d.pos] = (uint8_t)symbol; // window.PutByte :168 (all lanes, same byte, same address)
            d.pos++;
            if (++d.wpos >= d.dict_size) d.wpos -= d.dict_size; // window.go:38-41
            d.prev_byte = symbol;
            d.state = upd_literal(d.state); // :171
            d.bytes_left--;                 // :172 (wraps harmlessly when the size is undefined)
            continue;
        }

        BIT(P_IS_REP + d.state, bit); // :195-213,669-683
        if (bit == 0) {
            // ---- simple match, :215-668 ----
            d.rep3 = d.rep2;
            d.rep2 = d.rep1;
            d.rep1 = d.rep0; // :216
            LEN_DECODE(P_LEN, length);
            d.state = d.state < 7 ? 7 : 10; // stateUpdateMatch :431
            const uint32_t len_state = length > 3 ? 3 : length;
            uint32_t pos_slot;
            TREE(P_POS_SLOT + (len_state << 6), 6, pos_slot); // :441-486
            pos_slot -= 64;
            if (pos_slot < 4) {
                d.rep0 = pos_slot; // :488-489
            } else {
                const uint32_t nbits = (pos_slot >> 1) - 1;
                uint32_t dist = (2 | (pos_slot & 1)) << nbits; // :491-492
                uint32_t sym;
                if (pos_slot < kEndPosModelIndex) {
                    RTREE(P_POS_DEC + dist - pos_slot, nbits, sym); // :495-546
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
                    RTREE(P_ALIGN, kNumAlignBits, sym); // :579-625
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
            // :651-653  rep0 >= size || !CheckDistance(rep0)   (window.go:89-91)
            {
                const bool is_full = (d.pos - d.wbase) >= d.dict_size;
                if (d.rep0 >= d.dict_size || !(is_full || d.rep0 <= d.wpos)) return RUN_ERR_RESULT;
            }
            length += kMatchMinLen; // :656
        } else {
            // ---- rep match, :685-1123 ----
            if (d.size_defined && d.bytes_left == 0) return RUN_ERR_RESULT; // :686-688
            if (d.pos == d.wbase) return RUN_ERR_RESULT;                    // window.IsEmpty :690-692
            BIT(P_IS_REP_G0 + d.state, bit);                                // :694-772
            if (bit == 0) {
                BIT(P_IS_REP0_LONG + state2, bit); // :715-756
                if (bit == 0) {                    // short rep :735-739
                    d.state = d.state < 7 ? 9 : 11;
                    if (d.pos >= d.out_cap) return RUN_OUT_CAP;
                    uint32_t dist = d.rep0 + 1;
                    if (dist == 0) dist = d.dict_size;
                    wave_copy(out, d, dist, 1, lane);
                    d.pos++;
                    if (++d.wpos >= d.dict_size) d.wpos -= d.dict_size;
                    d.bytes_left--;
                    continue;
                }
            } else {
                uint32_t dist;
                BIT_NN(P_IS_REP_G1 + d.state, bit); // :777-813
                if (bit == 0) {
                    dist = d.rep1;
                    d.rep1 = d.rep0;
                    d.rep0 = dist; // rotated before the normalise (:785-798)
                    NORMALIZE();
                } else {
                    NORMALIZE();
                    BIT_NN(P_IS_REP_G2 + d.state, bit); // :816-861
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
            LEN_DECODE(P_REP_LEN, length);
            d.state = d.state < 7 ? 8 : 11; // stateUpdateRep :933,1027,1103
            length += kMatchMinLen;
        }

        // window.CopyMatch + size bookkeeping, :657-668, 936-947, 1030-1041, 1106-1117
        {
            bool truncated = false;
            if (d.size_defined && d.bytes_left < length) {
                length = d.bytes_left;
                truncated = true;
            }
            bool overflow = false;
            if (length > d.out_cap - d.pos) {
                length = d.out_cap - d.pos;
                overflow = true;
            }
            uint32_t dist = d.rep0 + 1;
            if (dist == 0) dist = d.dict_size; // CopyMatch(0, n) re-reads the slot being written
            if (length > 0) wave_copy(out, d, dist, length, lane);
            d.pos += length;
            d.wpos += length;
            if (d.wpos >= d.dict_size) d.wpos -= d.dict_size; // window.go:67-71
            d.bytes_left -= length;
            if (overflow) return RUN_OUT_CAP;
            if (truncated) return RUN_ERR_RESULT;
        }
    }
}

// range_decoder.go:27-46.  0 ok, 1 io.EOF, 2 first byte != 0
__device__ __forceinline__ int rc_init(Dec &d)
{
    uint32_t b;
    d.range = 0xFFFFFFFFu;
    d.code = 0;
    IN_BYTE(d, b, return 1);
    if (b != 0) return 2;
    for (int i = 0; i < 4; i++) {
        IN_BYTE(d, b, return 1);
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
    d.size_defined = u != ~(uint64_t)0;
    d.bytes_left = (uint32_t)u; // host guarantees a defined size is < 4 GiB
}

__global__ __launch_bounds__(64) void xlz_decode_kernel(LaunchParams p)
{
    extern __shared__ __attribute__((aligned(16))) uint16_t probs[];
    const uint32_t lane = threadIdx.x;

    for (;;) {
        // dequeue: lane 0 bumps the head; lanes 1..63 add 0 to pad words of the same
        // 256-byte block (a branch here would make the whole loop divergent, see wave_copy)
        const uint32_t q = RFL(atomicAdd(p.queue + lane, lane == 0 ? 1u : 0u));
        if (q >= p.n_units) break; // every wave reaches this once the queue is drained

        const uint32_t ui = RFL(p.order[q]);
        const Unit *up = p.units + ui;
        Dec d;
        const uint64_t in_off = rfl64(up->in_off);
        const uint64_t out_off = rfl64(up->out_off);
        const uint64_t unpack = rfl64(up->unpack_size);
        const uint32_t in_len = RFL(up->in_len);
        d.out_cap = RFL(up->out_cap);
        d.dict_size = RFL(up->dict_size);
        const uint32_t lc = RFL(up->lc), lp = RFL(up->lp), pb = RFL(up->pb);
        uint8_t *__restrict__ out = p.out_arena + out_off;

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

        int32_t status;
        probs_reset(probs, num_probs(lc + lp), lane); // newState -> Reset (state.go:47-61)
        set_unpack_size(d, unpack);
        in_open(d, p.in_arena, in_off, in_len);
        const int ir = rc_init(d); // Reader1.initialize, reader1.go:149-159
        if (ir == 1) {
            status = ST_ERR_HEADER_EOF;
        } else if (ir == 2) {
            status = ST_ERR_RC_INIT;
        } else {
            const int r = lzma_run(d, probs, out, lane);
            status = r == RUN_END ? ST_OK
                                  : r == RUN_INPUT_EOF ? ST_OK_INPUT_EOF
                                                       : r == RUN_OUT_CAP ? ST_ERR_OUT_CAP : ST_ERR_RESULT;
        }
        {
            UnitResult res; // every lane stores the same 16 bytes
            res.out_len = d.pos;
            res.in_consumed = in_len - d.in_remain;
            res.status = status;
            res.aux = d.stale << 1;
            p.results[ui] = res;
        }
    }
}

uint32_t decode_lds_bytes(uint32_t max_lc_lp) { return num_probs(max_lc_lp) * 2u; }

int launch_decode(const LaunchParams &p, int num_cus, void *stream)
{
    const uint32_t lds = decode_lds_bytes(p.max_lc_lp);
    if (lds > kMaxLdsBytes) return -1;
    uint32_t per_cu = kMaxLdsBytes / lds;
    if (per_cu > 16) per_cu = 16;
    uint32_t grid = (uint32_t)num_cus * per_cu;
    if (grid > p.n_units) grid = p.n_units;
    if (grid == 0) return 0;
    if (lds > 64u * 1024u &&
        hipFuncSetAttribute(reinterpret_cast<const void *>(xlz_decode_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLdsBytes) != hipSuccess)
        return -2;
    hipLaunchKernelGGL(xlz_decode_kernel, dim3(grid), dim3(kWave), lds, (hipStream_t)stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

} // namespace xlz
