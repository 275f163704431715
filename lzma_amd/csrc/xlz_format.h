// xlz_format.h -- structures and constants shared by the host library and the
// gfx950 kernels.  Names follow the reference's domain (state.go, types.go).
#pragma once
#include <stdint.h>

namespace xlz {

// ---- probability-table layout in LDS (units: 16-bit probs) ------------------
// Same inventory as the reference's `state` struct (state.go:3-27); the order is
// ours.  Every bit-tree base is a multiple of 4 probs (8 bytes) so that a node's
// two children / four grandchildren sit in one aligned LDS word / double word.
constexpr uint32_t kNumStates = 12;      // types.go:17
constexpr uint32_t kPosBitsMax = 4;      // types.go:15
// Two layouts of the same inventory (round 5).  The FULL one has room for 2^4 posStates in every table that is indexed by one
// (isMatch, isRep0Long: (state << 4) + posState; the length coders' low / mid trees: posState << 3) -- pb <= 4 is what the
// reference accepts (reader1.go:210-221).  liblzma's and 7-Zip's default is pb = 2, which uses a quarter of those tables:
// the COMPACT layout has room for 2^2 posStates, the model of lc+lp = 3 is 5944 bytes instead of 7416 -- FIVE of gfx950's
// 1280-byte LDS granules instead of six -- and 24 workgroups fit a CU instead of 21 (+3.8 % on launches of many rounds,
// profiles/r05/ab_waves6.txt).  A launch uses it when every unit's pb is <= 2 (LZMA2: every chunk the host's scan saw; a
// chunk of a damaged stream that brings a larger pb stops its unit like one that brings a larger lc+lp does, and the
// stream is decoded again by the widest launch, which uses the full layout).  Only the table BASES differ between the two.
template <bool COMPACT> struct ModelLayout {
    static constexpr uint32_t kPosBits = COMPACT ? 2u : kPosBitsMax; // posState bits the tables have room for
    static constexpr uint32_t kPosStates = 1u << kPosBits;
    static constexpr uint32_t P_IS_MATCH = 0;                               // [12 << kPosBits]  index (state << kPosBits) + posState
    static constexpr uint32_t P_IS_REP = kNumStates << kPosBits;            // [12]
    static constexpr uint32_t P_IS_REP_G0 = P_IS_REP + 12;                  // [12]
    static constexpr uint32_t P_IS_REP_G1 = P_IS_REP + 24;                  // [12]
    static constexpr uint32_t P_IS_REP_G2 = P_IS_REP + 36;                  // [12]
    static constexpr uint32_t P_IS_REP0_LONG = P_IS_REP + 48;               // [12 << kPosBits]
    static constexpr uint32_t P_POS_SLOT = P_IS_REP0_LONG + (kNumStates << kPosBits); // [4][64]
    static constexpr uint32_t P_POS_DEC = P_POS_SLOT + 256;                 // [115] (+1 pad) posDecoders, state.go:7
    static constexpr uint32_t P_ALIGN = P_POS_DEC + 116;                    // [16]
    // length coder: choice, choice2, 2 pad, low[kPosStates][8], mid[kPosStates][8], high[256]
    static constexpr uint32_t LEN_CHOICE = 0, LEN_CHOICE2 = 1, LEN_LOW = 4;
    static constexpr uint32_t LEN_MID = LEN_LOW + (8u << kPosBits);
    static constexpr uint32_t LEN_HIGH = LEN_LOW + (16u << kPosBits);
    static constexpr uint32_t LEN_CODER_SIZE = LEN_HIGH + 256;
    static constexpr uint32_t P_LEN = P_ALIGN + 16;
    static constexpr uint32_t P_REP_LEN = P_LEN + LEN_CODER_SIZE;
    // The rep-length coder's HIGH tree (256 probs: rep matches of 18 bytes and more, rare) is not in LDS: it is the first
    // kRepHigh entries of the model's HBM part (in front of the matched-literal tables).
    static constexpr uint32_t REP_LEN_LDS_SIZE = LEN_HIGH;                  // choice, choice2, 2 pad, low, mid
    static constexpr uint32_t P_LIT = P_REP_LEN + REP_LEN_LDS_SIZE;
};
using FullLayout = ModelLayout<false>;
constexpr uint32_t kCompactPosBits = ModelLayout<true>::kPosBits;
// the full layout's constants by their old names (the host side and everything that is not a decode launch use them)
constexpr uint32_t P_IS_MATCH = FullLayout::P_IS_MATCH, P_IS_REP = FullLayout::P_IS_REP, P_IS_REP_G0 = FullLayout::P_IS_REP_G0,
                   P_IS_REP_G1 = FullLayout::P_IS_REP_G1, P_IS_REP_G2 = FullLayout::P_IS_REP_G2, P_IS_REP0_LONG = FullLayout::P_IS_REP0_LONG,
                   P_POS_SLOT = FullLayout::P_POS_SLOT, P_POS_DEC = FullLayout::P_POS_DEC, P_ALIGN = FullLayout::P_ALIGN,
                   P_LEN = FullLayout::P_LEN, P_REP_LEN = FullLayout::P_REP_LEN, P_LIT = FullLayout::P_LIT;
constexpr uint32_t LEN_CHOICE = 0, LEN_CHOICE2 = 1, LEN_LOW = 4, LEN_MID = FullLayout::LEN_MID, LEN_HIGH = FullLayout::LEN_HIGH,
                   LEN_CODER_SIZE = FullLayout::LEN_CODER_SIZE, REP_LEN_LDS_SIZE = FullLayout::REP_LEN_LDS_SIZE;
static_assert(P_IS_REP == 192 && P_IS_REP0_LONG == 240 && P_POS_SLOT == 432 && P_POS_DEC == 688 && P_ALIGN == 804 && P_LEN == 820 &&
                  P_REP_LEN == 1336 && P_LIT == 1596 && LEN_MID == 132 && LEN_HIGH == 260, "the full layout is round 3's");
static_assert(ModelLayout<true>::P_LIT == 924 && ModelLayout<true>::P_LEN == 532 && ModelLayout<true>::P_REP_LEN == 856, "compact layout");
constexpr uint32_t kRepHigh = 256;                 // HBM part: entries [0, 256) = rep-length high tree (index = tree slot)
// The reference's literal coder has 0x300 probs per literal state (state.go:4,49): 0x100 for the
// plain 8-bit tree and 0x200 used only by the FIRST literal after a match ("matched literal",
// decompress.go:59-114).  The plain part stays in the LDS model; the matched part lives in an
// HBM scratch slot of the workgroup, fetched by one 8-lane gather when needed.  Halving the
// LDS model is what lets 16 instead of 10 units be resident per CU.
constexpr uint32_t kLitPlain = 0x100;   // per literal state, in LDS at P_LIT
constexpr uint32_t kLitMatched = 0x200; // per literal state, in HBM: index (matchBit << 8) + symbol

static inline constexpr uint32_t num_probs(uint32_t lc_plus_lp, bool compact = false)
{
    return (compact ? ModelLayout<true>::P_LIT : FullLayout::P_LIT) + (kLitPlain << lc_plus_lp);
}
// the model's HBM part: the rep-length high tree, then the matched-literal tables
static inline constexpr uint32_t num_matched_probs(uint32_t lc_plus_lp) { return kRepHigh + (kLitMatched << lc_plus_lp); }

constexpr uint32_t kMaxLdsBytes = 160u * 1024u; // MI355X LDS per CU
constexpr uint32_t kWave = 64;
constexpr uint32_t kMaxLcLpLds = 8;  // (1852 + (0x100 << 8)) probs = 132 KiB still fits one CU's LDS
constexpr uint32_t kMaxLcLp = 12;    // reference limit: lc <= 8, lp <= 4 (reader1.go:210-221)

// ---- unit of work: one LZMA1 stream, or one run of LZMA2 chunks ------------
enum : uint32_t {
    UNIT_LZMA1 = 0, // payload starts at the first range-coder byte (header already parsed)
    UNIT_LZMA2 = 2  // payload starts at an LZMA2 control byte; the wave walks the chunks
};

enum : uint32_t {
    UNIT_F_LAST = 1,        // the parent stream's input ends with this unit
    UNIT_F_HAVE_READER = 2, // an LZMA chunk precedes this unit in its stream: Reader2.lzmaReader exists
    UNIT_F_BIG_MODEL = 4,   // model does not fit LDS: decoded by the HBM-model launch
    UNIT_F_RESUME = 8,      // continue from the UnitState saved by an earlier launch (pull readers)
    UNIT_F_MORE_INPUT = 16, // in_len is a window of a longer input: pause (not EOF) when it runs low
    UNIT_F_NOT_FIRST = 32,  // LZMA2: earlier units of the stream precede this one (their bytes may be read as
                            // stale window content, window.go:135-140)
    UNIT_F_RESET_MODEL = 64, // UNIT_F_RESUME + (*Reader1).Reset (reader1.go:161-164): state.Reset before continuing
    UNIT_F_REOPEN = 128      // UNIT_F_RESUME + (*Reader1).Reopen (reader1.go:166-176): the input is a NEW stream
                             // (range coder re-initialised, unpack_size applies), window and model live on
};

struct Unit {
    uint64_t in_off;      // byte offset of the payload in the input arena
    uint64_t out_off;     // byte offset of this unit's output in the output arena
    uint64_t unpack_size; // LZMA1: header size, all-ones = undefined (state.go:135-151)
    uint64_t state;       // resumable units: device address of the unit's UnitState, else 0
    uint32_t in_len;      // payload bytes available to this unit
    uint32_t out_cap;     // bytes this unit may write
    uint32_t dict_size;   // window size (already clamped the way the reference clamps it)
    uint32_t stream;      // index of the parent stream
    uint8_t lc, lp, pb, kind; // LZMA2: lc = the largest lc+lp the host's header scan saw, lp = pb = 0
    uint32_t flags;
    uint32_t expect_out;  // LZMA2: output the host scan predicts for this unit
    uint32_t pause_at;    // resumable units: stop at the first packet / chunk boundary with pos >= pause_at
    uint32_t rebase;      // UNIT_F_RESUME: the host moved the window down by this many bytes (pos, wbase follow)
    uint32_t in_skip;     // UNIT_F_RESUME: input bytes consumed before in_off (the input window moved)
    uint32_t pad[1];
};
static_assert(sizeof(Unit) == 80, "Unit layout is shared with the host");

struct UnitResult {
    uint64_t out_len;     // bytes in the unit's output range after this launch (pos)
    uint64_t in_consumed; // input bytes pulled from the unit's source so far
    int32_t status;
    uint32_t aux;         // AUX_* bits
    uint32_t t_start, t_end; // 100 MHz device clock (s_memrealtime), low 32 bits: slot occupancy, tail
};
static_assert(sizeof(UnitResult) == 32, "UnitResult layout is shared with the host");

enum : uint32_t {
    AUX_END_MARK = 1,  // LZMA2 unit ended on an end-of-stream control byte
    AUX_STALE = 2,     // a copy reached in front of the current dictionary epoch and the launch had no
                       // epoch table: the bytes read there are NOT exact, the host decodes the stream
                       // again with one (window.go:135-140 does not clear the buffer)
    AUX_NEED_INPUT = 4, // paused because the input window ran low (UNIT_F_MORE_INPUT)
    AUX_GROW = 8,       // paused in front of an LZMA2 chunk whose properties need a larger model than the unit's state
                        // block holds: lc+lp wanted in bits 8..11
    AUX_GROW_SHIFT = 8,
    AUX_GROW_MASK = 0xF00,
    AUX_SHADOW = 16     // paused in front of an LZMA2 chunk that resets the dictionary behind a non-empty epoch while the
                        // session has no window image yet: the host allocates it (dictSize bytes) and resumes -- a stream
                        // whose only dictionary reset is its first chunk's never pays for one
};

// device-side status values = include/xlz.h
enum : int32_t {
    ST_OK = 0,
    ST_OK_INPUT_EOF = 1,
    ST_PAUSED = 2, // resumable unit stopped at pause_at / for more input; state saved
    ST_ERR_RESULT = -1,
    ST_ERR_PROPS = -2,
    ST_ERR_HEADER_EOF = -3,
    ST_ERR_RC_INIT = -4,
    ST_ERR_UNEXPECTED_EOF = -5,
    ST_ERR_OUT_CAP = -6,
    ST_ERR_UNSUPPORTED = -9
};

// Saved decoder state of a resumable unit (HBM): 64 words of registers, then the LDS model, then the
// matched-literal half of the model (which lives here for the unit's whole life, not in a
// workgroup slot).
constexpr uint32_t kStateWords = 64;
constexpr uint32_t kStateShadowWord = 32; // words 32, 33 of the header: device address of the session's window image
                                          // (LZMA2 readers; written by the host, read by the wave)
static inline constexpr uint32_t state_probs_off() { return kStateWords * 4; }
static inline constexpr uint32_t state_mprobs_off(uint32_t lc_lp) { return state_probs_off() + ((num_probs(lc_lp) * 2 + 15) & ~15u); }
static inline constexpr uint32_t state_bytes(uint32_t lc_lp) { return state_mprobs_off(lc_lp) + num_matched_probs(lc_lp) * 2; }

// one dictionary epoch of an LZMA2 unit that a later epoch may still read (exact launches only)
struct Epoch {
    uint32_t start, len; // output offset of the epoch's first byte, bytes it wrote
};
constexpr uint32_t kMaxEpochs = 8192; // strictly decreasing lengths: needs > 32 MiB of output to overflow

struct LaunchParams {
    const uint8_t *in_arena;
    uint8_t *out_arena;
    const Unit *units;
    const uint32_t *order; // work-queue order (heaviest first)
    UnitResult *results;
    uint32_t *queue;       // 64 words of queue (word 0 = head), then 256 bytes of dump space for predicated stores
    uint32_t n_units;
    uint32_t max_lc_lp;    // sizes the model: dynamic LDS, or one scratch slot per workgroup
    // "big model" launch (lc+lp too large for 160 KiB of LDS; the reference allows lc<=8, lp<=4):
    // the model lives in HBM, one slot of scratch_stride probs per workgroup, and only the
    // plain C++ packet decoder runs.  nullptr for the normal LDS launch.
    uint16_t *scratch;
    uint32_t scratch_stride;
    uint32_t order_base;   // first entry of `order` this launch works on
    // matched-literal probabilities: one slot of mlit_stride probs per workgroup (normal launch;
    // the HBM-model launch keeps them behind the model inside its scratch slot)
    uint16_t *mlit;
    uint32_t mlit_stride;
    // exact launches (streams that read across an LZMA2 dictionary reset): kMaxEpochs entries per
    // workgroup; nullptr in ordinary launches, which only flag AUX_STALE
    Epoch *epochs;
    // one word per hardware wave slot of the device (kPrioTabWords): the compressed bytes the wave in
    // that slot still has to decode, 0 when idle -- the waves of a SIMD rank themselves by it (rotate_priority)
    uint32_t *prio_tab;
    // Sliced launches (a call of ONE wave round whose download overlaps its decode, xlz_host.hip: decode_batch):
    // launch k of a sequence advances every unit to the first packet / chunk boundary at or behind
    // slice_bound(out_cap, slice_frac) and saves its state in the unit's state block (Unit.state; all blocks sized for
    // max_lc_lp); launch k + 1 resumes the units that paused (results[].status == ST_PAUSED) and skips the others.  The units
    // are batch units: whole input and output in the arenas, no requests to the host (AUX_GROW, AUX_SHADOW), copies behind a
    // dictionary reset are flagged (AUX_STALE) and settled by the exact re-run.  Between two launches the bytes in front of the bound are final and visible
    // (a kernel boundary), so the host downloads them while the next launch decodes: the reference's Read pump in
    // batch form (reader1.go:223-254 drains window.pending while the decoder keeps its state, window.go:97-133).
    uint32_t slice_frac;   // 0: not a sliced launch; else the bound of this launch in 1/65536 of out_cap (65536: run to the end)
    uint32_t slice_k;      // index of the launch in its sequence (0: every unit starts, > 0: paused units resume)
    uint32_t head_frac;    // slice_k == 0 only: the launch sees the first slice_head(in_len, head_frac) bytes of every unit's
                           // input (the rest is still being uploaded) and pauses a unit that runs out of them; 0: all of it
    uint32_t compact;      // != 0: the LDS model uses the COMPACT layout (ModelLayout<true>: every unit's pb <= 2); the host sets it
                           // when every unit of the launch allows it, never for sessions or the HBM-model launch
    uint32_t pad_;
    uint32_t call_units;   // != 0: the launch is one piece of a pipelined call of that many units in all, whose pieces' launches
                           // follow each other without a gap (two streams): workgroups per CU as ONE launch over call_units units
                           // would take (decode_per_cu), whatever this piece's n_units
};
constexpr uint32_t kSliceOne = 65536;
// output position (relative to the unit) at which a sliced launch with this slice_frac pauses a unit of out_cap bytes
static inline constexpr uint32_t slice_bound(uint32_t out_cap, uint32_t frac)
{
    return frac >= kSliceOne ? 0xFFFFFFFFu : (uint32_t)(((uint64_t)out_cap * frac) >> 16) & ~255u;
}
// input bytes of a unit that are on the device when the first launch of a sequence starts: its share + 4 KiB; an LZMA2
// unit + 72 KiB (its wave wants a whole chunk -- 6 bytes of header, 64 KiB of payload -- in front of it before it starts one:
// a small unit would only pause); a multiple of 256, or everything
static inline constexpr uint32_t slice_head(uint32_t in_len, uint32_t frac, bool lzma2)
{
    if (frac == 0 || frac >= kSliceOne) return in_len;
    const uint64_t h = ((((uint64_t)in_len * frac) >> 16) + (lzma2 ? 73728u : 4096u)) & ~(uint64_t)255;
    return h < in_len ? (uint32_t)h : in_len;
}

constexpr uint32_t kPrioTabWords = 1u << 20; // XCC_ID[3:0] : HW_ID[15:0]

// One piece of a sliced batch's download: `len` bytes at `src_off` of the output arena (what one launch finished of one
// unit) go to `pack_off` of the launch's packed image (equal modulo 256), which ONE linear copy then takes to the host:
// thousands of strided copies cost this stack 12 us each, a 2-D copy wants equal rows, a packed image wants neither
// (tools/ubench/copy2d.hip, profiles/r05/copy2d.txt).
struct SlicePiece {
    uint64_t src_off, pack_off;
    uint32_t len, unit;
};
static_assert(sizeof(SlicePiece) == 24, "SlicePiece layout is shared with the host");

// implemented in xlz_kernel.hip
// scatter = false: arena -> packed image (download); true: packed image -> arena (the upload's heads / tails)
int launch_gather(const SlicePiece *pieces, uint32_t n_pieces, uint8_t *arena, uint8_t *pack, uint64_t pack_bytes, int num_cus,
                  void *stream /* hipStream_t */, bool scatter = false);
int launch_decode(const LaunchParams &p, int num_cus, void *stream /* hipStream_t */, uint32_t max_grid = 0 /* HBM-model launch: at most that many workgroups */);
uint32_t decode_lds_bytes(uint32_t max_lc_lp, bool compact = false);
uint32_t big_model_grid(int num_cus);
bool decode_branchy(uint32_t max_lc_lp, int num_cus, uint32_t n_units, bool compact); // the launch runs the branchy loop (xlz_kernel.hip)
uint32_t decode_grid(uint32_t max_lc_lp, int num_cus, uint32_t n_units, bool compact = false); // resident workgroups of the LDS-model launch over n_units units (~0u: the most)

} // namespace xlz
