/*
 * xlz_oracle.c -- CPU oracle: plain-C restatement of kulaginds/lzma's decode path.
 *
 * TEST INFRASTRUCTURE ONLY (see xlz_oracle.h).  Not linked into the product.
 *
 * The reference hand-inlines every range-coder decision into one 1100-line loop
 * (decompress.go:8-1136).  This file states the same arithmetic with one macro
 * per primitive, keeping the ORDER of state mutations identical so that even an
 * aborted packet (input exhausted mid-packet inside an LZMA2 chunk) leaves the
 * same probabilities / reps / state behind as the reference would.
 *
 * file:line citations are into /root/reference (kulaginds/lzma @ 2025-06-14).
 */
#include "xlz_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef uint16_t prob_t; /* types.go:98 */

/* types.go:12-36 */
enum {
    K_BIT_MODEL_BITS = 11,
    K_MOVE_BITS = 5,
    K_PROB_INIT = 1024,
    K_POS_BITS_MAX = 4,
    K_NUM_STATES = 12,
    K_LEN_TO_POS_STATES = 4,
    K_ALIGN_BITS = 4,
    K_END_POS_MODEL = 14,
    K_FULL_DISTANCES = 128,
    K_MATCH_MIN_LEN = 2
};
#define K_TOP ((uint32_t)1 << 24)
#define LZMA_DIC_MIN ((uint32_t)1 << 12) /* types.go:8 */

/* ---------------------------------------------------------------- source ---- */
/* io.ByteReader over a memory buffer with an optional limitedByteReader on top
 * (bytereader.go:7-28). limit < 0 means "no limiter installed". */
typedef struct {
    const uint8_t *p;
    size_t len, pos;
    int64_t limit;
} src_t;

static inline int src_byte(src_t *s, uint8_t *b)
{
    if (s->limit >= 0) {
        if (s->limit <= 0) return 0; /* bytereader.go:20-22 */
        if (s->pos >= s->len) return 0;
        *b = s->p[s->pos++];
        s->limit--; /* bytereader.go:24-26 */
        return 1;
    }
    if (s->pos >= s->len) return 0;
    *b = s->p[s->pos++];
    return 1;
}

/* ---------------------------------------------------------------- window ---- */
/* window.go:8-29: circular dictionary of exactly dictSize bytes, zero-filled at
 * allocation (Go make()).  Every byte put into the window is also appended to
 * the caller's flat output: that is what Read/ReadPending deliver for reads
 * inside the reference's safe envelope (SURVEY parity note 6). */
typedef struct {
    uint8_t *buf;
    uint32_t pos, size;
    int is_full;
    uint8_t *out;
    size_t out_cap;
    uint64_t out_len;
    int overflow;
} win_t;

static int win_init(win_t *w, uint32_t dict_size, uint8_t *out, size_t out_cap)
{
    memset(w, 0, sizeof *w);
    w->buf = (uint8_t *)calloc(dict_size, 1);
    if (!w->buf) return 0;
    w->size = dict_size;
    w->out = out;
    w->out_cap = out_cap;
    return 1;
}

static inline void win_emit(win_t *w, uint8_t b)
{
    if (w->out_len < w->out_cap)
        w->out[w->out_len] = b;
    else
        w->overflow = 1;
    w->out_len++;
}

/* window.go:31-42 */
static inline void win_put(win_t *w, uint8_t b)
{
    w->buf[w->pos] = b;
    w->pos++;
    if (w->pos >= w->size) {
        w->pos -= w->size;
        w->is_full = 1;
    }
    win_emit(w, b);
}

/* window.go:44-53 (uint32 arithmetic, as in Go) */
static inline uint8_t win_get(const win_t *w, uint32_t dist)
{
    uint32_t i = w->pos - dist;
    if (dist > w->pos) i = w->size - dist + w->pos;
    return w->buf[i];
}

/* window.go:55-87: byte-wise, overlap-replicating, two wrap checks */
static void win_copy(win_t *w, uint32_t dist, uint32_t len)
{
    uint32_t from, to = w->pos, limit = w->size;
    if (dist <= w->pos)
        from = w->pos - dist;
    else
        from = w->size - dist + w->pos;
    w->pos += len;
    if (w->pos >= w->size) {
        w->pos -= w->size;
        w->is_full = 1;
    }
    for (; len > 0; len--) {
        uint8_t b = w->buf[from];
        w->buf[to] = b;
        win_emit(w, b);
        from++;
        to++;
        if (from == limit) from -= w->size;
        if (to == limit) to -= w->size;
    }
}

/* window.go:89-95 */
static inline int win_check_distance(const win_t *w, uint32_t d) { return w->is_full || d <= w->pos; }
static inline int win_is_empty(const win_t *w) { return w->pos == 0 && !w->is_full; }
/* window.go:135-140: Reset does NOT clear the buffer */
static inline void win_reset(win_t *w)
{
    w->pos = 0;
    w->is_full = 0;
}

/* ----------------------------------------------------------------- state ---- */
/* state.go:3-45 */
typedef struct {
    prob_t *lit;
    size_t lit_cap;
    prob_t pos_slot[K_LEN_TO_POS_STATES][64];
    prob_t pos_dec[1 + K_FULL_DISTANCES - K_END_POS_MODEL];
    prob_t align[1 << K_ALIGN_BITS];
    prob_t len_choice, len_choice2;
    prob_t len_low[1 << K_POS_BITS_MAX][8], len_mid[1 << K_POS_BITS_MAX][8], len_high[256];
    prob_t rep_choice, rep_choice2;
    prob_t rep_low[1 << K_POS_BITS_MAX][8], rep_mid[1 << K_POS_BITS_MAX][8], rep_high[256];
    prob_t is_match[K_NUM_STATES << K_POS_BITS_MAX];
    prob_t is_rep[K_NUM_STATES], is_rep_g0[K_NUM_STATES], is_rep_g1[K_NUM_STATES],
        is_rep_g2[K_NUM_STATES];
    prob_t is_rep0_long[K_NUM_STATES << K_POS_BITS_MAX];
    int unpack_size_defined;
    uint8_t lc, pb, lp;
    uint64_t bytes_left;
    uint32_t pos_mask;
    uint32_t rep0, rep1, rep2, rep3;
    uint32_t state, pos_state;
} st_t;

static inline void fill(prob_t *p, size_t n) /* prob.go:3-7 */
{
    for (size_t i = 0; i < n; i++) p[i] = K_PROB_INIT;
}
#define FILL(a) fill((prob_t *)(a), sizeof(a) / sizeof(prob_t))

/* state.go:79-121 */
static void st_reset(st_t *s)
{
    fill(s->lit, (size_t)0x300 << (s->lc + s->lp));
    FILL(s->pos_slot);
    FILL(s->align);
    FILL(s->pos_dec);
    FILL(s->is_match);
    FILL(s->is_rep);
    FILL(s->is_rep_g0);
    FILL(s->is_rep_g1);
    FILL(s->is_rep_g2);
    FILL(s->is_rep0_long);
    s->len_choice = s->len_choice2 = K_PROB_INIT;
    FILL(s->len_high);
    FILL(s->len_low);
    FILL(s->len_mid);
    s->rep_choice = s->rep_choice2 = K_PROB_INIT;
    FILL(s->rep_high);
    FILL(s->rep_low);
    FILL(s->rep_mid);
    s->rep0 = s->rep1 = s->rep2 = s->rep3 = 0;
    s->state = 0;
    s->pos_state = 0;
}

/* state.go:47-77 (newState and Renew: same effect on a fresh or recycled state) */
static int st_renew(st_t *s, uint8_t lc, uint8_t pb, uint8_t lp)
{
    size_t n = (size_t)0x300 << (lc + lp);
    s->lc = lc;
    s->pb = pb;
    s->lp = lp;
    s->pos_mask = ((uint32_t)1 << pb) - 1;
    if (n > s->lit_cap) {
        free(s->lit);
        s->lit = (prob_t *)malloc(n * sizeof(prob_t));
        if (!s->lit) return 0;
        s->lit_cap = n;
    }
    st_reset(s);
    return 1;
}

/* state.go:123-151 */
static void st_set_unpack_size(st_t *s, uint64_t u)
{
    int defined = 0;
    uint64_t t = u;
    s->bytes_left = u;
    for (int i = 0; i < 8; i++) {
        if ((t & 0xFF) != 0xFF) defined = 1;
        t >>= 8;
    }
    s->unpack_size_defined = defined;
}

/* state.go:153-187 */
static inline uint32_t upd_literal(uint32_t s) { return s < 4 ? 0 : (s < 10 ? s - 3 : s - 6); }
static inline uint32_t upd_match(uint32_t s) { return s < 7 ? 7 : 10; }
static inline uint32_t upd_rep(uint32_t s) { return s < 7 ? 8 : 11; }
static inline uint32_t upd_shortrep(uint32_t s) { return s < 7 ? 9 : 11; }

/* reader1.go:210-221 */
int xlzo_decode_prop(uint8_t d, uint8_t *lc, uint8_t *pb, uint8_t *lp)
{
    if (d >= 9 * 5 * 5) return 0;
    *lc = d % 9;
    d /= 9;
    *pb = d / 5;
    *lp = d % 5;
    return 1;
}

/* reader1.go:193-208 (the > lzmaDicMax branch is unreachable for a uint32) */
uint32_t xlzo_decode_dict_size(const uint8_t p[4])
{
    uint32_t d = 0;
    for (int i = 0; i < 4; i++) d |= (uint32_t)p[i] << (8 * i);
    if (d < LZMA_DIC_MIN) d = LZMA_DIC_MIN;
    return d;
}

/* reader2.go:296-298 (uint32 shift; wraps exactly like Go for silly inputs) */
uint32_t xlzo_decode_dict_size2(uint8_t b)
{
    unsigned sh = (unsigned)(b / 2 + 11);
    uint32_t base = 2u | (b & 1u);
    return sh >= 32 ? 0u : base << sh;
}

/* reader1.go:178-191 */
uint64_t xlzo_decode_unpack_size(const uint8_t h[8])
{
    uint64_t u = 0;
    for (int i = 0; i < 8; i++) u |= (uint64_t)h[i] << (8 * i);
    return u;
}

/* --------------------------------------------------------------- decoder ---- */
typedef struct {
    src_t src;
    win_t win;
    st_t st;
    uint32_t range, code; /* range_decoder.go:10-11 */
} dec_t;

enum { RUN_END = 0, RUN_INPUT_EOF = 1, RUN_ERR_RESULT = 2, RUN_OUT_CAP = 3 };

/* range_decoder.go:27-46.  returns 0 ok, 1 EOF, 2 first byte != 0 */
static int rc_init(dec_t *d)
{
    uint8_t b;
    if (!src_byte(&d->src, &b)) return 1;
    if (b != 0) return 2;
    for (int i = 0; i < 4; i++) {
        if (!src_byte(&d->src, &b)) return 1;
        d->code = (d->code << 8) | b;
    }
    return 0;
}

/* The binary decision, decompress.go:26-43 + 176-190 (template repeated ~60x in
 * the reference; readable form range_decoder.go:57-98).  BIT_NN leaves the
 * normalisation to the caller because the reference sometimes mutates state
 * between the probability update and the normalisation (decompress.go:785-798). */
#define NORMALIZE()                                                                               \
    do {                                                                                          \
        if (range < K_TOP) {                                                                      \
            uint8_t b_;                                                                           \
            if (!src_byte(&d->src, &b_)) return RUN_INPUT_EOF; /* decompress.go:35-38 */          \
            range <<= 8;                                                                          \
            code = (code << 8) | b_;                                                              \
        }                                                                                         \
    } while (0)

#define BIT_NN(P, BIT)                                                                            \
    do {                                                                                          \
        prob_t *p_ = (P);                                                                         \
        uint32_t v_ = *p_;                                                                        \
        uint32_t bound_ = (range >> K_BIT_MODEL_BITS) * v_;                                       \
        if (code < bound_) {                                                                      \
            *p_ = (prob_t)(v_ + (((1u << K_BIT_MODEL_BITS) - v_) >> K_MOVE_BITS));                \
            range = bound_;                                                                       \
            (BIT) = 0;                                                                            \
        } else {                                                                                  \
            *p_ = (prob_t)(v_ - (v_ >> K_MOVE_BITS));                                             \
            code -= bound_;                                                                       \
            range -= bound_;                                                                      \
            (BIT) = 1;                                                                            \
        }                                                                                         \
    } while (0)

#define BIT(P, B)                                                                                 \
    do {                                                                                          \
        BIT_NN(P, B);                                                                             \
        NORMALIZE();                                                                              \
    } while (0)

/* forward bit tree of NB bits (bit_tree_decoder.go:18-40, inlined e.g. at
 * decompress.go:237-283): result in M still carries the leading 1 */
#define TREE(PROBS, NB, M)                                                                        \
    do {                                                                                          \
        (M) = 1;                                                                                  \
        for (int i_ = 0; i_ < (NB); i_++) {                                                       \
            uint32_t b__;                                                                         \
            BIT(&(PROBS)[(M)], b__);                                                              \
            (M) = ((M) << 1) | b__;                                                               \
        }                                                                                         \
    } while (0)

/* reverse bit tree (bit_tree_decoder.go:42-70; decompress.go:495-546,580-625) */
#define RTREE(PROBS, NB, SYM)                                                                     \
    do {                                                                                          \
        uint32_t m_ = 1;                                                                          \
        (SYM) = 0;                                                                                \
        for (uint32_t i_ = 0; i_ < (NB); i_++) {                                                  \
            uint32_t b__;                                                                         \
            BIT(&(PROBS)[m_], b__);                                                               \
            m_ = (m_ << 1) | b__;                                                                 \
            (SYM) |= b__ << i_;                                                                   \
        }                                                                                         \
    } while (0)

/* len_decoder.go:34-60 / decompress.go:218-429, 870-1123.  LEN is the raw
 * length (without kMatchMinLen). */
#define LEN_DECODE(CH, CH2, LOW, MID, HIGH, LEN)                                                  \
    do {                                                                                          \
        uint32_t c_, m__;                                                                         \
        BIT(&(CH), c_);                                                                           \
        if (c_ == 0) {                                                                            \
            TREE((LOW)[s->pos_state], 3, m__);                                                    \
            (LEN) = m__ - 8;                                                                      \
        } else {                                                                                  \
            BIT(&(CH2), c_);                                                                      \
            if (c_ == 0) {                                                                        \
                TREE((MID)[s->pos_state], 3, m__);                                                \
                (LEN) = 8 + m__ - 8;                                                              \
            } else {                                                                              \
                TREE((HIGH), 8, m__);                                                             \
                (LEN) = 16 + m__ - 256;                                                           \
            }                                                                                     \
        }                                                                                         \
    } while (0)

/* (*Reader1).decompress driven to the end of the stream, i.e. what
 * io.Copy(dst, reader1) makes of it (decompress.go:8-1136, reader1.go:223-254). */
static int lzma_run(dec_t *d)
{
    st_t *s = &d->st;
    win_t *w = &d->win;
    uint32_t range = d->range, code = d->code;
    int rc = RUN_END;

    for (;;) {
        uint32_t bit, state2, length;

        if (w->overflow) {
            rc = RUN_OUT_CAP;
            break;
        }
        /* decompress.go:14-20 */
        if (s->unpack_size_defined && s->bytes_left == 0 && code == 0) break;

        s->pos_state = w->pos & s->pos_mask;                 /* :22 (wrapped pos) */
        state2 = (s->state << K_POS_BITS_MAX) + s->pos_state; /* :23 */

        BIT(&s->is_match[state2], bit); /* :25-43,176-190 */
        if (bit == 0) {
            /* literal, decompress.go:44-175 */
            uint32_t prev = 0, symbol = 1, lit_state;
            prob_t *probs;
            if (s->unpack_size_defined && s->bytes_left == 0) return RUN_ERR_RESULT; /* :45-47 */
            if (!win_is_empty(w)) prev = win_get(w, 1);                              /* :50-53 */
            lit_state = ((w->pos & (((uint32_t)1 << s->lp) - 1)) << s->lc) + (prev >> (8 - s->lc));
            probs = &s->lit[(size_t)0x300 * lit_state]; /* :56-57 */
            if (s->state >= 7) {                        /* matched literal :59-114 */
                uint8_t match_byte = win_get(w, s->rep0 + 1);
                while (symbol < 0x100) {
                    uint32_t match_bit = (match_byte >> 7) & 1;
                    match_byte <<= 1;
                    BIT(&probs[((1 + match_bit) << 8) + symbol], bit);
                    symbol = (symbol << 1) | bit;
                    if (match_bit != bit) break; /* :77-79,97-99; deferred normalise :116-125
                                                    is the same normalise BIT() just did */
                }
            }
            while (symbol < 0x100) { /* :127-166 */
                BIT(&probs[symbol], bit);
                symbol = (symbol << 1) | bit;
            }
            win_put(w, (uint8_t)(symbol - 0x100)); /* :168 */
            s->state = upd_literal(s->state);      /* :171 */
            s->bytes_left--;                       /* :172 */
            continue;
        }

        BIT(&s->is_rep[s->state], bit); /* :195-213,669-683 */
        if (bit == 0) {
            /* simple match, :215-668 */
            uint32_t len_state, pos_slot;
            s->rep3 = s->rep2;
            s->rep2 = s->rep1;
            s->rep1 = s->rep0; /* :216 */
            LEN_DECODE(s->len_choice, s->len_choice2, s->len_low, s->len_mid, s->len_high, length);
            s->state = upd_match(s->state); /* :431 */
            len_state = length > K_LEN_TO_POS_STATES - 1 ? K_LEN_TO_POS_STATES - 1 : length;
            TREE(s->pos_slot[len_state], 6, pos_slot); /* :441-486 */
            pos_slot -= 64;
            if (pos_slot < 4) {
                s->rep0 = pos_slot; /* :488-489 */
            } else {
                uint32_t nbits = (pos_slot >> 1) - 1;
                uint32_t dist = (2 | (pos_slot & 1)) << nbits; /* :491-492 */
                uint32_t sym;
                if (pos_slot < K_END_POS_MODEL) {
                    prob_t *pp = &s->pos_dec[dist - pos_slot]; /* :496 */
                    RTREE(pp, nbits, sym);
                    dist += sym;
                    s->rep0 = dist; /* :544-545 */
                } else {
                    uint32_t res = 0; /* direct bits :549-577 */
                    for (uint32_t n = nbits - K_ALIGN_BITS; n > 0; n--) {
                        uint32_t t;
                        range >>= 1;
                        code -= range;
                        t = 0 - (code >> 31);
                        code += range & t;
                        res = (res << 1) + (t + 1);
                        NORMALIZE();
                    }
                    dist += res << K_ALIGN_BITS;
                    RTREE(s->align, K_ALIGN_BITS, sym); /* :579-625 */
                    dist += sym;
                    s->rep0 = dist; /* :627-628 */
                }
            }
            if (s->rep0 == 0xFFFFFFFFu) { /* end marker :633-645 */
                if (code == 0) {
                    if (s->unpack_size_defined && s->bytes_left > 0) return RUN_ERR_RESULT;
                    break; /* err = io.EOF; falls out, saving Code/Range :1132-1133 */
                }
                return RUN_ERR_RESULT;
            }
            if (s->unpack_size_defined && s->bytes_left == 0) return RUN_ERR_RESULT; /* :647-649 */
            if (s->rep0 >= w->size || !win_check_distance(w, s->rep0))               /* :651-653 */
                return RUN_ERR_RESULT;
            length += K_MATCH_MIN_LEN; /* :656 */
        } else {
            /* rep match, :685-1123 */
            if (s->unpack_size_defined && s->bytes_left == 0) return RUN_ERR_RESULT; /* :686-688 */
            if (win_is_empty(w)) return RUN_ERR_RESULT;                              /* :690-692 */
            BIT(&s->is_rep_g0[s->state], bit);                                       /* :694-772 */
            if (bit == 0) {
                BIT(&s->is_rep0_long[state2], bit); /* :715-756 */
                if (bit == 0) {                     /* short rep :735-739 */
                    s->state = upd_shortrep(s->state);
                    win_put(w, win_get(w, s->rep0 + 1));
                    s->bytes_left--;
                    continue;
                }
            } else {
                uint32_t dist;
                BIT_NN(&s->is_rep_g1[s->state], bit); /* :777-813 */
                if (bit == 0) {
                    dist = s->rep1;
                    s->rep1 = s->rep0;
                    s->rep0 = dist; /* rotated before the normalise, :785-798 */
                    NORMALIZE();
                } else {
                    NORMALIZE();
                    BIT_NN(&s->is_rep_g2[s->state], bit); /* :816-861 */
                    if (bit == 0) {
                        dist = s->rep2;
                        s->rep2 = s->rep1;
                    } else {
                        dist = s->rep3;
                        s->rep3 = s->rep2;
                        s->rep2 = s->rep1;
                    }
                    s->rep1 = s->rep0;
                    s->rep0 = dist;
                    NORMALIZE();
                }
            }
            LEN_DECODE(s->rep_choice, s->rep_choice2, s->rep_low, s->rep_mid, s->rep_high, length);
            s->state = upd_rep(s->state); /* :933,1027,1103 */
            length += K_MATCH_MIN_LEN;
        }

        /* :657-668, 936-947, 1030-1041, 1106-1117.  Note the 32-bit truncation
         * of bytesLeft (SURVEY parity note 8). */
        if (s->unpack_size_defined && (uint32_t)s->bytes_left < length) {
            length = (uint32_t)s->bytes_left;
            win_copy(w, s->rep0 + 1, length);
            s->bytes_left -= length;
            return RUN_ERR_RESULT;
        }
        win_copy(w, s->rep0 + 1, length);
        s->bytes_left -= length;
    }

    d->code = code; /* :1132-1133 */
    d->range = range;
    return rc;
}

static void dec_free(dec_t *d)
{
    free(d->win.buf);
    free(d->st.lit);
}

static void finish(dec_t *d, xlzo_result *res, int status)
{
    res->out_len = d->win.out_len;
    if (d->win.overflow) {
        status = XLZO_ERR_OUT_CAP;
        res->out_len = d->win.out_cap;
    }
    res->in_consumed = d->src.pos;
    res->status = status;
    dec_free(d);
}

static int run_to_status(int r)
{
    switch (r) {
    case RUN_END: return XLZO_OK;
    case RUN_INPUT_EOF: return XLZO_OK_INPUT_EOF;
    case RUN_OUT_CAP: return XLZO_ERR_OUT_CAP;
    default: return XLZO_ERR_RESULT;
    }
}

/* Reader1.initialize (reader1.go:149-159) + Read to EOF */
static int lzma1_body(dec_t *d, uint8_t props, uint64_t unpack_size, xlzo_result *res)
{
    uint8_t lc, pb, lp;
    int r;
    if (!xlzo_decode_prop(props, &lc, &pb, &lp)) {
        finish(d, res, XLZO_ERR_PROPS);
        return 0;
    }
    if (!st_renew(&d->st, lc, pb, lp)) {
        finish(d, res, XLZO_ERR_BAD_ARG);
        return 0;
    }
    st_set_unpack_size(&d->st, unpack_size);
    d->range = 0xFFFFFFFFu; /* range_decoder.go:15-21 */
    d->code = 0;
    r = rc_init(d);
    if (r) {
        finish(d, res, r == 1 ? XLZO_ERR_HEADER_EOF : XLZO_ERR_RC_INIT);
        return 0;
    }
    finish(d, res, run_to_status(lzma_run(d)));
    return 0;
}

int xlzo_lzma1_raw(uint8_t props, uint32_t dict_size, uint64_t unpack_size, const uint8_t *in,
                   size_t in_len, uint8_t *out, size_t out_cap, xlzo_result *res)
{
    dec_t d;
    if (!res || (!in && in_len) || (!out && out_cap)) return XLZO_ERR_BAD_ARG;
    memset(&d, 0, sizeof d);
    memset(res, 0, sizeof *res);
    d.src.p = in;
    d.src.len = in_len;
    d.src.limit = -1;
    if (dict_size < LZMA_DIC_MIN) dict_size = LZMA_DIC_MIN; /* reader1.go:199-201 */
    if (!win_init(&d.win, dict_size, out, out_cap)) return XLZO_ERR_BAD_ARG;
    return lzma1_body(&d, props, unpack_size, res);
}

/* reader1.go:77-101 */
int xlzo_lzma1_alone(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                     xlzo_result *res)
{
    dec_t d;
    uint8_t hdr[13];
    if (!res || (!in && in_len) || (!out && out_cap)) return XLZO_ERR_BAD_ARG;
    memset(&d, 0, sizeof d);
    memset(res, 0, sizeof *res);
    d.src.p = in;
    d.src.len = in_len;
    d.src.limit = -1;
    d.win.out = out;
    d.win.out_cap = out_cap;
    /* props byte first: a bad props byte is reported even if the rest of the
     * header is missing (reader1.go:78-86) */
    if (!src_byte(&d.src, &hdr[0])) {
        finish(&d, res, XLZO_ERR_HEADER_EOF);
        return 0;
    }
    if (hdr[0] >= 9 * 5 * 5) {
        finish(&d, res, XLZO_ERR_PROPS);
        return 0;
    }
    for (int i = 1; i < 13; i++)
        if (!src_byte(&d.src, &hdr[i])) {
            finish(&d, res, XLZO_ERR_HEADER_EOF);
            return 0;
        }
    if (!win_init(&d.win, xlzo_decode_dict_size(hdr + 1), out, out_cap)) return XLZO_ERR_BAD_ARG;
    return lzma1_body(&d, hdr[0], xlzo_decode_unpack_size(hdr + 5), res);
}

/* ------------------------------------------------------------------ LZMA2 ---- */
/* types.go:50-60 */
enum { CH_END = 0, CH_UNC_RESET, CH_UNC, CH_LZMA, CH_LZMA_STATE, CH_LZMA_PROP, CH_LZMA_PROP_DICT };

/* reader2.go:175-199: 0x03..0x7F fall through to end-of-stream (parity note 7) */
static int chunk_type(uint8_t c)
{
    if (c == 0) return CH_END;
    if (c == 1) return CH_UNC_RESET;
    if (c == 2) return CH_UNC;
    switch (c >> 5) {
    case 4: return CH_LZMA;
    case 5: return CH_LZMA_STATE;
    case 6: return CH_LZMA_PROP;
    case 7: return CH_LZMA_PROP_DICT;
    }
    return CH_END;
}

/* reader2.go:201-214 */
static int chunk_hdr_len(int t)
{
    switch (t) {
    case CH_UNC_RESET:
    case CH_UNC: return 3;
    case CH_LZMA:
    case CH_LZMA_STATE: return 5;
    case CH_LZMA_PROP:
    case CH_LZMA_PROP_DICT: return 6;
    }
    return 1;
}

int xlzo_lzma2_raw(uint32_t dict_size, uint32_t flags, const uint8_t *in, size_t in_len,
                   uint8_t *out, size_t out_cap, xlzo_result *res)
{
    dec_t d;
    uint8_t header[6] = {0, 0, 0, 0, 0, 0}; /* reader2.go:37: persistent across chunks */
    int have_reader = 0, status = XLZO_OK;
    if (!res || (!in && in_len) || (!out && out_cap)) return XLZO_ERR_BAD_ARG;
    memset(&d, 0, sizeof d);
    memset(res, 0, sizeof *res);
    d.src.p = in;
    d.src.len = in_len;
    d.src.limit = -1;
    if (dict_size < LZMA_DIC_MIN) dict_size = 8u * 1024 * 1024; /* reader2.go:88-91 */
    if (!win_init(&d.win, dict_size, out, out_cap)) return XLZO_ERR_BAD_ARG;

    for (;;) {
        int type, hl;
        uint32_t unc, comp;
        uint8_t lc, pb, lp;
        int r;

        if (d.win.overflow) break;
        /* startChunk, reader2.go:100-173 */
        d.src.limit = -1;
        if (!src_byte(&d.src, &header[0])) {
            status = XLZO_ERR_UNEXPECTED_EOF; /* :104-110 */
            break;
        }
        type = chunk_type(header[0]);
        if (type == CH_END) break; /* :117-119 -> Read returns io.EOF :221-222 */
        hl = chunk_hdr_len(type);
        {
            int short_read = 0;
            for (int i = 1; i < hl; i++)
                if (!src_byte(&d.src, &header[i])) {
                    short_read = 1;
                    break;
                }
            if (short_read) {
                status = XLZO_ERR_UNEXPECTED_EOF; /* :121-128 */
                break;
            }
        }
        unc = ((uint32_t)header[1] << 8) | header[2];                      /* :130 */
        if (type == CH_UNC_RESET || type == CH_LZMA_PROP_DICT) win_reset(&d.win); /* :132-134 */
        if (type == CH_UNC_RESET || type == CH_UNC) {
            /* stored chunk: uncompressedRead, reader2.go:252-294 + window.go:142-155.
             * A short source simply delivers what is there; the next startChunk
             * then reports io.ErrUnexpectedEOF. */
            unc++;
            while (unc > 0 && d.src.pos < d.src.len) {
                win_put(&d.win, d.src.p[d.src.pos++]);
                unc--;
            }
            continue;
        }
        unc |= (uint32_t)(header[0] & 0x1F) << 16; /* :141 */
        unc++;
        comp = (((uint32_t)header[3] << 8) | header[4]) + 1; /* :143-144 */
        if (flags & XLZO_FLAG_REF_U16_COMPSIZE) comp &= 0xFFFFu; /* uint16 wrap, :21 */
        d.src.limit = (int64_t)comp;                             /* limitByteReader :147,167 */

        if (!have_reader) {
            /* NewReader1ForReader2, reader1.go:63-75: props come from header[5]
             * whatever the chunk type was (reader2.go:146-153) */
            if (!xlzo_decode_prop(header[5], &lc, &pb, &lp)) {
                status = XLZO_ERR_PROPS;
                break;
            }
            if (!st_renew(&d.st, lc, pb, lp)) {
                status = XLZO_ERR_BAD_ARG;
                break;
            }
            have_reader = 1;
            st_set_unpack_size(&d.st, unc);
            d.range = 0xFFFFFFFFu;
            d.code = 0;
            r = rc_init(&d);
            if (r) { /* "rangeDec.Init: %w" -- a wrapped EOF is an error for io.Copy */
                status = r == 1 ? XLZO_ERR_HEADER_EOF : XLZO_ERR_RC_INIT;
                break;
            }
        } else {
            if (type == CH_LZMA_STATE) {
                st_reset(&d.st); /* :156-157 */
            } else if (type == CH_LZMA_PROP || type == CH_LZMA_PROP_DICT) {
                if (!xlzo_decode_prop(header[5], &lc, &pb, &lp)) { /* :159-162 */
                    status = XLZO_ERR_PROPS;
                    break;
                }
                if (!st_renew(&d.st, lc, pb, lp)) { /* :164 */
                    status = XLZO_ERR_BAD_ARG;
                    break;
                }
            }
            /* Reopen, reader1.go:166-176 + range_decoder.go:48-55 */
            st_set_unpack_size(&d.st, unc);
            d.range = 0xFFFFFFFFu;
            d.code = 0;
            r = rc_init(&d);
            if (r == 1) { /* raw io.EOF travels up through Reader2.Read as a clean EOF */
                status = XLZO_OK_INPUT_EOF;
                break;
            }
            if (r == 2) {
                status = XLZO_ERR_RC_INIT;
                break;
            }
        }
        r = lzma_run(&d);
        if (r == RUN_ERR_RESULT) {
            status = XLZO_ERR_RESULT;
            break;
        }
        if (r == RUN_OUT_CAP) break;
        /* RUN_END / RUN_INPUT_EOF: io.EOF from the chunk -> next startChunk (:234-241).
         * Unread bytes of the chunk are NOT skipped by the reference. */
    }
    d.src.limit = -1;
    finish(&d, res, status);
    return 0;
}

/* --------------------------------------------------- multi-thread batch ---- */
typedef struct {
    const xlzo_job *jobs;
    xlzo_result *res;
    size_t n;
    size_t next;
    pthread_mutex_t mu;
} pool_t;

static void *worker(void *arg)
{
    pool_t *p = (pool_t *)arg;
    for (;;) {
        size_t i;
        pthread_mutex_lock(&p->mu);
        i = p->next++;
        pthread_mutex_unlock(&p->mu);
        if (i >= p->n) break;
        const xlzo_job *j = &p->jobs[i];
        if (j->fmt == 0)
            xlzo_lzma1_alone(j->in, j->in_len, j->out, j->out_cap, &p->res[i]);
        else
            xlzo_lzma2_raw(j->dict_size, 0, j->in, j->in_len, j->out, j->out_cap, &p->res[i]);
    }
    return NULL;
}

int xlzo_decode_batch_mt(const xlzo_job *jobs, size_t n, int nthreads, xlzo_result *res)
{
    pool_t p;
    pthread_t th[256];
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    p.jobs = jobs;
    p.res = res;
    p.n = n;
    p.next = 0;
    pthread_mutex_init(&p.mu, NULL);
    for (int t = 0; t < nthreads; t++) pthread_create(&th[t], NULL, worker, &p);
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
    pthread_mutex_destroy(&p.mu);
    return 0;
}
