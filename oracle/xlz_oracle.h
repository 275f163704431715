/*
 * xlz_oracle.h -- CPU oracle for the batched LZMA/LZMA2 decode path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under lzma_amd/ may include, link or
 * call this.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / reported CPU baseline.
 *
 * It is a plain-C restatement of the algorithm of the pure-Go reference
 * kulaginds/lzma (decompress.go, window.go, state.go, range_decoder.go,
 * reader1.go, reader2.go, bytereader.go).  Each function cites the reference
 * file:line it follows.  Pinned against the reference's own test vectors
 * (reader1_test.go:15-107, reader2_test.go:12-29): see tests/test_oracle_golden.py.
 */
#ifndef XLZ_ORACLE_H
#define XLZ_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Status codes.  Same numbering as include/xlz.h (tests assert they agree). */
enum {
    XLZO_OK = 0,                 /* reader reached io.EOF cleanly (reader1.go:239-243)            */
    XLZO_OK_INPUT_EOF = 1,       /* input ran out; reference treats ReadByte's io.EOF as a clean
                                    end of stream (decompress.go:35-38 + reader1.go:246-249)      */
    XLZO_ERR_RESULT = -1,        /* ErrResultError out of decompress (errors.go:8)                */
    XLZO_ERR_PROPS = -2,         /* ErrIncorrectProperties (reader1.go:211-213)                   */
    XLZO_ERR_HEADER_EOF = -3,    /* constructor failed: input ended inside header / rc init
                                    (reader1.go:78-98,153-156, range_decoder.go:28-43)            */
    XLZO_ERR_RC_INIT = -4,       /* first range-coder byte != 0 (range_decoder.go:32-34)          */
    XLZO_ERR_UNEXPECTED_EOF = -5,/* io.ErrUnexpectedEOF from LZMA2 startChunk (reader2.go:104-127)*/
    XLZO_ERR_OUT_CAP = -6,       /* (no reference analogue) caller's output buffer too small      */
    XLZO_ERR_BAD_ARG = -7
};

typedef struct xlzo_result {
    uint64_t out_len;      /* bytes the decoder produced (put into the window), error or not */
    uint64_t in_consumed;  /* input bytes the reader pulled from the source                   */
    int32_t status;
    int32_t reserved;
} xlzo_result;

/* flags */
#define XLZO_FLAG_REF_U16_COMPSIZE 1u /* reproduce reader2.go:21,143-144: uint16 chunkCompressedSize
                                         wraps 65536 -> 0 (off by default: SURVEY parity note 7)  */

/* NewReader1 + io.Copy: 13-byte .lzma header in-band (reader1.go:18-24,77-101). */
int xlzo_lzma1_alone(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                     xlzo_result *res);

/* NewLZMADecompressorForSevenZip: props byte, dict size and unpack size
 * out-of-band (reader1.go:32-61). */
int xlzo_lzma1_raw(uint8_t props, uint32_t dict_size, uint64_t unpack_size, const uint8_t *in,
                   size_t in_len, uint8_t *out, size_t out_cap, xlzo_result *res);

/* NewReader2(in, dictSize) + io.Copy (reader2.go:26-41,216-250). */
int xlzo_lzma2_raw(uint32_t dict_size, uint32_t flags, const uint8_t *in, size_t in_len,
                   uint8_t *out, size_t out_cap, xlzo_result *res);

/* Helpers mirroring exported reference helpers. */
int xlzo_decode_prop(uint8_t d, uint8_t *lc, uint8_t *pb, uint8_t *lp); /* reader1.go:210-221 */
uint32_t xlzo_decode_dict_size(const uint8_t p[4]);                     /* reader1.go:193-208 */
uint32_t xlzo_decode_dict_size2(uint8_t b);                             /* reader2.go:296-298 */
uint64_t xlzo_decode_unpack_size(const uint8_t h[8]);                   /* reader1.go:178-191 */

/* Batch driver for the CPU baseline: decode n streams with nthreads pthreads
 * (one stream per thread at a time).  fmt: 0 = alone, 2 = lzma2 raw.       */
typedef struct xlzo_job {
    const uint8_t *in;
    size_t in_len;
    uint8_t *out;
    size_t out_cap;
    uint32_t fmt;
    uint32_t dict_size; /* lzma2 only */
} xlzo_job;
int xlzo_decode_batch_mt(const xlzo_job *jobs, size_t n, int nthreads, xlzo_result *res);

#ifdef __cplusplus
}
#endif
#endif
