/* A plain C99 caller of the C ABI (no HIP headers, no C++): the closest stand-in this image has
 * for the cgo binding of INTEGRATION.md.  It does what the reference's own test does with
 * testassets/a.lzma (reader1_test.go:69-80): NewReader1(file) then io.Copy with a small buffer.
 *
 * The compressed side is pulled from the file piece by piece as the decoder asks for it
 * (xlz_reader_expect_more / XLZ_NEED_INPUT / xlz_reader_feed), as the Go shim does with its io.Reader.
 *
 *   reader_demo <file.lzma> [buffer bytes] [piece bytes]     prints: status, bytes, fnv1a-64 of the output
 * exit 0: decoded to io.EOF; 3: no usable GPU (XLZ_ERR_DEVICE); 1: any other error. */
#include <stdio.h>
#include <stdlib.h>

#include "xlz.h"

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    size_t bufsz = argc > 2 ? (size_t)atol(argv[2]) : 32768; /* io.Copy's buffer */
    size_t piece = argc > 3 ? (size_t)atol(argv[3]) : 262144; /* compressed bytes pulled at a time */
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    uint8_t *in = (uint8_t *)malloc(piece);
    size_t n = fread(in, 1, piece, f); /* the first piece: header + range-coder bytes are in it */
    int more = n == piece;

    xlz_ctx *ctx = NULL;
    int st = xlz_ctx_create(0, &ctx);
    if (st != XLZ_OK) {
        printf("ctx: %s\n", xlz_strerror(st));
        return st == XLZ_ERR_DEVICE ? 3 : 1;
    }
    int err = 0;
    xlz_reader *r = xlz_new_reader1(ctx, in, (size_t)n, &err); /* NewReader1 (reader1.go:18) */
    if (!r) {
        printf("constructor: %s\n", xlz_strerror(err));
        xlz_ctx_destroy(ctx);
        return 1;
    }
    if (more && xlz_reader_expect_more(r) != XLZ_OK) return 1;
    uint8_t *buf = (uint8_t *)malloc(bufsz);
    uint64_t total = 0, h = 1469598103934665603ull;
    for (;;) { /* io.Copy */
        long got = xlz_reader_read(r, buf, bufsz, &err);
        for (long i = 0; i < got; i++) h = (h ^ buf[i]) * 1099511628211ull;
        total += (uint64_t)got;
        if (err == XLZ_NEED_INPUT) { /* the next piece of the source, or its end */
            n = fread(in, 1, piece, f);
            if (n && xlz_reader_feed(r, in, n) != XLZ_OK) return 1;
            if (n < piece && xlz_reader_feed_eof(r) != XLZ_OK) return 1;
            continue;
        }
        if (err != XLZ_OK) break;
    }
    free(in);
    fclose(f);
    printf("%s %llu %016llx\n", err == XLZ_EOF ? "EOF" : xlz_strerror(err), (unsigned long long)total,
           (unsigned long long)h);
    int rc = xlz_reader_close(r) == XLZ_OK && xlz_reader_close(r) == XLZ_ERR_CLOSED ? 0 : 1; /* readcloser.go:16-28 */
    xlz_reader_free(r);
    xlz_ctx_destroy(ctx);
    free(buf);
    return err == XLZ_EOF ? rc : 1;
}
