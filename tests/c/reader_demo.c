/* A plain C99 caller of the C ABI (no HIP headers, no C++): the closest stand-in this image has
 * for the cgo binding of INTEGRATION.md.  It does what the reference's own test does with
 * testassets/a.lzma (reader1_test.go:69-80): NewReader1(file) then io.Copy with a small buffer.
 *
 *   reader_demo <file.lzma> [buffer bytes]     prints: status, bytes, fnv1a-64 of the output
 * exit 0: decoded to io.EOF; 3: no usable GPU (XLZ_ERR_DEVICE); 1: any other error. */
#include <stdio.h>
#include <stdlib.h>

#include "xlz.h"

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    size_t bufsz = argc > 2 ? (size_t)atol(argv[2]) : 32768; /* io.Copy's buffer */
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    uint8_t *in = (uint8_t *)malloc(n > 0 ? (size_t)n : 1);
    if (fread(in, 1, (size_t)n, f) != (size_t)n) return 2;
    fclose(f);

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
    free(in); /* the reader owns a copy */
    uint8_t *buf = (uint8_t *)malloc(bufsz);
    uint64_t total = 0, h = 1469598103934665603ull;
    for (;;) { /* io.Copy */
        long got = xlz_reader_read(r, buf, bufsz, &err);
        for (long i = 0; i < got; i++) h = (h ^ buf[i]) * 1099511628211ull;
        total += (uint64_t)got;
        if (err != XLZ_OK) break;
    }
    printf("%s %llu %016llx\n", err == XLZ_EOF ? "EOF" : xlz_strerror(err), (unsigned long long)total,
           (unsigned long long)h);
    int rc = xlz_reader_close(r) == XLZ_OK && xlz_reader_close(r) == XLZ_ERR_CLOSED ? 0 : 1; /* readcloser.go:16-28 */
    xlz_reader_free(r);
    xlz_ctx_destroy(ctx);
    free(buf);
    return err == XLZ_EOF ? rc : 1;
}
