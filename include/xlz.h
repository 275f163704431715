/*
 * xlz.h -- C ABI of the MI355X-native batched LZMA / LZMA2 decoder (libxlz.so).
 *
 * This is the drop-in boundary for the hot path of kulaginds/lzma:
 * (*Reader1).decompress (decompress.go:8-1136) and everything it drives
 * (window.go, state.go, range_decoder.go), plus the LZMA2 chunk framing of
 * reader2.go:100-298.  The reference is pure Go with no FFI; the entry points
 * below are what a cgo binding for that path binds (see INTEGRATION.md for the
 * Go side).  Plain pointers and sizes only; no C++ or torch types.
 *
 * All compute runs in hand-written HIP kernels on gfx950.  There is NO CPU
 * decode path in this library: without a usable HIP device every decode entry
 * point fails with XLZ_ERR_DEVICE.
 *
 * The library reads ONE environment variable, a debugging aid: XLZ_DEBUG (any
 * value) prints failed HIP calls and the phase times of xlz_decode_batch to
 * stderr.  Nothing tunes the decode.  (An A/B build made with -DXLZ_DEV_KNOBS also reads
 * XLZ_STORED_UNIT_KIB, the least size of a unit of stored LZMA2 chunks; the shipped build
 * does not contain that code.)
 *
 * file:line citations are into the reference repository.
 */
#ifndef XLZ_H
#define XLZ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XLZ_VERSION_MAJOR 0
#define XLZ_VERSION_MINOR 1

/* ---- per-stream status -------------------------------------------------- */
/* >= 0: the reference's reader would have ended with io.EOF (no error).
 *  < 0: the reference's constructor or Read would have returned an error.   */
enum {
    XLZ_OK = 0,                  /* clean end: size reached with Code==0, or end marker
                                    (decompress.go:14-20,633-641)                               */
    XLZ_OK_INPUT_EOF = 1,        /* input exhausted; the reference turns ReadByte's io.EOF into a
                                    normal end of stream (decompress.go:35-38, reader1.go:246)  */
    XLZ_ERR_RESULT = -1,         /* ErrResultError (errors.go:8)                                 */
    XLZ_ERR_PROPS = -2,          /* ErrIncorrectProperties (reader1.go:211-213)                  */
    XLZ_ERR_HEADER_EOF = -3,     /* constructor error: input ended inside the 13-byte header or
                                    the 5 range-coder init bytes (reader1.go:78-98,153-156)     */
    XLZ_ERR_RC_INIT = -4,        /* first range-coder byte != 0 (range_decoder.go:32-34)         */
    XLZ_ERR_UNEXPECTED_EOF = -5, /* io.ErrUnexpectedEOF from LZMA2 framing (reader2.go:104-127)  */
    XLZ_ERR_OUT_CAP = -6,        /* new: out_cap smaller than the decoded size (the reference has no
                                    output limit); out_len==out_cap, in_consumed = the input read
                                    when the packet / stored chunk that did not fit was complete   */
    XLZ_ERR_BAD_ARG = -7,        /* new: NULL pointer / unknown format                           */
    XLZ_ERR_DEVICE = -8,         /* new: HIP runtime failure or no gfx950 device                 */
    XLZ_ERR_UNSUPPORTED = -9,    /* new: stream outside what the GPU path implements (DESIGN.md
                                    section 5.2: a dictionary > 2 GiB on a stream >= 4 GiB; a unit
                                    >= 4 GiB of a device-resident xlz_batch; container features) */
    XLZ_ERR_CLOSED = -10,        /* errAlreadyClosed (readcloser.go:14)                          */
    XLZ_ERR_NEED_ONE_READER = -11, /* errNeedOneReader (reader1.go:26)                           */
    XLZ_ERR_INSUFFICIENT_PROPS = -12 /* errInsufficientProperties (reader2.go:43)                */
};

/* ---- stream formats ------------------------------------------------------ */
enum {
    XLZ_FMT_LZMA_ALONE = 0, /* NewReader1: 13-byte header in-band (reader1.go:18-24,77-101)       */
    XLZ_FMT_LZMA_RAW = 1,   /* NewLZMADecompressorForSevenZip: props byte, dict size and unpack
                               size out of band (reader1.go:32-61)                               */
    XLZ_FMT_LZMA2_RAW = 2   /* NewReader2(in, dictSize) (reader2.go:26-41)                        */
};

typedef struct xlz_stream_desc {
    const uint8_t *in;    /* compressed bytes (host memory)                                       */
    size_t in_len;
    uint8_t *out;         /* host destination; may be NULL for device-resident batches            */
    size_t out_cap;       /* capacity reserved for this stream's output                           */
    uint32_t format;      /* XLZ_FMT_*                                                            */
    uint32_t dict_size;   /* LZMA_RAW: value of DecodeDictSize(props[1:5]); LZMA2_RAW: dictSize
                             argument of NewReader2 (values < 4096 mean 8 MiB, reader2.go:88-91)  */
    uint64_t unpack_size; /* LZMA_RAW only; all-ones = unknown (state.go:135-151)                 */
    uint8_t props;        /* LZMA_RAW only: the lc/lp/pb byte                                     */
    uint8_t flags;        /* XLZ_STREAM_F_* (0 for an ordinary stream); any other bit, or the slice flag
                             on a format that has no slices: XLZ_ERR_BAD_ARG for the call           */
    uint8_t reserved[6];  /* must be zero (memset the descriptor): XLZ_ERR_BAD_ARG otherwise        */
} xlz_stream_desc;

/* LZMA2_RAW: `in` is a SLICE of a longer stream that does not begin at the stream's start -- it begins
 * where a unit of xlz_lzma2_units begins (callers that deal the units of one stream to several GPUs or
 * processes: xlz_decode_batch_multi does it by itself).  Bytes in front of the slice's first
 * dictionary epoch belong to units that are not in this call: a (malformed) stream that reads them,
 * or whose units do not decode to what their headers announce, ends in XLZ_ERR_UNSUPPORTED instead
 * of being settled inside the slice -- decode the whole stream then.  A slice that ends before its
 * stream does (no end byte) ends in XLZ_ERR_UNEXPECTED_EOF with all of its input consumed: that is
 * its clean outcome.                                                                              */
#define XLZ_STREAM_F_LZMA2_SLICE 1u

typedef struct xlz_result {
    uint64_t out_len;     /* bytes produced by the decoder (even when status < 0)                 */
    uint64_t in_consumed; /* input bytes pulled from the source, header included                  */
    int32_t status;       /* XLZ_OK ... */
    int32_t reserved;
} xlz_result;

/* ---- library ------------------------------------------------------------- */
const char *xlz_version(void);
const char *xlz_build_id(void);       /* hash of the sources this binary was compiled from (lzma_amd/build.py);
                                         measurements quote it so that a number names the kernel that ran   */
const char *xlz_kernel_id(void);      /* the same over the device code's sources only: what rocprof profiles are tied to   */
const char *xlz_strerror(int status); /* text of the matching reference error (errors.go:5-12)    */
int xlz_device_count(void);           /* number of HIP devices, 0 if none                         */

/* helpers with the reference's exported names */
int xlz_decode_prop(uint8_t d, uint8_t *lc, uint8_t *pb, uint8_t *lp); /* DecodeProp reader1.go:210 */
uint32_t xlz_decode_dict_size(const uint8_t properties[4]);            /* DecodeDictSize   :193    */
uint32_t xlz_decode_dict_size2(uint8_t encoded);                       /* DecodeDictSize2 reader2.go:296 */
uint64_t xlz_decode_unpack_size(const uint8_t header[8]);              /* DecodeUnpackSize :178    */

/* ---- context: one per (host thread, GPU) ---------------------------------- */
typedef struct xlz_ctx xlz_ctx;
int xlz_ctx_create(int device, xlz_ctx **ctx); /* XLZ_OK or XLZ_ERR_DEVICE                         */
void xlz_ctx_destroy(xlz_ctx *ctx);
int xlz_ctx_device(const xlz_ctx *ctx);

/* HIP events on the context's stream, for callers that time a region of enqueued work
 * (bench.py): slot 0..63.  elapsed_ms waits for event `b`.                           */
int xlz_ctx_event_record(xlz_ctx *ctx, int slot);
int xlz_ctx_event_elapsed_ms(xlz_ctx *ctx, int slot_a, int slot_b, float *ms);

/* Coalescing of pull-style readers.  The reference's readers are
 * independent single-goroutine objects; many goroutines each doing io.Copy(dst, NewReader1(src))
 * would each occupy one wave of the GPU with a launch of its own.  After this call, every refill
 * of a reader of `ctx` is handed to a background thread that waits up to `window_us` for refills
 * of other readers (at most `max_streams`) and runs them as ONE launch; xlz_reader_read blocks
 * until its refill is done.                                                                      */
int xlz_ctx_enable_batching(xlz_ctx *ctx, uint32_t window_us, uint32_t max_streams);
int xlz_ctx_batching_stats(xlz_ctx *ctx, uint64_t *batches, uint64_t *streams);

/* ---- one-shot batch decode: host buffers in, host buffers out -------------- */
/* Replaces a loop of `r, _ := NewReader1(src); io.Copy(dst, r)` over n independent
 * streams (reader1_test.go:76-80).  One bad stream never fails the batch: the return
 * value is XLZ_OK unless the call itself could not run; per-stream outcomes are in
 * results[i].  Thread-safe across contexts; calls on one context are serialised.
 * A call of several wave rounds (>= 8192 streams and >= 1 GiB of output) is cut into up
 * to ten sub-batches whose upload, decode and download overlap; a call of ONE wave round (and
 * at least 256 MiB of output) runs as up to eight launches whose downloads overlap the decode
 * (xlz_call_stats.slices, xlz_ctx_set_slicing).  Bytes of `out` beyond out_len are unspecified.
 * BREAK-EVEN: one wave decodes one unit (a stream; an LZMA2 dictionary-reset unit) at
 * 4-6 MB/s and the chip holds 4096 of them (up to 6144 on calls of many rounds), so a call with few units is slower than the
 * host's own cores: measured on 1 MiB text streams, 64 units 0.34 GiB/s (16 host cores:
 * 1.27), 256 units 1.37 (1.28), 1024 units 5.2, 4096 units 17.2.  Below about 250 units
 * per 16 host cores decode on the CPU, or gather more streams first;
 * xlz_batch_advice answers that question for a given call before anything is uploaded.  */
int xlz_decode_batch(xlz_ctx *ctx, const xlz_stream_desc *streams, size_t n, xlz_result *results);
/* Host only (no device needed; only out_cap of every stream is looked at): how xlz_decode_batch would cut this call into
 * pieces.  cuts[k] = index of the first stream of piece k, one entry more than there are pieces (the last is n); *n_cuts =
 * entries (XLZ_ERR_OUT_CAP if max_cuts is smaller; cuts may be NULL with max_cuts = 0 to ask for the count); *mode (may be
 * NULL): 0 one piece (a call of one wave round overlaps its copies with its own decode: slices), 1 a pipeline of pieces
 * whose launches overlap on two streams (many short streams), 2 pieces of exactly one wave round of 4096 streams, one
 * behind the other, each sliced (few rounds of long streams: 256 KiB and more on average).                          */
int xlz_decode_batch_plan(const xlz_stream_desc *streams, size_t n, size_t *cuts, size_t max_cuts, size_t *n_cuts, int *mode);

/* BEFORE uploading anything: how much of a GPU would this call fill, and is the host the faster
 * decoder for it?  Host only (reads stream and LZMA2 chunk headers; no device needed; `ctx` may be
 * NULL: an MI355X's 4096 wave slots are assumed).  Two estimated times are compared, in compressed
 * bytes of serial work (decode time tracks them):
 *   GPU: one wave decodes one unit -- an LZMA1 stream, or one unit of an LZMA2 stream's plan
 *        (xlz_lzma2_units) -- and `wave_slots` units run at once:
 *        gpu_cost = max(largest unit, all bytes / wave_slots);
 *   CPU: a host core decodes about 16 times as fast as a wave (65-80 MB/s of output against 4-6), but
 *        ONE stream is ONE thread whatever its format -- the reference's Reader2 is one goroutine
 *        (reader2.go:216-250), the units of an LZMA2 stream are parallel work for the GPU only:
 *        cpu_cost = max(largest stream, all bytes / host_threads) / 16.
 * prefer_cpu = cpu_cost < gpu_cost.  For equal LZMA1 streams that is "fewer than 16 units per host
 * thread" (break_even_units; bench.py: stream_count_sweep, 256 units for 16 threads); one LZMA2 stream
 * of 100 units is the GPU's job on any host, 64 big LZMA1 streams are the host's.  `host_threads` = 0:
 * the hardware concurrency of this machine.  The library has no CPU decoder: a caller that is told
 * prefer_cpu = 1 decodes with the reference's own readers (reader1.go:18-24, reader2.go:26-41; the Go
 * shim does that by itself) or gathers more streams. */
typedef struct xlz_advice {
    uint64_t units;            /* independent work units the call would launch                    */
    uint64_t in_bytes;         /* compressed bytes of all streams                                  */
    uint32_t wave_slots;       /* units a GPU decodes at once (resident single-wave workgroups)    */
    uint32_t break_even_units; /* 16 x host_threads: the break-even for EQUAL LZMA1 streams         */
    double   fill;             /* min(1, units / wave_slots): the share of the chip the call uses  */
    int32_t  prefer_cpu;       /* 1: cpu_cost < gpu_cost (or nothing to decode)                    */
    uint32_t reserved;
    double   cpu_cost;         /* estimated serial work per host thread, in wave-equivalent bytes  */
    double   gpu_cost;         /* estimated serial work per wave slot, in compressed bytes         */
} xlz_advice;
int xlz_batch_advice(const xlz_ctx *ctx, const xlz_stream_desc *streams, size_t n, uint32_t host_threads,
                     xlz_advice *out);

/* What the most recent successful xlz_decode_batch on `ctx` spent where, and how well it filled the GPU.
 * ONE wave decodes one stream (LZMA1 has no parallelism inside a stream: decompress.go:13 is a serial
 * chain), at about 5 MB/s of output; the chip holds `wave_slots` (4096) of them.  A call with few
 * streams leaves most of the chip idle -- slot_occupancy says how much of it worked -- and below the
 * break-even stated in INTEGRATION.md the host's own cores are faster.                              */
typedef struct xlz_call_stats {
    double upload_ms;      /* parse headers, pack the inputs into pinned memory, host -> device     */
    double decode_ms;      /* decode launch(es) until the per-stream results are on the host        */
    double download_ms;    /* device -> host and scatter into the callers' buffers                  */
    double total_ms;
    double kernel_span_ms; /* first wave's start to last wave's end over all launches of the call (device clock) */
    double slot_occupancy; /* busy wave time / (wave_slots x kernel span): 1.0 = every slot busy throughout */
    uint64_t streams;      /* n of the call                                                          */
    uint64_t units;        /* work units of the main launch (streams; LZMA2: dictionary-reset units and
                              256 KiB pieces of runs of stored chunks)                               */
    uint32_t wave_slots;   /* resident single-wave workgroups of the launch                          */
    uint32_t sub_batches;  /* > 1: the call ran as a pipeline (upload k+1 / decode k / download k-1); then
                              upload_ms = until the first sub-batch was on the device, decode_ms = first
                              launch to last results, download_ms = what was left after that            */
    uint32_t slices;       /* > 1: a call of ONE wave round -- or every one-round piece of a call of few rounds
                              of long streams (sub_batches > 1) -- ran as a sequence of that many launches, each
                              advancing every unit by a share of its output, the download of share k-1 under
                              the decode of share k (the reference's Read pump in batch form,
                              reader1.go:223-254); for a call of one piece decode_ms = the launches (HIP
                              events), download_ms = what was left after them; slot_occupancy = the mean over
                              the launches                                                                */
    uint32_t reserved;
} xlz_call_stats;
int xlz_ctx_last_call_stats(xlz_ctx *ctx, xlz_call_stats *out);
/* xlz_decode_batch keeps the device and pinned memory of its (sub-)batches in the context between calls (a
 * call of the same shape finds its blocks again; what two calls in a row did not use is released by
 * itself).  xlz_ctx_trim releases all of it now; *released (may be NULL) = the bytes given back.      */
int xlz_ctx_trim(xlz_ctx *ctx, uint64_t *released);
/* Tuning of xlz_decode_batch's sliced form (xlz_call_stats.slices): a call of one wave round with at least
 * min_call_bytes of output room runs as one launch for every slice_bytes of it, at most max_slices (<= 63; from
 * four on the last share is cut in two -- its download is the one nothing overlaps --: one launch more).
 * 0 = the default of that argument (256 MiB, 128 MiB, 8); max_slices = 1 turns slicing off.  The decoded
 * bytes, statuses and consumed input do not depend on it.  The reference's counterpart is the size of the
 * buffer its caller hands to Read (reader1.go:223-254: decompress(need) runs until `need` bytes are pending). */
int xlz_ctx_set_slicing(xlz_ctx *ctx, uint64_t min_call_bytes, uint64_t slice_bytes, uint32_t max_slices);

/* ---- device-resident batch: upload once, decode many times ---------------- */
typedef struct xlz_batch xlz_batch;
/* Parses headers / LZMA2 framing on the host, uploads the compressed bytes and
 * allocates the output arena in HBM.  streams[i].out may be NULL.                 */
int xlz_batch_create(xlz_ctx *ctx, const xlz_stream_desc *streams, size_t n, xlz_batch **batch);
int xlz_batch_run(xlz_batch *batch);  /* enqueue one full decode pass on the context's stream     */
int xlz_batch_sync(xlz_batch *batch); /* wait for everything enqueued so far                      */
int xlz_batch_results(xlz_batch *batch, xlz_result *results); /* sync + fetch per-stream results  */
int xlz_batch_download(xlz_batch *batch, size_t i, uint8_t *dst, size_t cap); /* copy out stream i */
int xlz_batch_device_output(xlz_batch *batch, size_t i, void **dptr, size_t *cap);
/* HIP-event time of the decode kernel(s) of the most recent completed run, in ms */
int xlz_batch_last_kernel_ms(xlz_batch *batch, float *ms);
/* algorithmic bytes (compressed in + decoded out) the last run moved, and units  */
int xlz_batch_stats(xlz_batch *batch, uint64_t *in_bytes, uint64_t *out_bytes, uint64_t *units);
/* When each unit of the last run started and ended on its wave, in ticks of the device's 100 MHz
 * clock since the first unit started, and its compressed size (the work-queue key).  Arrays of
 * `cap` entries (any may be NULL); *n_units = units of the batch.  For slot-occupancy / tail
 * analysis of the persistent grid (bench.py: roofline.issue.slot_occupancy).                */
int xlz_batch_unit_trace(xlz_batch *batch, uint32_t *t_start, uint32_t *t_end, uint32_t *in_len,
                         size_t cap, size_t *n_units);
/* shape of the decode launch: resident single-wave workgroups (= wave slots) and LDS bytes each */
int xlz_batch_launch_info(xlz_batch *batch, uint32_t *workgroups, uint32_t *lds_bytes);
/* which kernel the batch's main launch is (profiles are matched by it): "xlz::xlz_decode_kernel" -- the model laid out
 * with room for 16 posStates (pb <= 4) --, "xlz::xlz_decode_kernel_pb2" -- room for 4: every unit's pb <= 2, the
 * default of liblzma and 7-Zip; five LDS granules instead of six for lc+lp = 3, 24 workgroups per CU instead of 21 --
 * or "xlz::xlz_decode_kernel_hbm_model" (only models beyond LDS, lc+lp > 8)                                          */
const char *xlz_batch_kernel_name(xlz_batch *batch);
void xlz_batch_destroy(xlz_batch *batch);

/* ---- pull-style readers mirroring the reference's Go surface --------------- */
/* Constructors take the compressed stream as a buffer (a Go shim slurps its io.Reader
 * first) and copy it.  Constructor-time errors are the ones the reference's
 * constructors return; decode errors surface from xlz_reader_read once all bytes
 * produced before the error have been delivered.
 *
 * A reader is a resumable decode on the device (reader1.go:223-254 / decompress.go:13:
 * decompress(need) returns once `need` bytes are pending): every refill continues the
 * saved decoder state for about 1 MiB of output and stops.  Memory is bounded like the
 * reference's: one refill chunk on the host, 2 x dictSize + one chunk of window on the
 * device (less for streams of known size), a 4 MiB input window on the device.  The
 * stream is decoded exactly once, whatever its size or compression ratio.  An LZMA2
 * stream whose headers announce eight or more dictionary-reset units is served in runs
 * of whole units (<= 64 MiB of output each) decoded unit-parallel by the batch engine.
 * Sessions cover the reference's whole parameter range: a model with lc+lp > 8 (it does not fit
 * a CU's LDS) runs in the HBM-model launch, an LZMA2 chunk that renews the model with larger
 * properties makes the session grow its state block, and copies that read behind an LZMA2
 * dictionary reset are served from an image of the reference's uncleared window buffer
 * (window.go:135-140) that the device keeps per reader (dictSize bytes, like the window itself). */
typedef struct xlz_reader xlz_reader;
xlz_reader *xlz_new_reader1(xlz_ctx *ctx, const uint8_t *in, size_t in_len, int *err); /* NewReader1 */
xlz_reader *xlz_new_reader2(xlz_ctx *ctx, const uint8_t *in, size_t in_len, int dict_size,
                            int *err);                                              /* NewReader2 */
/* NewLZMADecompressorForSevenZip(props, unpackSize, readers) reader1.go:32-61 */
xlz_reader *xlz_new_lzma_decompressor_for_sevenzip(xlz_ctx *ctx, const uint8_t *props,
                                                   size_t props_len, uint64_t unpack_size,
                                                   const uint8_t *const *readers,
                                                   const size_t *reader_lens, size_t n_readers,
                                                   int *err);
/* NewLZMA2DecompressorForSevenZip(props, _, readers) reader2.go:49-75 */
xlz_reader *xlz_new_lzma2_decompressor_for_sevenzip(xlz_ctx *ctx, const uint8_t *props,
                                                    size_t props_len, uint64_t unpack_size,
                                                    const uint8_t *const *readers,
                                                    const size_t *reader_lens, size_t n_readers,
                                                    int *err);
/* Read(p): returns bytes copied (>= 0).  *err: XLZ_OK while more may follow;
 * XLZ_EOF at end of stream; a negative status on error.                          */
#define XLZ_EOF 100
#define XLZ_NEED_INPUT 101 /* streaming input only: feed the next piece (or declare the end), read again */
long xlz_reader_read(xlz_reader *r, uint8_t *p, size_t n, int *err);
/* Streaming input: the reference's readers pull from an io.Reader as they go (reader1.go:18-24,
 * decompress.go:35: one ReadByte per normalisation); the C ABI takes buffers, so the pull is turned
 * around.  Construct the reader from the FIRST piece of the stream (at least the .lzma header / the
 * first LZMA2 chunk header and the five range-coder bytes: the constructors check those), call
 * xlz_reader_expect_more once, then: xlz_reader_read returns XLZ_NEED_INPUT whenever the decoder has
 * used up its input -- feed the next piece (any size; >= 128 KiB keeps LZMA2 chunks whole) or
 * declare the end, and read again.  The library keeps only the bytes the decoder has not consumed:
 * with this, a reader's memory is bounded on BOTH sides.  A fed reader decodes everything a reader
 * over the whole buffer decodes (models up to lc = 8, lp = 4 taken up in mid-stream; copies that read
 * behind an LZMA2 dictionary reset).  xlz_reader_expect_more is also accepted right after
 * xlz_reader_reopen: (*Reader1).Reopen takes an io.ByteReader (reader1.go:166-176), `in` is then the
 * first piece of the new stream (at least its five range-coder bytes).                            */
int xlz_reader_expect_more(xlz_reader *r);
int xlz_reader_feed(xlz_reader *r, const uint8_t *data, size_t n);
int xlz_reader_feed_eof(xlz_reader *r);
/* (*Reader1).Reset (reader1.go:161-164): the probability model, state and reps start over, the window
 * and the input position stay.  (*Reader1).Reopen (reader1.go:166-176): continue on a NEW raw LZMA
 * stream (no header) with the given unpack size (all-ones = unknown), same window and model; its
 * return value is rangeDec.Reopen's error (XLZ_ERR_HEADER_EOF = io.EOF, XLZ_ERR_RESULT).  Readers
 * made by xlz_new_reader1 / ..._lzma_decompressor_for_sevenzip only.                              */
int xlz_reader_reset(xlz_reader *r);
int xlz_reader_reopen(xlz_reader *r, const uint8_t *in, size_t in_len, uint64_t unpack_size);
int xlz_reader_close(xlz_reader *r); /* readCloser.Close (readcloser.go:16-28); second call ->
                                        XLZ_ERR_CLOSED; the handle stays valid until _free       */
void xlz_reader_free(xlz_reader *r);
/* launches that refilled this reader, whole-stream fallback decodes (0 unless one of the two
 * cases above), compressed bytes uploaded so far -- for tests of the one-pass property     */
int xlz_reader_stats(const xlz_reader *r, uint64_t *refills, uint64_t *whole_decodes,
                     uint64_t *in_uploaded);
/* device memory the reader holds right now: its sliding output window, and the image of the
 * reference's uncleared window buffer (dictSize bytes; window.go:135-140) -- 0 until the stream's
 * first dictionary reset behind a non-empty epoch, i.e. for nearly every stream                 */
int xlz_reader_memory(const xlz_reader *r, uint64_t *window_bytes, uint64_t *image_bytes);

/* Multi-GPU form of xlz_decode_batch: one context per GPU; every context
 * decodes its shard on its own host thread, results come back in input order.  No device-to-device
 * traffic.  The work is dealt by COMPRESSED bytes (what the kernel's own work queue is keyed by):
 * whole streams, and -- for a raw LZMA2 stream that is a large part of the call -- runs of its units
 * (xlz_lzma2_units: a dictionary reset with new properties starts an independent unit,
 * reader2.go:100-173), each GPU getting a slice of the compressed input and a disjoint slice of the
 * caller's output buffer.  A stream whose slices do not decode to exactly what their headers
 * announce (malformed streams only) is decoded again as a whole on one context, so bytes, status
 * and in_consumed are the single-GPU call's.                                                     */
int xlz_decode_batch_multi(xlz_ctx *const *ctxs, size_t n_ctx, const xlz_stream_desc *streams,
                           size_t n, xlz_result *results);
/* The plan of such a call alone (host only, no GPU): the items -- whole streams and runs of units of
 * LZMA2 streams -- and the context each goes to.  At most max_items entries are filled, *n_items =
 * their number (XLZ_ERR_OUT_CAP when larger than max_items > 0; max_items = 0 counts).  The items of
 * a stream are adjacent and in stream order.                                                      */
typedef struct xlz_multi_item {
    uint64_t stream;           /* index into streams                                               */
    uint64_t in_off, in_len;   /* the slice of the stream's input                                  */
    uint64_t out_off, out_len; /* the slice of its output (whole streams: 0, out_cap)              */
    uint32_t context;          /* index into ctxs                                                  */
    uint32_t flags;            /* 1 whole stream, 2 begins at the stream's start, 4 ends at its end */
} xlz_multi_item;
int xlz_decode_batch_multi_plan(size_t n_ctx, const xlz_stream_desc *streams, size_t n, xlz_multi_item *items,
                                size_t max_items, size_t *n_items);

/* ---- the unit plan of a raw LZMA2 stream (host only, no GPU needed) ------------------------
 * What a decode of `in` as XLZ_FMT_LZMA2_RAW launches: the stream is cut where a chunk starts
 * that depends on nothing before it (Reader2.startChunk, reader2.go:100-173: a dictionary reset
 * with new properties; a stored chunk that resets the dictionary when no LZMA chunk behind it
 * continues an earlier model; every 256 KiB inside a run of stored chunks that ends at a dictionary
 * reset or at the end of the stream) -- one wave per unit.  Fills at most `max_units` entries,
 * *n_units = the number of units (XLZ_ERR_OUT_CAP when larger than max_units > 0; pass
 * max_units = 0 to count).  Offsets are trusted from the chunk headers: a stream whose real
 * decode leaves them is decoded again as ONE unit after the launch.                          */
typedef struct xlz_lzma2_unit {
    uint64_t in_off, in_len;   /* bytes of `in` the unit walks                                   */
    uint64_t out_off, out_len; /* where its output goes and how much its headers announce        */
    uint32_t have_reader;      /* an LZMA chunk precedes it in the stream (Reader2.lzmaReader)   */
    uint32_t reserved;
} xlz_lzma2_unit;
int xlz_lzma2_units(const uint8_t *in, size_t len, xlz_lzma2_unit *units, size_t max_units, size_t *n_units);

/* ---- .xz container front-end ----------------------------
 * Outside the reference (which has no container code): an .xz file is a list of independent
 * blocks, each ONE raw LZMA2 stream with its own dictionary -- what NewReader2(in, dictSize)
 * takes (reader2.go:26-41) -- so a file is one batch.  Only filter chains made of a single
 * LZMA2 filter are accepted (BCJ / delta: XLZ_ERR_UNSUPPORTED).                              */
typedef struct xlz_xz_block {
    uint64_t comp_off;   /* raw LZMA2 payload inside the file                                   */
    uint64_t comp_len;
    uint64_t uncomp_off; /* where the block's bytes go in the decoded file                      */
    uint64_t uncomp_len;
    uint64_t check_off;  /* the block's integrity check inside the file                         */
    uint32_t dict_size;
    uint32_t check_type; /* 0 none, 1 CRC32, 4 CRC64, 10 SHA-256                                */
} xlz_xz_block;

/* Block index of a whole .xz file (concatenated streams and stream padding included), host
 * only.  blocks may be NULL with max_blocks 0 to obtain the counts.  XLZ_ERR_OUT_CAP: more
 * blocks than max_blocks (*n_blocks is the full count).                                      */
int xlz_xz_index(const uint8_t *file, size_t len, xlz_xz_block *blocks, size_t max_blocks,
                 size_t *n_blocks, uint64_t *total_uncompressed);
/* Decode a whole .xz file into out as ONE GPU batch.  verify != 0: check every block's CRC32 /
 * CRC64 / SHA-256 on the host; *unverified (optional) = number of blocks whose check type is a
 * reserved one.  A failed check or a block that does not match the index:
 * XLZ_ERR_RESULT.                                                                             */
int xlz_xz_decode(xlz_ctx *ctx, const uint8_t *file, size_t len, uint8_t *out, size_t out_cap,
                  uint64_t *out_len, int verify, size_t *unverified);
/* the same over several contexts (one per GPU): the blocks -- and the units inside large blocks -- are
 * dealt to the contexts by xlz_decode_batch_multi                                                */
int xlz_xz_decode_multi(xlz_ctx *const *ctxs, size_t n_ctx, const uint8_t *file, size_t len, uint8_t *out,
                        size_t out_cap, uint64_t *out_len, int verify, size_t *unverified);

/* ---- .7z container front-end ----------------------------
 * (The parser was written from 7-Zip's published format description.  It is exercised on archives
 * built from that description by tests/sevenzip_craft.py and their mutations AND on archives by an
 * independent writer -- libarchive's 7zip writer, which `cmake -E tar cf x.7z --format=7zip` drives
 * in this image: solid LZMA1 folders, LZMA-encoded headers, per-file CRCs, an 8 MiB dictionary that
 * wraps; tests/golden/libarchive_solid.7z is one of them.  7-Zip's own binary is not in the image.)
 * Outside the reference, which only offers the two bodgit/sevenzip decompressor constructors
 * (reader1.go:28-61 method 03 01 01, reader2.go:45-75 method 21).  A .7z archive keeps its data in
 * folders, each ONE compressed stream with out-of-band properties -- exactly what those
 * constructors take -- and folders are independent, so an archive is one batch.  Folders with a
 * single LZMA, LZMA2 or Copy coder are decoded; coder chains (BCJ + LZMA ...), encryption and
 * external / multi-volume layouts are reported as unsupported.  File names are not parsed: the
 * output is the folders' bytes back to back = the archive's files back to back.                */
enum { XLZ_7Z_UNSUPPORTED = 0, XLZ_7Z_LZMA = 1, XLZ_7Z_LZMA2 = 2, XLZ_7Z_COPY = 3 };
typedef struct xlz_7z_folder {
    uint64_t pack_off;   /* the folder's packed stream inside the file                           */
    uint64_t pack_len;
    uint64_t unpack_off; /* where its bytes go in the decoded output                             */
    uint64_t unpack_len;
    uint32_t method;     /* XLZ_7Z_*                                                             */
    uint32_t dict_size;  /* LZMA: LE32 of props[1:5]; LZMA2: DecodeDictSize2(props[0])           */
    uint32_t crc;        /* CRC32 of the folder's output when has_crc                            */
    uint32_t first_substream, n_substreams; /* its files in the substream array                  */
    uint8_t props;       /* LZMA: the lc/lp/pb byte; LZMA2: the dictionary byte                  */
    uint8_t has_crc;
    uint8_t reserved[2];
} xlz_7z_folder;
typedef struct xlz_7z_substream { /* one file's bytes inside a (solid) folder                    */
    uint64_t size;
    uint32_t crc;
    uint32_t has_crc;
} xlz_7z_substream;
/* Folder list of a .7z archive.  An encoded (compressed) header -- what 7-Zip writes by default --
 * is itself an LZMA folder and is decoded on the GPU first: ctx may be NULL only for archives with a
 * plain header.  Arrays may be NULL with capacity 0 to obtain the counts; XLZ_ERR_OUT_CAP when a
 * non-zero capacity is too small (the counts are still set).                                      */
int xlz_7z_index(xlz_ctx *ctx, const uint8_t *file, size_t len, xlz_7z_folder *folders,
                 size_t max_folders, size_t *n_folders, xlz_7z_substream *substreams,
                 size_t max_substreams, size_t *n_substreams, uint64_t *total_unpacked);
/* Decode every folder of a .7z archive into out as ONE GPU batch.  verify != 0: CRC32 of every file
 * (or folder) that carries one; *unverified (optional) = folders without any CRC.  XLZ_ERR_UNSUPPORTED
 * when a folder's coder chain is not a single LZMA / LZMA2 / Copy coder.                         */
int xlz_7z_decode(xlz_ctx *ctx, const uint8_t *file, size_t len, uint8_t *out, size_t out_cap,
                  uint64_t *out_len, int verify, size_t *unverified);
/* the same over several contexts (one per GPU; encoded headers are decoded on the first)         */
int xlz_7z_decode_multi(xlz_ctx *const *ctxs, size_t n_ctx, const uint8_t *file, size_t len, uint8_t *out,
                        size_t out_cap, uint64_t *out_len, int verify, size_t *unverified);

#ifdef __cplusplus
}
#endif
#endif /* XLZ_H */
