// Package lzmagpu is the cgo shim a maintainer of kulaginds/lzma would add to route the
// decode hot path ((*Reader1).decompress and the LZMA2 framing around it) through libxlz.so,
// the MI355X decoder declared in include/xlz.h.  It keeps the reference's constructors, the
// concrete *Reader1 / *Reader2 types with their Read / Reset / Reopen methods and the
// io.ReadCloser of the sevenzip constructors.  Both sides are streamed: a constructor reads the
// first MiB of its source, Read pulls the next MiB whenever the decoder asks for input
// (XLZ_NEED_INPUT -> xlz_reader_feed), and the decoded side arrives one refill chunk at a time
// (include/xlz.h) -- a reader's memory does not depend on the stream's size.
//
// Below the break-even (16 units per host thread: xlz_batch_advice) the constructors return the reference's OWN
// readers -- the library has no CPU decoder, the reference package is the CPU side (SetExpectedConcurrency).
//
// NOT COMPILED IN THIS REPOSITORY: the build image has no Go toolchain (go, gccgo: not found).
// The same C entry points are exercised by a plain C caller (tests/c/reader_demo.c) and through
// ctypes (lzma_amd/__init__.py, tests/).  cgo rules this file follows: no Go pointer to memory
// that itself holds Go pointers crosses into C (arrays of descriptors are C.malloc'ed and the
// buffers they point to are pinned with runtime.Pinner for the duration of the call); C handles
// are released by Close or by a finalizer.
package lzmagpu

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../lzma_amd -lxlz -Wl,-rpath,${SRCDIR}/../../lzma_amd
#include <stdlib.h>
#include "xlz.h"
*/
import "C"

import (
	"bytes"
	"errors"
	"fmt"
	"io"
	"runtime"
	"sync"
	"sync/atomic"
	"unsafe"

	ref "github.com/kulaginds/lzma" // the reference package itself: the CPU side of the break-even (below)
)

// ---- the CPU side of the break-even ------------------------------------------------------------------
// One GPU wave decodes one unit -- an LZMA1 stream, one dictionary-reset unit of an LZMA2 stream -- about 16 times
// slower than one host core decodes it (include/xlz.h: xlz_batch_advice; bench.py: stream_count_sweep), so the GPU
// only wins when a set of concurrent LZMA1 readers brings at least 16 units per host thread the application
// could use instead -- or when ONE LZMA2 stream is many units: on the host it is one goroutine whatever it holds.  libxlz.so has no CPU decoder (and must not have one: the product path fails loudly without
// a device); the shim's fallback is the reference's OWN readers (reader1.go:18-24, reader2.go:26-41): below the
// break-even the constructors return a Reader1 / Reader2 that wraps lzma.NewReader1 / lzma.NewReader2.
//
// What the shim cannot see is how many readers the application runs at once: SetExpectedConcurrency says it
// (default 1: a lone NewReader1 + io.Copy stays on the CPU, exactly as fast as before the shim was linked in;
// a server that keeps hundreds of readers in flight sets it once and gets coalesced GPU launches).

var expectedReaders atomic.Int64

// HostThreads is the number of host threads the application would decode with (0: GOMAXPROCS).
var HostThreads = 0

// ForceGPU routes every reader to the device whatever the break-even says (tests, benchmarks).
var ForceGPU = false

func init() { expectedReaders.Store(1) }

// SetExpectedConcurrency tells the shim how many readers the application keeps in flight at once.
func SetExpectedConcurrency(n int) {
	if n < 1 {
		n = 1
	}
	expectedReaders.Store(int64(n))
}

// BreakEvenUnits is xlz_batch_advice's break-even for EQUAL LZMA1 streams: 16 units per host thread.
func BreakEvenUnits() int {
	return coreOverWave * hostThreads()
}

func hostThreads() int {
	t := HostThreads
	if t <= 0 {
		t = runtime.GOMAXPROCS(0)
	}
	return t
}

// coreOverWave: a host core decodes a stream this many times as fast as one wave decodes a unit; waveSlots: units an
// MI355X decodes at once (xlz_batch_advice: kCoreOverWave, wave_slots).
const (
	coreOverWave = 16
	waveSlots    = 4096
)

// worthTheDevice is xlz_batch_advice's rule for ONE reader among the readers expected in flight: two estimated times in
// compressed bytes of serial work.  On the host this reader's stream is ONE goroutine whatever its format (reader1.go,
// reader2.go:216-250: the units of an LZMA2 stream are parallel work for the device only), and the readers share the
// threads; on the device every unit is a wave.  streamBytes: the stream (or the part of it at hand), maxUnit: its longest
// unit (= streamBytes for LZMA1 and for an LZMA2 stream without inner dictionary resets).  Round 4 compared a unit COUNT
// with 16 x threads: one LZMA2 stream of 100 units on 16 threads went to one host thread.
func worthTheDevice(streamBytes, maxUnit int) bool {
	if ForceGPU {
		return true
	}
	e, t := float64(expectedReaders.Load()), float64(hostThreads())
	perThread := 1.0 // readers per host thread, at least the one
	if e > t {
		perThread = e / t
	}
	cpu := float64(streamBytes) * perThread / coreOverWave
	gpu := float64(maxUnit)
	if all := e * float64(streamBytes) / waveSlots; all > gpu {
		gpu = all
	}
	return gpu <= cpu
}

// lzma2LongestUnit is the longest unit the device would launch for the part of a raw LZMA2 stream at hand
// (xlz_lzma2_units, host only): a stream written by a multi-threaded encoder is hundreds of units although it is one reader.
func lzma2LongestUnit(data []byte) int {
	p, n := cbuf(data)
	var units C.size_t
	if st := C.xlz_lzma2_units(p, n, nil, 0, &units); st != C.XLZ_OK || units == 0 {
		return len(data)
	}
	plan := make([]C.xlz_lzma2_unit, int(units))
	if st := C.xlz_lzma2_units(p, n, &plan[0], units, &units); st != C.XLZ_OK {
		return len(data)
	}
	runtime.KeepAlive(data)
	longest := 0
	for i := range plan {
		if l := int(plan[i].in_len); l > longest {
			longest = l
		}
	}
	return longest
}

// The reference's sentinels (errors.go:5-12, reader1.go:26, reader2.go:43, readcloser.go:14).
var (
	// (the reference's own values: a reader below the break-even IS the reference's, and errors.Is must not care which side decoded)
	ErrResultError            = ref.ErrResultError
	ErrIncorrectProperties    = ref.ErrIncorrectProperties
	errNeedOneReader          = errors.New("lzma: need exactly one reader")
	errInsufficientProperties = errors.New("lzma2: not enough properties")
	errAlreadyClosed          = errors.New("lzma: already closed")
	ErrDevice                 = errors.New("lzma: HIP device error")
	ErrUnsupported            = errors.New("lzma: stream not supported by the GPU path")
)

// readError maps a status of xlz_reader_read / a per-stream batch status onto the value the
// reference's Read returns.
func readError(st C.int) error {
	switch st {
	case C.XLZ_OK, C.XLZ_OK_INPUT_EOF:
		return nil
	case C.XLZ_EOF:
		return io.EOF
	case C.XLZ_ERR_RESULT, C.XLZ_ERR_RC_INIT:
		return ErrResultError
	case C.XLZ_ERR_PROPS:
		return ErrIncorrectProperties
	case C.XLZ_ERR_HEADER_EOF: // "rangeDec.Init: EOF" out of a first LZMA2 chunk (reader2.go:146-153)
		return fmt.Errorf("rangeDec.Init: %w", io.EOF)
	case C.XLZ_ERR_UNEXPECTED_EOF:
		return io.ErrUnexpectedEOF
	case C.XLZ_ERR_CLOSED:
		return errAlreadyClosed
	case C.XLZ_ERR_UNSUPPORTED:
		return ErrUnsupported
	}
	return ErrDevice
}

// reader1CtorError reproduces the wrapping of initializeFull / initialize (reader1.go:77-159).
// The C constructor reports WHAT failed; WHERE follows from how much of the header was there.
func reader1CtorError(st C.int, headerBytes int) error {
	switch st {
	case C.XLZ_ERR_PROPS:
		return fmt.Errorf("decode prop: %w", ErrIncorrectProperties) // reader1.go:85
	case C.XLZ_ERR_HEADER_EOF:
		switch {
		case headerBytes == 0:
			return io.EOF // the very first ReadByte fails: returned as is (reader1.go:78-81)
		case headerBytes < 5:
			return fmt.Errorf("decode dict size: %w", io.EOF) // reader1.go:90
		case headerBytes < 13:
			return fmt.Errorf("decode unpack size: %w", io.EOF) // reader1.go:97
		}
		return fmt.Errorf("rangeDec.Init: %w", io.EOF) // reader1.go:155
	case C.XLZ_ERR_RC_INIT:
		return fmt.Errorf("rangeDec.Init: %w", ErrResultError) // reader1.go:155
	case C.XLZ_ERR_UNSUPPORTED:
		return ErrUnsupported
	}
	return ErrDevice
}

var (
	ctxOnce sync.Once
	ctx     *C.xlz_ctx
	ctxErr  error
)

func context() (*C.xlz_ctx, error) {
	ctxOnce.Do(func() {
		if st := C.xlz_ctx_create(0, &ctx); st != C.XLZ_OK {
			ctxErr = ErrDevice
			return
		}
		// concurrent readers (one goroutine each, as with the reference) share launches
		C.xlz_ctx_enable_batching(ctx, 200, 4096)
	})
	return ctx, ctxErr
}

const pieceSize = 1 << 20 // compressed bytes pulled from the source at a time

// handle owns an xlz_reader; freed by Close-then-finalizer or by the finalizer alone.  src is the
// rest of the compressed stream (nil once it has ended or when the whole stream was given).
type handle struct {
	h   *C.xlz_reader
	src io.Reader
}

func newHandle(h *C.xlz_reader, src io.Reader) *handle {
	r := &handle{h: h, src: src}
	if src != nil {
		C.xlz_reader_expect_more(h)
	}
	runtime.SetFinalizer(r, func(r *handle) { C.xlz_reader_free(r.h) })
	return r
}

// firstPiece reads up to pieceSize bytes; the returned reader is nil when the source ended inside them.
func firstPiece(src io.Reader) ([]byte, io.Reader, error) {
	buf := make([]byte, pieceSize)
	n, err := io.ReadFull(src, buf)
	if err == io.EOF || err == io.ErrUnexpectedEOF {
		return buf[:n], nil, nil
	}
	if err != nil {
		return nil, nil, err
	}
	return buf, src, nil
}

func (r *handle) read(p []byte) (int, error) {
	if len(p) == 0 {
		return 0, nil // the reference never returns from Read(p) with len(p) == 0 (SURVEY parity note 6)
	}
	got := 0
	for {
		var st C.int
		// p is a Go slice of bytes (no pointers inside): passing &p[got] for the duration of the call is allowed
		n := C.xlz_reader_read(r.h, (*C.uint8_t)(unsafe.Pointer(&p[got])), C.size_t(len(p)-got), &st)
		got += int(n)
		if st != C.XLZ_NEED_INPUT {
			runtime.KeepAlive(r)
			return got, readError(st)
		}
		// the decoder has used up its input: the next piece of the source, or its end
		piece := make([]byte, pieceSize)
		k, err := io.ReadFull(r.src, piece)
		if k > 0 {
			C.xlz_reader_feed(r.h, (*C.uint8_t)(unsafe.Pointer(&piece[0])), C.size_t(k)) // copied by the library
		}
		if err == io.EOF || err == io.ErrUnexpectedEOF {
			C.xlz_reader_feed_eof(r.h)
		} else if err != nil {
			return got, err // the source's own error, as the reference's ReadByte would surface it
		}
		if got == len(p) {
			return got, nil
		}
	}
}

// Reader1 replaces lzma.Reader1 (reader1.go:10-16).  cpu != nil: below the break-even -- the reference's own reader.
type Reader1 struct {
	*handle
	cpu *ref.Reader1
}

// OnDevice reports whether this reader decodes on the GPU (false: the reference's CPU reader, below the break-even).
func (r *Reader1) OnDevice() bool { return r.cpu == nil }

// Read is (*Reader1).Read (reader1.go:223-254).
func (r *Reader1) Read(p []byte) (int, error) {
	if r.cpu != nil {
		return r.cpu.Read(p)
	}
	return r.read(p)
}

// Reset is (*Reader1).Reset (reader1.go:161-164).
func (r *Reader1) Reset() {
	if r.cpu != nil {
		r.cpu.Reset()
		return
	}
	C.xlz_reader_reset(r.h)
	runtime.KeepAlive(r)
}

// Reopen is (*Reader1).Reopen (reader1.go:166-176): a new raw stream on the same window and model.  Like the
// constructors it takes the FIRST piece of the source now and feeds the rest as the decoder asks for it
// (xlz_reader_expect_more is accepted right after xlz_reader_reopen): the source is never slurped.
func (r *Reader1) Reopen(inStream io.ByteReader, unpackSize uint64) error {
	if r.cpu != nil {
		return r.cpu.Reopen(inStream, unpackSize)
	}
	data, rest, err := firstPiece(asReader(inStream))
	if err != nil {
		return err
	}
	p, n := cbuf(data)
	st := C.xlz_reader_reopen(r.h, p, n, C.uint64_t(unpackSize)) // the library copies the bytes
	runtime.KeepAlive(data)
	r.src = nil
	switch st {
	case C.XLZ_OK:
		if rest != nil {
			C.xlz_reader_expect_more(r.h)
			r.src = rest
		}
		runtime.KeepAlive(r)
		return nil
	case C.XLZ_ERR_HEADER_EOF:
		return io.EOF // rangeDec.Reopen returns Init's error unwrapped (reader1.go:170-173)
	}
	return readError(st)
}

// Reader2 replaces lzma.Reader2 (reader2.go:10-24).  cpu != nil: below the break-even -- the reference's own reader.
type Reader2 struct {
	*handle
	cpu *ref.Reader2
}

// OnDevice reports whether this reader decodes on the GPU.
func (r *Reader2) OnDevice() bool { return r.cpu == nil }

// Read is (*Reader2).Read (reader2.go:216-250).
func (r *Reader2) Read(p []byte) (int, error) {
	if r.cpu != nil {
		return r.cpu.Read(p)
	}
	return r.read(p)
}

// readCloser is readcloser.go:9-41.
type readCloser struct {
	*handle
	c io.Closer
}

func (rc *readCloser) Read(p []byte) (int, error) {
	n, err := rc.read(p)
	if err != nil && err != io.EOF && err != errAlreadyClosed {
		err = fmt.Errorf("lzma: error reading: %w", err) // readcloser.go:36-38
	}
	return n, err
}

func (rc *readCloser) Close() error {
	if st := C.xlz_reader_close(rc.h); st != C.XLZ_OK {
		return errAlreadyClosed // readcloser.go:17-19
	}
	runtime.KeepAlive(rc)
	if err := rc.c.Close(); err != nil {
		return fmt.Errorf("lzma: error closing: %w", err) // readcloser.go:21-23
	}
	return nil
}

func cbuf(b []byte) (*C.uint8_t, C.size_t) {
	if len(b) == 0 {
		return nil, 0
	}
	return (*C.uint8_t)(unsafe.Pointer(&b[0])), C.size_t(len(b))
}

// NewReader1 replaces lzma.NewReader1 (reader1.go:18-24).
func NewReader1(inStream io.ByteReader) (*Reader1, error) {
	if !worthTheDevice(1, 1) { // one LZMA1 stream is one unit of its own length: the reference's own reader (reader1.go:18-24), untouched source
		cpu, err := ref.NewReader1(inStream)
		if err != nil {
			return nil, err
		}
		return &Reader1{cpu: cpu}, nil
	}
	c, err := context()
	if err != nil {
		return nil, err
	}
	data, rest, err := firstPiece(asReader(inStream))
	if err != nil {
		return nil, err
	}
	p, n := cbuf(data)
	var st C.int
	h := C.xlz_new_reader1(c, p, n, &st) // the library copies the bytes
	runtime.KeepAlive(data)
	if h == nil {
		return nil, reader1CtorError(st, len(data))
	}
	return &Reader1{handle: newHandle(h, rest)}, nil
}

// asReader: bufio.Reader, bytes.Reader ... are io.Readers already; a bare io.ByteReader is adapted.
func asReader(br io.ByteReader) io.Reader {
	if r, ok := br.(io.Reader); ok {
		return r
	}
	return byteReaderAdapter{br}
}

type byteReaderAdapter struct{ br io.ByteReader }

func (a byteReaderAdapter) Read(p []byte) (int, error) {
	for i := range p {
		b, err := a.br.ReadByte()
		if err != nil {
			return i, err
		}
		p[i] = b
	}
	return len(p), nil
}

// NewReader2 replaces lzma.NewReader2 (reader2.go:26-41).
func NewReader2(inStream io.Reader, dictSize int) (*Reader2, error) {
	data, rest, err := firstPiece(inStream)
	if err != nil {
		return nil, err
	}
	// The units of the part at hand (the whole stream when it is shorter than a piece; a lower bound otherwise): one
	// LZMA2 stream of many dictionary-reset units fills the chip by itself, one without inner resets is one wave.
	if !worthTheDevice(len(data), lzma2LongestUnit(data)) {
		var src io.Reader = bytes.NewReader(data)
		if rest != nil {
			src = io.MultiReader(src, rest)
		}
		cpu, err := ref.NewReader2(src, dictSize) // reader2.go:26-41
		if err != nil {
			return nil, err
		}
		return &Reader2{cpu: cpu}, nil
	}
	c, err := context() // (only a reader that goes to the device needs one)
	if err != nil {
		return nil, err
	}
	p, n := cbuf(data)
	var st C.int
	h := C.xlz_new_reader2(c, p, n, C.int(dictSize), &st)
	runtime.KeepAlive(data)
	if h == nil {
		switch st { // startChunk's errors, returned unwrapped by the constructor (reader2.go:77-86)
		case C.XLZ_ERR_PROPS:
			return nil, ErrIncorrectProperties
		case C.XLZ_ERR_HEADER_EOF:
			return nil, fmt.Errorf("rangeDec.Init: %w", io.EOF)
		case C.XLZ_ERR_RC_INIT:
			return nil, fmt.Errorf("rangeDec.Init: %w", ErrResultError)
		}
		return nil, readError(st)
	}
	return &Reader2{handle: newHandle(h, rest)}, nil
}

// NewLZMADecompressorForSevenZip replaces the bodgit/sevenzip constructor of reader1.go:32-61.
//
//	sevenzip.RegisterDecompressor([]byte{0x03, 0x01, 0x01}, sevenzip.Decompressor(lzmagpu.NewLZMADecompressorForSevenZip))
func NewLZMADecompressorForSevenZip(props []byte, unpackSize uint64, readers []io.ReadCloser) (io.ReadCloser, error) {
	return sevenzip(false, props, unpackSize, readers)
}

// NewLZMA2DecompressorForSevenZip replaces reader2.go:49-75 (method id 0x21).
func NewLZMA2DecompressorForSevenZip(props []byte, unpackSize uint64, readers []io.ReadCloser) (io.ReadCloser, error) {
	return sevenzip(true, props, unpackSize, readers)
}

func sevenzip(lzma2 bool, props []byte, unpackSize uint64, readers []io.ReadCloser) (io.ReadCloser, error) {
	if len(readers) != 1 {
		return nil, errNeedOneReader
	}
	if lzma2 && len(props) != 1 {
		return nil, errInsufficientProperties // reader2.go:54-56
	}
	if !lzma2 && !worthTheDevice(1, 1) { // an LZMA folder is one unit of its own length: the reference's constructor (reader1.go:32-61)
		return ref.NewLZMADecompressorForSevenZip(props, unpackSize, readers)
	}
	data, rest, err := firstPiece(readers[0])
	if err != nil {
		return nil, err
	}
	if lzma2 && !worthTheDevice(len(data), lzma2LongestUnit(data)) { // reader2.go:49-75 on what has been pulled + the rest of the source
		var src io.Reader = bytes.NewReader(data)
		if rest != nil {
			src = io.MultiReader(src, rest)
		}
		return ref.NewLZMA2DecompressorForSevenZip(props, unpackSize, []io.ReadCloser{&prefixed{Reader: src, c: readers[0]}})
	}
	c, err := context()
	if err != nil {
		return nil, err
	}
	dp, dn := cbuf(data)
	pp, pn := cbuf(props)
	// the one-element pointer and length arrays live in C memory; the buffers they point to are pinned
	ptrs := (**C.uint8_t)(C.malloc(C.size_t(unsafe.Sizeof(dp))))
	lens := (*C.size_t)(C.malloc(C.size_t(unsafe.Sizeof(dn))))
	defer C.free(unsafe.Pointer(ptrs))
	defer C.free(unsafe.Pointer(lens))
	var pin runtime.Pinner
	defer pin.Unpin()
	if dp != nil {
		pin.Pin(dp)
	}
	*ptrs, *lens = dp, dn
	var st C.int
	var h *C.xlz_reader
	if lzma2 {
		h = C.xlz_new_lzma2_decompressor_for_sevenzip(c, pp, pn, C.uint64_t(unpackSize), ptrs, lens, 1, &st)
	} else {
		h = C.xlz_new_lzma_decompressor_for_sevenzip(c, pp, pn, C.uint64_t(unpackSize), ptrs, lens, 1, &st)
	}
	runtime.KeepAlive(props)
	if h == nil {
		if lzma2 {
			switch st {
			case C.XLZ_ERR_PROPS:
				return nil, ErrIncorrectProperties
			case C.XLZ_ERR_HEADER_EOF:
				return nil, fmt.Errorf("rangeDec.Init: %w", io.EOF)
			case C.XLZ_ERR_RC_INIT:
				return nil, fmt.Errorf("rangeDec.Init: %w", ErrResultError)
			}
			return nil, readError(st)
		}
		// NewLZMADecompressorForSevenZip returns DecodeProp's error unwrapped (reader1.go:37-40) and
		// initialize's wrapped (:57-60)
		if st == C.XLZ_ERR_PROPS {
			return nil, ErrIncorrectProperties
		}
		return nil, reader1CtorError(st, 13)
	}
	return &readCloser{handle: newHandle(h, rest), c: readers[0]}, nil
}

// prefixed: the bytes already pulled from a source in front of the source itself, closing the source.
type prefixed struct {
	io.Reader
	c io.Closer
}

func (p *prefixed) Close() error { return p.c.Close() }

// DecodeBatch is the new entry the reference has no analogue for: n independent .lzma
// streams decoded concurrently on the GPU (one wave per stream).  outs[i] must have the
// capacity of stream i's decoded size.  Returns the decoded lengths and per-stream errors.
func DecodeBatch(streams [][]byte, outs [][]byte) ([]int, []error, error) {
	c, err := context()
	if err != nil {
		return nil, nil, err
	}
	n := len(streams)
	if n != len(outs) {
		return nil, nil, errors.New("lzma: DecodeBatch needs one output buffer per stream")
	}
	if n == 0 {
		return nil, nil, nil
	}
	// descriptor and result arrays in C memory (they hold pointers); every Go buffer they point
	// to is pinned until the call returns
	descs := (*C.xlz_stream_desc)(C.calloc(C.size_t(n), C.size_t(unsafe.Sizeof(C.xlz_stream_desc{}))))
	results := (*C.xlz_result)(C.calloc(C.size_t(n), C.size_t(unsafe.Sizeof(C.xlz_result{}))))
	if descs == nil || results == nil {
		C.free(unsafe.Pointer(descs))
		C.free(unsafe.Pointer(results))
		return nil, nil, ErrDevice
	}
	defer C.free(unsafe.Pointer(descs))
	defer C.free(unsafe.Pointer(results))
	dv := unsafe.Slice(descs, n)
	rv := unsafe.Slice(results, n)
	var pin runtime.Pinner
	defer pin.Unpin()
	for i := range streams {
		ip, il := cbuf(streams[i])
		op, ol := cbuf(outs[i][:cap(outs[i])])
		if ip != nil {
			pin.Pin(ip)
		}
		if op != nil {
			pin.Pin(op)
		}
		dv[i].in, dv[i].in_len = ip, il
		dv[i].out, dv[i].out_cap = op, ol
		dv[i].format = C.XLZ_FMT_LZMA_ALONE
	}
	if st := C.xlz_decode_batch(c, descs, C.size_t(n), results); st != C.XLZ_OK {
		return nil, nil, ErrDevice
	}
	lens := make([]int, n)
	errs := make([]error, n)
	for i := range rv {
		lens[i] = int(rv[i].out_len)
		switch rv[i].status {
		case C.XLZ_ERR_PROPS, C.XLZ_ERR_HEADER_EOF, C.XLZ_ERR_RC_INIT: // what NewReader1 would have returned
			errs[i] = reader1CtorError(rv[i].status, len(streams[i]))
		default:
			errs[i] = readError(rv[i].status)
		}
	}
	return lens, errs, nil
}

// SetSlicing tunes how DecodeBatch overlaps its copies with its decode when a call is ONE wave round of streams
// (xlz_ctx_set_slicing): such a call runs as up to maxSlices launches that each advance every stream by a share of its
// output, and share k-1 goes to the callers' buffers while share k decodes -- the batch form of what Read does for one
// reader (reader1.go:223-254: decompress(need) produces, window.ReadPending drains).  0 = the default of that argument;
// maxSlices = 1 turns it off.  The decoded bytes and errors do not depend on it.
func SetSlicing(minCallBytes, sliceBytes uint64, maxSlices uint32) error {
	c, err := context()
	if err != nil {
		return err
	}
	if st := C.xlz_ctx_set_slicing(c, C.uint64_t(minCallBytes), C.uint64_t(sliceBytes), C.uint32_t(maxSlices)); st != C.XLZ_OK {
		return ErrDevice
	}
	return nil
}

// Trim gives back the device and pinned memory the context keeps between DecodeBatch calls (xlz_ctx_trim): a call of the
// same shape as the one before finds its blocks again; a process that is done with large batches need not hold them.
func Trim() (released uint64, err error) {
	c, err := context()
	if err != nil {
		return 0, err
	}
	var n C.uint64_t
	if st := C.xlz_ctx_trim(c, &n); st != C.XLZ_OK {
		return 0, ErrDevice
	}
	return uint64(n), nil
}

// DecodeXZ decodes a whole .xz file (all streams, all blocks) as ONE GPU batch: every block is a
// raw LZMA2 stream of its own -- what NewReader2(in, dictSize) takes -- so the file's block index
// is the batch (xlz_xz_index / xlz_xz_decode, include/xlz.h).  Not part of the reference, which has
// no container code; verify checks each block's CRC32 / CRC64.
func DecodeXZ(file []byte, verify bool) ([]byte, error) {
	c, err := context()
	if err != nil {
		return nil, err
	}
	fp, fl := cbuf(file)
	var nBlocks C.size_t
	var total C.uint64_t
	if st := C.xlz_xz_index(fp, fl, nil, 0, &nBlocks, &total); st != C.XLZ_OK {
		return nil, readError(st)
	}
	out := make([]byte, int(total)+1)
	op, _ := cbuf(out)
	var outLen C.uint64_t
	v := C.int(0)
	if verify {
		v = 1
	}
	st := C.xlz_xz_decode(c, fp, fl, op, C.size_t(total), &outLen, v, nil)
	runtime.KeepAlive(file)
	if st != C.XLZ_OK {
		return nil, readError(st)
	}
	return out[:int(outLen)], nil
}
