// Package lzmagpu is the cgo shim a maintainer of kulaginds/lzma would add to route the
// decode hot path ((*Reader1).decompress and the LZMA2 framing around it) through libxlz.so,
// the MI355X decoder declared in include/xlz.h.  It keeps the reference's constructors
// and the io.Reader / io.ReadCloser surface; the GPU needs a whole compressed stream, so a
// constructor slurps its source first.
//
// NOT COMPILED IN THIS REPOSITORY: the build image has no Go toolchain.  The same C entry
// points are exercised through ctypes (lzma_amd/__init__.py, tests/test_gpu_parity.py).
package lzmagpu

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../lzma_amd -lxlz -Wl,-rpath,${SRCDIR}/../../lzma_amd
#include <stdlib.h>
#include "xlz.h"
*/
import "C"

import (
	"errors"
	"fmt"
	"io"
	"sync"
	"unsafe"
)

// The reference's sentinels (errors.go:5-12, reader1.go:26, reader2.go:43, readcloser.go:14).
var (
	ErrResultError            = errors.New("result error")
	ErrIncorrectProperties    = errors.New("incorrect LZMA properties")
	errNeedOneReader          = errors.New("lzma: need exactly one reader")
	errInsufficientProperties = errors.New("lzma2: not enough properties")
	errAlreadyClosed          = errors.New("lzma: already closed")
	ErrDevice                 = errors.New("lzma: HIP device error")
	ErrUnsupported            = errors.New("lzma: stream not supported by the GPU path")
)

// statusToError maps include/xlz.h status codes back onto the values the reference returns.
func statusToError(st C.int, constructor bool) error {
	switch st {
	case C.XLZ_OK, C.XLZ_OK_INPUT_EOF:
		return nil
	case C.XLZ_EOF:
		return io.EOF
	case C.XLZ_ERR_RESULT:
		return ErrResultError
	case C.XLZ_ERR_PROPS:
		if constructor {
			return fmt.Errorf("decode prop: %w", ErrIncorrectProperties) // reader1.go:85
		}
		return ErrIncorrectProperties
	case C.XLZ_ERR_HEADER_EOF:
		return io.EOF // wrapped "decode dict size: %w" etc. by the caller's context if needed
	case C.XLZ_ERR_RC_INIT:
		return fmt.Errorf("rangeDec.Init: %w", ErrResultError) // reader1.go:155
	case C.XLZ_ERR_UNEXPECTED_EOF:
		return io.ErrUnexpectedEOF
	case C.XLZ_ERR_CLOSED:
		return errAlreadyClosed
	case C.XLZ_ERR_NEED_ONE_READER:
		return errNeedOneReader
	case C.XLZ_ERR_INSUFFICIENT_PROPS:
		return errInsufficientProperties
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
		}
	})
	return ctx, ctxErr
}

// reader wraps an xlz_reader handle; it is what NewReader1 / NewReader2 return.
type reader struct {
	h      *C.xlz_reader
	closer io.Closer // non-nil for the sevenzip constructors (readCloser, readcloser.go:9-12)
	wrap   bool
}

func (r *reader) Read(p []byte) (int, error) {
	if len(p) == 0 {
		return 0, nil // the reference spins forever here (SURVEY parity note 6)
	}
	var st C.int
	n := C.xlz_reader_read(r.h, (*C.uint8_t)(unsafe.Pointer(&p[0])), C.size_t(len(p)), &st)
	err := statusToError(st, false)
	if err != nil && err != io.EOF && r.wrap {
		err = fmt.Errorf("lzma: error reading: %w", err) // readcloser.go:36-38
	}
	return int(n), err
}

func (r *reader) Close() error {
	if st := C.xlz_reader_close(r.h); st != C.XLZ_OK {
		return errAlreadyClosed // readcloser.go:17-19
	}
	if r.closer != nil {
		if err := r.closer.Close(); err != nil {
			return fmt.Errorf("lzma: error closing: %w", err) // readcloser.go:21-23
		}
	}
	return nil
}

func slurp(br io.ByteReader) []byte {
	var buf []byte
	for {
		b, err := br.ReadByte()
		if err != nil {
			return buf
		}
		buf = append(buf, b)
	}
}

func cbuf(b []byte) (*C.uint8_t, C.size_t) {
	if len(b) == 0 {
		return nil, 0
	}
	return (*C.uint8_t)(unsafe.Pointer(&b[0])), C.size_t(len(b))
}

// NewReader1 replaces lzma.NewReader1 (reader1.go:18-24).
func NewReader1(inStream io.ByteReader) (io.Reader, error) {
	c, err := context()
	if err != nil {
		return nil, err
	}
	data := slurp(inStream)
	p, n := cbuf(data)
	var st C.int
	h := C.xlz_new_reader1(c, p, n, &st) // the library copies the bytes
	if h == nil {
		return nil, statusToError(st, true)
	}
	return &reader{h: h}, nil
}

// NewReader2 replaces lzma.NewReader2 (reader2.go:26-41).
func NewReader2(inStream io.Reader, dictSize int) (io.Reader, error) {
	c, err := context()
	if err != nil {
		return nil, err
	}
	data, _ := io.ReadAll(inStream)
	p, n := cbuf(data)
	var st C.int
	h := C.xlz_new_reader2(c, p, n, C.int(dictSize), &st)
	if h == nil {
		return nil, statusToError(st, true)
	}
	return &reader{h: h}, nil
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
	c, err := context()
	if err != nil {
		return nil, err
	}
	data, _ := io.ReadAll(readers[0])
	dp, dn := cbuf(data)
	pp, pn := cbuf(props)
	ptrs := (**C.uint8_t)(C.malloc(C.size_t(unsafe.Sizeof(dp))))
	lens := (*C.size_t)(C.malloc(C.size_t(unsafe.Sizeof(dn))))
	defer C.free(unsafe.Pointer(ptrs))
	defer C.free(unsafe.Pointer(lens))
	*ptrs, *lens = dp, dn
	var st C.int
	var h *C.xlz_reader
	if lzma2 {
		h = C.xlz_new_lzma2_decompressor_for_sevenzip(c, pp, pn, C.uint64_t(unpackSize), ptrs, lens, 1, &st)
	} else {
		h = C.xlz_new_lzma_decompressor_for_sevenzip(c, pp, pn, C.uint64_t(unpackSize), ptrs, lens, 1, &st)
	}
	if h == nil {
		return nil, statusToError(st, true)
	}
	return &reader{h: h, closer: readers[0], wrap: true}, nil
}

// DecodeBatch is the new entry the reference has no analogue for: n independent .lzma
// streams decoded concurrently on the GPU (one wave per stream).  outs[i] must have the
// capacity of stream i's decoded size.
func DecodeBatch(streams [][]byte, outs [][]byte) ([]int, []error, error) {
	c, err := context()
	if err != nil {
		return nil, nil, err
	}
	n := len(streams)
	descs := make([]C.xlz_stream_desc, n)
	results := make([]C.xlz_result, n)
	for i := range streams {
		descs[i].in, descs[i].in_len = cbuf(streams[i])
		descs[i].out, descs[i].out_cap = cbuf(outs[i][:cap(outs[i])])
		descs[i].format = C.XLZ_FMT_LZMA_ALONE
	}
	// the desc array holds Go pointers: pin them for the duration of the call (runtime.Pinner, Go >= 1.21)
	if st := C.xlz_decode_batch(c, &descs[0], C.size_t(n), &results[0]); st != C.XLZ_OK {
		return nil, nil, ErrDevice
	}
	lens := make([]int, n)
	errs := make([]error, n)
	for i := range results {
		lens[i] = int(results[i].out_len)
		errs[i] = statusToError(results[i].status, false)
	}
	return lens, errs, nil
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
		return nil, statusToError(st, true)
	}
	out := make([]byte, int(total)+1)
	op, _ := cbuf(out)
	var outLen C.uint64_t
	v := C.int(0)
	if verify {
		v = 1
	}
	if st := C.xlz_xz_decode(c, fp, fl, op, C.size_t(total), &outLen, v, nil); st != C.XLZ_OK {
		return nil, statusToError(st, false)
	}
	return out[:int(outLen)], nil
}
