package ring

// #include "lattigo_ring.h"
import "C"

import (
	"errors"
	"runtime"
	"unsafe"
)

// Context mirrors ring.Context (ring/ring_context.go:18-51).  The tables live on the device; the getters
// return the reference's host-format copies (Montgomery form, bit-reversed order).
type Context struct {
	N       uint64
	Modulus []uint64

	h      *C.lr_context
	device int
}

// NewContext + SetParameters + GenNTTParams in one call (ring/ring_context.go:54-66).
func NewContextWithParams(N uint64, Moduli []uint64) (*Context, error) {
	return NewContextWithParamsOnDevice(N, Moduli, 0)
}

// NewContextWithParamsOnDevice binds the context (and everything created from it) to one GPU.
func NewContextWithParamsOnDevice(N uint64, Moduli []uint64, device int) (*Context, error) {
	c := &Context{N: N, Modulus: append([]uint64{}, Moduli...), device: device}
	rc := C.lr_context_create(C.uint64_t(N), (*C.uint64_t)(unsafe.Pointer(&c.Modulus[0])), C.int(len(Moduli)), C.int(device), &c.h)
	switch rc {
	case C.LR_OK:
	case C.LR_ERR_NOT_NTT_FRIENDLY:
		return c, errors.New("warning : provided modulus does not allow NTT") // ring/ring_context.go:141-146
	case C.LR_ERR_INVALID_DEGREE:
		panic("invalid ring degree (must be a power of 2)") // :71-73
	default:
		return nil, statusErr(rc)
	}
	runtime.SetFinalizer(c, func(c *Context) { C.lr_context_destroy(c.h) })
	return c, nil
}

func (c *Context) table(which C.int, n int) []uint64 {
	out := make([]uint64, n)
	check(C.lr_context_get_table(c.h, which, (*C.uint64_t)(unsafe.Pointer(&out[0])), C.size_t(n)))
	return out
}

func (c *Context) rows(flat []uint64, per int) [][]uint64 {
	out := make([][]uint64, len(c.Modulus))
	for i := range out {
		out[i] = flat[i*per : (i+1)*per]
	}
	return out
}

// GetNttPsi etc. (ring/ring_context.go:253-285).
func (c *Context) GetNttPsi() [][]uint64    { return c.rows(c.table(C.LR_TAB_NTT_PSI, len(c.Modulus)*int(c.N)), int(c.N)) }
func (c *Context) GetNttPsiInv() [][]uint64 { return c.rows(c.table(C.LR_TAB_NTT_PSI_INV, len(c.Modulus)*int(c.N)), int(c.N)) }
func (c *Context) GetNttNInv() []uint64     { return c.table(C.LR_TAB_NTT_N_INV, len(c.Modulus)) }
func (c *Context) GetMredParams() []uint64  { return c.table(C.LR_TAB_MRED, len(c.Modulus)) }
func (c *Context) GetBredParams() [][]uint64 {
	return c.rows(c.table(C.LR_TAB_BRED, 2*len(c.Modulus)), 2)
}

// NewPoly / NewPolyLvl (ring/ring_context.go:288,300).
func (c *Context) NewPoly() *Poly { return c.NewPolyLvl(uint64(len(c.Modulus) - 1)) }

func (c *Context) NewPolyLvl(level uint64) *Poly {
	p := &Poly{Coeffs: make([][]uint64, level+1)}
	for i := range p.Coeffs {
		p.Coeffs[i] = make([]uint64, c.N)
	}
	check(C.lr_poly_alloc(c.h, C.int(level+1), 1, &p.d))
	runtime.SetFinalizer(p, func(p *Poly) { C.lr_poly_free(p.d) })
	return p
}

// Sync waits for the device work queued by this context's handles.
func (c *Context) Sync() { check(C.lr_context_sync(c.h)) }
