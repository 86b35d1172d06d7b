package ring

// #include "lattigo_ring.h"
import "C"

import "runtime"

// The rescale family of ring/ring_scaling.go:9-164.  Like the reference they shorten the polynomial:
// p0.Coeffs = p0.Coeffs[:level] (:33,54,113,149); the device image keeps its stride and lowers its limb count.
func (c *Context) rescale(p0 *Poly, drop uint64, f func() C.int) {
	c.use(p0)
	call(f)
	p0.Coeffs = p0.Coeffs[:uint64(len(p0.Coeffs))-drop]
	done(p0)
}

func (c *Context) DivFloorByLastModulusNTT(p0 *Poly) { // :9
	c.rescale(p0, 1, func() C.int { return C.lr_div_floor_by_last_modulus_ntt(c.h, p0.d) })
}
func (c *Context) DivFloorByLastModulus(p0 *Poly) { // :37
	c.rescale(p0, 1, func() C.int { return C.lr_div_floor_by_last_modulus(c.h, p0.d) })
}
func (c *Context) DivRoundByLastModulusNTT(p0 *Poly) { // :72
	c.rescale(p0, 1, func() C.int { return C.lr_div_round_by_last_modulus_ntt(c.h, p0.d) })
}
func (c *Context) DivRoundByLastModulus(p0 *Poly) { // :117
	c.rescale(p0, 1, func() C.int { return C.lr_div_round_by_last_modulus(c.h, p0.d) })
}
func (c *Context) DivFloorByLastModulusManyNTT(p0 *Poly, nbRescales uint64) { // :58
	c.rescale(p0, nbRescales, func() C.int { return C.lr_div_floor_by_last_modulus_many(c.h, p0.d, C.int(nbRescales), 1) })
}
func (c *Context) DivFloorByLastModulusMany(p0 *Poly, nbRescales uint64) { // :65
	c.rescale(p0, nbRescales, func() C.int { return C.lr_div_floor_by_last_modulus_many(c.h, p0.d, C.int(nbRescales), 0) })
}
func (c *Context) DivRoundByLastModulusManyNTT(p0 *Poly, nbRescales uint64) { // :153
	c.rescale(p0, nbRescales, func() C.int { return C.lr_div_round_by_last_modulus_many(c.h, p0.d, C.int(nbRescales), 1) })
}
func (c *Context) DivRoundByLastModulusMany(p0 *Poly, nbRescales uint64) { // :160
	c.rescale(p0, nbRescales, func() C.int { return C.lr_div_round_by_last_modulus_many(c.h, p0.d, C.int(nbRescales), 0) })
}

// SimpleScaler (ring/ring_scaling.go:168-300): round(t/Q * x) mod t, tables in double-double on the host, Scale on the device.
type SimpleScaler struct {
	context *Context
	t       uint64
	h       *C.lr_simple_scaler
}

func NewSimpleScaler(t uint64, context *Context) *SimpleScaler { // :186
	s := &SimpleScaler{context: context, t: t}
	call(func() C.int { return C.lr_simple_scaler_create(context.h, C.uint64_t(t), &s.h) })
	runtime.SetFinalizer(s, func(s *SimpleScaler) { C.lr_simple_scaler_destroy(s.h) })
	return s
}

// Scale (:275): p2 may belong to another context of the same degree (bfv's contextT, bfv/encoder.go:142).
func (s *SimpleScaler) Scale(p1, p2 *Poly) {
	s.context.use(p1)
	s.context.want(p2)
	call(func() C.int { return C.lr_simple_scale(s.h, p1.d, p2.d) })
	done(p2)
}
