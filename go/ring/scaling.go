package ring

// #include "lattigo_ring.h"
import "C"

import "runtime"

// The rescale family of ring/ring_scaling.go:9-164.  Like the reference they shorten the polynomial:
// p0.Coeffs = p0.Coeffs[:level] (:33,54,113,149); the device image keeps its stride and lowers its limb count.
func (c *Context) rescale(p0 *Poly, f func() C.int) {
	in(p0)
	check(f())
	p0.Coeffs = p0.Coeffs[:len(p0.Coeffs)-1]
	out(p0)
}

func (c *Context) DivFloorByLastModulusNTT(p0 *Poly) { // :9
	c.rescale(p0, func() C.int { return C.lr_div_floor_by_last_modulus_ntt(c.h, p0.d) })
}
func (c *Context) DivFloorByLastModulus(p0 *Poly) { // :37
	c.rescale(p0, func() C.int { return C.lr_div_floor_by_last_modulus(c.h, p0.d) })
}
func (c *Context) DivRoundByLastModulusNTT(p0 *Poly) { // :72
	c.rescale(p0, func() C.int { return C.lr_div_round_by_last_modulus_ntt(c.h, p0.d) })
}
func (c *Context) DivRoundByLastModulus(p0 *Poly) { // :117
	c.rescale(p0, func() C.int { return C.lr_div_round_by_last_modulus(c.h, p0.d) })
}

func (c *Context) many(p0 *Poly, nb uint64, ntt, round bool) {
	in(p0)
	domain := C.int(0)
	if ntt {
		domain = 1
	}
	if round {
		check(C.lr_div_round_by_last_modulus_many(c.h, p0.d, C.int(nb), domain))
	} else {
		check(C.lr_div_floor_by_last_modulus_many(c.h, p0.d, C.int(nb), domain))
	}
	p0.Coeffs = p0.Coeffs[:uint64(len(p0.Coeffs))-nb]
	out(p0)
}

func (c *Context) DivFloorByLastModulusManyNTT(p0 *Poly, nbRescales uint64) { c.many(p0, nbRescales, true, false) }  // :58
func (c *Context) DivFloorByLastModulusMany(p0 *Poly, nbRescales uint64)    { c.many(p0, nbRescales, false, false) } // :65
func (c *Context) DivRoundByLastModulusManyNTT(p0 *Poly, nbRescales uint64) { c.many(p0, nbRescales, true, true) }   // :153
func (c *Context) DivRoundByLastModulusMany(p0 *Poly, nbRescales uint64)    { c.many(p0, nbRescales, false, true) }  // :160

// SimpleScaler mirrors ring/ring_scaling.go:168-181: the tables (wi, ti) live behind the handle.
type SimpleScaler struct {
	context *Context
	t       uint64
	h       *C.lr_simple_scaler
}

func NewSimpleScaler(t uint64, context *Context) *SimpleScaler { // :186
	s := &SimpleScaler{context: context, t: t}
	check(C.lr_simple_scaler_create(context.h, C.uint64_t(t), &s.h))
	runtime.SetFinalizer(s, func(s *SimpleScaler) { C.lr_simple_scaler_destroy(s.h) })
	return s
}

// Scale returns the reconstruction of p1 scaled by t/Q modulo t on every limb of p2 (:275)
func (s *SimpleScaler) Scale(p1, p2 *Poly) {
	in(p1)
	check(C.lr_simple_scale(s.h, p1.d, p2.d))
	out(p2)
}
