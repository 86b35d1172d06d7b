package ring

// #include "lattigo_ring.h"
import "C"

func (c *Context) full() C.int { return C.int(len(c.Modulus) - 1) }

// NTT / NTTLvl / InvNTT / InvNTTLvl (ring/ntt.go:4-29).  p1 == p2 is allowed, as in the reference.
func (c *Context) NTT(p1, p2 *Poly) {
	in(p1)
	check(C.lr_ntt(c.h, c.full(), p1.d, p2.d))
	out(p2)
}

func (c *Context) NTTLvl(level uint64, p1, p2 *Poly) {
	in(p1)
	check(C.lr_ntt(c.h, C.int(level), p1.d, p2.d))
	out(p2)
}

func (c *Context) InvNTT(p1, p2 *Poly) {
	in(p1)
	check(C.lr_intt(c.h, c.full(), p1.d, p2.d))
	out(p2)
}

func (c *Context) InvNTTLvl(level uint64, p1, p2 *Poly) {
	in(p1)
	check(C.lr_intt(c.h, C.int(level), p1.d, p2.d))
	out(p2)
}
