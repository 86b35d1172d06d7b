package ring

// #include <stdlib.h>
// #include "lattigo_ring.h"
import "C"

import (
	"sync"
	"unsafe"
)

func (c *Context) top() uint64 { return uint64(len(c.Modulus) - 1) }

// NTT / NTTLvl / InvNTT / InvNTTLvl (ring/ntt.go:4-29).  p1 == p2 is allowed, as in the reference.
func (c *Context) NTT(p1, p2 *Poly) { c.NTTLvl(c.top(), p1, p2) }

func (c *Context) NTTLvl(level uint64, p1, p2 *Poly) {
	c.use(p1)
	c.want(p2)
	call(func() C.int { return C.lr_ntt(c.h, C.int(level), p1.d, p2.d) })
	done(p2)
}

func (c *Context) InvNTT(p1, p2 *Poly) { c.InvNTTLvl(c.top(), p1, p2) }

func (c *Context) InvNTTLvl(level uint64, p1, p2 *Poly) {
	c.use(p1)
	c.want(p2)
	call(func() C.int { return C.lr_intt(c.h, C.int(level), p1.d, p2.d) })
	done(p2)
}

// The package-level NTT / InvNTT of the reference (ring/ntt.go:53,89) take one limb as a host slice together with its
// table and constants; the evaluators call them to transform a limb under a given modulus (ckks/evaluator.go:1586,
// bfv/evaluator.go:766).  The table argument is redundant with (N, Q) -- GenNTTParams derives it deterministically -- so
// the call is served by a cached one-modulus context and lr_ntt_host_limb (upload, kernel, download).
var (
	limbMu  sync.Mutex
	limbCtx = map[[2]uint64]*Context{}
)

func limbContext(N, Q uint64) *Context {
	limbMu.Lock()
	defer limbMu.Unlock()
	key := [2]uint64{N, Q}
	if c, ok := limbCtx[key]; ok {
		return c
	}
	c, err := NewContextWithParams(N, []uint64{Q})
	if err != nil {
		panic(err)
	}
	limbCtx[key] = c
	return c
}

func limbTransform(coeffsIn, coeffsOut []uint64, N, Q uint64, inverse bool) {
	c := limbContext(N, Q)
	inv := C.int(0)
	if inverse {
		inv = 1
	}
	in := (*C.uint64_t)(unsafe.Pointer(&coeffsIn[0]))
	out := (*C.uint64_t)(unsafe.Pointer(&coeffsOut[0]))
	call(func() C.int { return C.lr_ntt_host_limb(c.h, 0, inv, in, out) })
}

// NTT (ring/ntt.go:53): forward transform of one limb.  nttPsi, mredParams and bredParams are accepted for source
// compatibility; the library's own tables for (N, Q) are the same values.
func NTT(coeffsIn, coeffsOut []uint64, N uint64, nttPsi []uint64, Q, mredParams uint64, bredParams []uint64) {
	limbTransform(coeffsIn, coeffsOut, N, Q, false)
}

// InvNTT (ring/ntt.go:89).
func InvNTT(coeffsIn, coeffsOut []uint64, N uint64, nttPsiInv []uint64, nttNInv, Q, mredParams uint64) {
	limbTransform(coeffsIn, coeffsOut, N, Q, true)
}

// Butterfly / InvButterfly (ring/ntt.go:32,43): the reference's scalar butterflies, for callers that build their own loops.
func Butterfly(U, V, Psi, Q, Qinv uint64) (X, Y uint64) {
	if U > 2*Q {
		U -= 2 * Q
	}
	V = MRedConstant(V, Psi, Q, Qinv)
	return U + V, U + 2*Q - V
}

func InvButterfly(U, V, Psi, Q, Qinv uint64) (X, Y uint64) {
	X = U + V
	if X > 2*Q {
		X -= 2 * Q
	}
	Y = MRedConstant(U+2*Q-V, Psi, Q, Qinv)
	return
}
