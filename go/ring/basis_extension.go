package ring

// #include "lattigo_ring.h"
import "C"

import "runtime"

// FastBasisExtender mirrors ring/ring_basis_extension.go:9-18; tables and scratch live on the device.
type FastBasisExtender struct {
	contextQ, contextP *Context
	h                  *C.lr_bext
}

func NewFastBasisExtender(contextQ, contextP *Context) *FastBasisExtender { // :57
	b := &FastBasisExtender{contextQ: contextQ, contextP: contextP}
	call(func() C.int { return C.lr_bext_create(contextQ.h, contextP.h, &b.h) })
	runtime.SetFinalizer(b, func(b *FastBasisExtender) { C.lr_bext_destroy(b.h) })
	return b
}

func (b *FastBasisExtender) ModUpSplitQP(level uint64, p1, p2 *Poly) { // :147
	b.contextQ.use(p1)
	b.contextP.want(p2)
	call(func() C.int { return C.lr_modup_split_qp(b.h, C.int(level), p1.d, p2.d) })
	done(p2)
}

func (b *FastBasisExtender) ModUpSplitPQ(level uint64, p1, p2 *Poly) { // :154
	b.contextP.use(p1)
	b.contextQ.want(p2)
	call(func() C.int { return C.lr_modup_split_pq(b.h, C.int(level), p1.d, p2.d) })
	done(p2)
}

func (b *FastBasisExtender) ModDownNTTPQ(level uint64, p1, p2 *Poly) { // :163
	b.contextQ.use(p1)
	b.contextQ.want(p2)
	call(func() C.int { return C.lr_moddown_ntt_pq(b.h, C.int(level), p1.d, p2.d) })
	done(p1, p2) // the P limbs of p1 come back in the coefficient domain, as in the reference (:172-174)
}

func (b *FastBasisExtender) ModDownSplitedNTTPQ(level uint64, p1Q, p1P, p2 *Poly) { // :207
	b.contextQ.use(p1Q)
	b.contextP.use(p1P)
	b.contextQ.want(p2)
	call(func() C.int { return C.lr_moddown_split_ntt_pq(b.h, C.int(level), p1Q.d, p1P.d, p2.d) })
	done(p1P, p2)
}

func (b *FastBasisExtender) ModDownPQ(level uint64, p1, p2 *Poly) { // :248
	b.contextQ.use(p1)
	b.contextQ.want(p2)
	call(func() C.int { return C.lr_moddown_pq(b.h, C.int(level), p1.d, p2.d) })
	done(p2)
}

func (b *FastBasisExtender) ModDownSplitedPQ(level uint64, p1Q, p1P, p2 *Poly) { // :281
	b.contextQ.use(p1Q)
	b.contextP.use(p1P)
	b.contextQ.want(p2)
	call(func() C.int { return C.lr_moddown_split_pq(b.h, C.int(level), p1Q.d, p1P.d, p2.d) })
	done(p2)
}

func (b *FastBasisExtender) ModDownSplitedQP(levelQ, levelP uint64, p1Q, p1P, p2 *Poly) { // :314
	b.contextQ.use(p1Q)
	b.contextP.use(p1P)
	b.contextP.want(p2)
	call(func() C.int { return C.lr_moddown_split_qp(b.h, C.int(levelQ), C.int(levelP), p1Q.d, p1P.d, p2.d) })
	done(p2)
}

// Decomposer mirrors ring/ring_basis_extension.go:398-472.  The reference constructor takes the two modulus lists and no
// degree -- its tables do not depend on N -- so the device handle (which lives on a device and launches kernels of a
// given degree) is created at the first Decompose / DecomposeAndSplit call, from the degree of the polynomial it is given.
type Decomposer struct {
	Q, P             []uint64
	nQprimes         uint64
	nPprimes         uint64
	alpha, beta      uint64
	xalpha           []uint64
	contextQ         *Context
	contextP         *Context
	h                *C.lr_decomposer
}

func NewDecomposer(Q, P []uint64) *Decomposer { // :415
	d := &Decomposer{Q: append([]uint64{}, Q...), P: append([]uint64{}, P...)}
	d.nQprimes, d.nPprimes = uint64(len(Q)), uint64(len(P))
	d.alpha = d.nPprimes
	d.beta = (d.nQprimes + d.alpha - 1) / d.alpha // ceil(len(Q) / alpha), :433
	d.xalpha = make([]uint64, d.beta)
	for i := range d.xalpha {
		d.xalpha[i] = d.alpha
	}
	if r := d.nQprimes % d.alpha; r != 0 {
		d.xalpha[d.beta-1] = r
	}
	return d
}

// Xalpha (:409): the number of moduli of each digit.
func (d *Decomposer) Xalpha() []uint64 { return d.xalpha }

func (d *Decomposer) handle(N uint64) {
	if d.h != nil && d.contextQ.N == N {
		return
	}
	var err error
	if d.contextQ, err = NewContextWithParams(N, d.Q); err != nil {
		panic(err)
	}
	if d.contextP, err = NewContextWithParams(N, d.P); err != nil {
		panic(err)
	}
	if d.h != nil {
		C.lr_decomposer_destroy(d.h)
		d.h = nil
	}
	call(func() C.int { return C.lr_decomposer_create(d.contextQ.h, d.contextP.h, &d.h) })
	runtime.SetFinalizer(d, func(d *Decomposer) { C.lr_decomposer_destroy(d.h) })
}

func (d *Decomposer) Decompose(level, crtDecompLevel uint64, p0, p1 *Poly) { // :476
	d.handle(uint64(len(p0.Coeffs[0])))
	d.contextQ.use(p0)
	d.contextQ.want(p1)
	call(func() C.int { return C.lr_decompose(d.h, C.int(level), C.int(crtDecompLevel), p0.d, p1.d) })
	done(p1)
}

func (d *Decomposer) DecomposeAndSplit(level, crtDecompLevel uint64, p0, p1Q, p1P *Poly) { // :601
	d.handle(uint64(len(p0.Coeffs[0])))
	d.contextQ.use(p0)
	d.contextQ.want(p1Q)
	d.contextP.want(p1P)
	call(func() C.int { return C.lr_decompose_and_split(d.h, C.int(level), C.int(crtDecompLevel), p0.d, p1Q.d, p1P.d) })
	done(p1Q, p1P)
}
