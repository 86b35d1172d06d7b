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
	check(C.lr_bext_create(contextQ.h, contextP.h, &b.h))
	runtime.SetFinalizer(b, func(b *FastBasisExtender) { C.lr_bext_destroy(b.h) })
	return b
}

func (b *FastBasisExtender) ModUpSplitQP(level uint64, p1, p2 *Poly) { // :147
	in(p1)
	check(C.lr_modup_split_qp(b.h, C.int(level), p1.d, p2.d))
	out(p2)
}

func (b *FastBasisExtender) ModUpSplitPQ(level uint64, p1, p2 *Poly) { // :154
	in(p1)
	check(C.lr_modup_split_pq(b.h, C.int(level), p1.d, p2.d))
	out(p2)
}

func (b *FastBasisExtender) ModDownNTTPQ(level uint64, p1, p2 *Poly) { // :163
	in(p1)
	check(C.lr_moddown_ntt_pq(b.h, C.int(level), p1.d, p2.d))
	out(p1, p2)
}

func (b *FastBasisExtender) ModDownSplitedNTTPQ(level uint64, p1Q, p1P, p2 *Poly) { // :207
	in(p1Q, p1P)
	check(C.lr_moddown_split_ntt_pq(b.h, C.int(level), p1Q.d, p1P.d, p2.d))
	out(p1P, p2)
}

func (b *FastBasisExtender) ModDownPQ(level uint64, p1, p2 *Poly) { // :248
	in(p1)
	check(C.lr_moddown_pq(b.h, C.int(level), p1.d, p2.d))
	out(p2)
}

func (b *FastBasisExtender) ModDownSplitedPQ(level uint64, p1Q, p1P, p2 *Poly) { // :281
	in(p1Q, p1P)
	check(C.lr_moddown_split_pq(b.h, C.int(level), p1Q.d, p1P.d, p2.d))
	out(p2)
}

func (b *FastBasisExtender) ModDownSplitedQP(levelQ, levelP uint64, p1Q, p1P, p2 *Poly) { // :314
	in(p1Q, p1P)
	check(C.lr_moddown_split_qp(b.h, C.int(levelQ), C.int(levelP), p1Q.d, p1P.d, p2.d))
	out(p2)
}

// Decomposer mirrors ring/ring_basis_extension.go:398-472.  The reference constructor takes the two modulus
// lists; the shim needs the contexts (they carry the lists and the device).
type Decomposer struct {
	h *C.lr_decomposer
}

func NewDecomposer(contextQ, contextP *Context) *Decomposer { // :415
	d := &Decomposer{}
	check(C.lr_decomposer_create(contextQ.h, contextP.h, &d.h))
	runtime.SetFinalizer(d, func(d *Decomposer) { C.lr_decomposer_destroy(d.h) })
	return d
}

func (d *Decomposer) Decompose(level, crtDecompLevel uint64, p0, p1 *Poly) { // :476
	in(p0)
	check(C.lr_decompose(d.h, C.int(level), C.int(crtDecompLevel), p0.d, p1.d))
	out(p1)
}

func (d *Decomposer) DecomposeAndSplit(level, crtDecompLevel uint64, p0, p1Q, p1P *Poly) { // :601
	in(p0)
	check(C.lr_decompose_and_split(d.h, C.int(level), C.int(crtDecompLevel), p0.d, p1Q.d, p1P.d))
	out(p1Q, p1P)
}
