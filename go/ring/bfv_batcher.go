package ring

// #include <stdlib.h>
// #include "lattigo_ring.h"
import "C"

import (
	"runtime"
	"sync"
	"unsafe"
)

// BfvBatcher merges the Mul and Relinearize calls of the BFV evaluators of many goroutines into batched device launches
// (lr_bfv_batcher_* in lattigo_ring.h).  This is the workload upstream itself hands to a pool of goroutines: every task of
// examples/dbfv/psi/psi.go:215-233 calls evaluator.Mul and evaluator.Relinearize on one ciphertext pair.  One batcher per parameter set,
// shared by the evaluators; each lane is a BfvPlan over (contextQ, contextQMul) and a key-switch plan over (contextQ, contextP).
type BfvBatcher struct {
	Q, P, QMul []uint64
	N, T       uint64
	muls       []*BfvPlan
	kss        []*CkksPlan
	h          *C.lr_bfv_batcher
	mu         sync.Mutex
	images     map[*Poly]*Poly // key image per SwitchingKey: ONE handle for all evaluators, so that their calls share batches
}

func NewBfvBatcher(N uint64, Q, P, QMul []uint64, t uint64, maxBatch, lanes int) *BfvBatcher {
	b := &BfvBatcher{N: N, T: t, Q: Q, P: P, QMul: QMul, images: map[*Poly]*Poly{}}
	size := C.size_t(lanes) * C.size_t(unsafe.Sizeof(uintptr(0)))
	rawM, rawK := C.malloc(size), C.malloc(size)
	defer C.free(rawM)
	defer C.free(rawK)
	am := (*[1 << 20]*C.lr_bfv_plan)(rawM)[:lanes:lanes]
	ak := (*[1 << 20]*C.lr_ckks_plan)(rawK)[:lanes:lanes]
	for i := 0; i < lanes; i++ {
		cq, err := NewContextWithParams(N, Q)
		if err != nil {
			panic(err)
		}
		cp, err := NewContextWithParams(N, P)
		if err != nil {
			panic(err)
		}
		cm, err := NewContextWithParams(N, QMul)
		if err != nil {
			panic(err)
		}
		mul, ks := NewBfvPlan(cq, cm, t, maxBatch), NewCkksPlan(cq, cp, maxBatch)
		b.muls, b.kss = append(b.muls, mul), append(b.kss, ks)
		am[i], ak[i] = mul.h, ks.h
	}
	call(func() C.int {
		return C.lr_bfv_batcher_create((**C.lr_bfv_plan)(rawM), (**C.lr_ckks_plan)(rawK), C.int(lanes), &b.h)
	})
	runtime.SetFinalizer(b, func(b *BfvBatcher) { C.lr_bfv_batcher_destroy(b.h) })
	return b
}

// KeyImage: the device image of a relinearisation key, uploaded once and shared by every caller.
func (b *BfvBatcher) KeyImage(evakey [][2]*Poly) *Poly {
	b.mu.Lock()
	defer b.mu.Unlock()
	if img, ok := b.images[evakey[0][0]]; ok {
		return img
	}
	img := b.kss[0].SwitchingKeyImage(evakey)
	b.kss[0].contextQ.Sync()
	b.images[evakey[0][0]] = img
	return img
}

// Mul = evaluator.Mul (bfv/evaluator.go:467) of two degree-1 ciphertexts; callerQ: the calling evaluator's contextQ.  Blocks until this
// call's result is complete; safe from any number of goroutines.
func (b *BfvBatcher) Mul(callerQ *Context, ct0, ct1 [2]*Poly, ctOut [3]*Poly) {
	callerQ.use(ct0[0], ct0[1], ct1[0], ct1[1])
	callerQ.want(ctOut[0], ctOut[1], ctOut[2])
	call(func() C.int {
		return C.lr_bfv_batcher_mul(b.h, ct0[0].d, ct0[1].d, ct1[0].d, ct1[1].d, ctOut[0].d, ctOut[1].d, ctOut[2].d)
	})
	done(ctOut[0], ctOut[1], ctOut[2])
}

// Relinearize = evaluator.Relinearize (bfv/evaluator.go:512) of a degree-2 ciphertext with the shared key image.
func (b *BfvBatcher) Relinearize(callerQ *Context, ct [3]*Poly, evakey *Poly, ctOut [2]*Poly) {
	callerQ.use(ct[0], ct[1], ct[2])
	callerQ.want(ctOut[0], ctOut[1])
	call(func() C.int {
		return C.lr_bfv_batcher_relinearize(b.h, ct[0].d, ct[1].d, ct[2].d, evakey.d, ctOut[0].d, ctOut[1].d)
	})
	done(ctOut[0], ctOut[1])
}

// Stats: launches so far, polys they carried, the largest batch.
func (b *BfvBatcher) Stats() (batches, products uint64, largest int) {
	var nb, np C.uint64_t
	var l C.int
	call(func() C.int { return C.lr_bfv_batcher_stats(b.h, &nb, &np, &l) })
	return uint64(nb), uint64(np), int(l)
}
