package ring

// #include <stdlib.h>
// #include "lattigo_ring.h"
import "C"

import (
	"runtime"
	"sync"
	"unsafe"
)

// CkksPlan owns what ckks.NewEvaluator builds around the ring (ckks/evaluator.go:63-112: base converter,
// decomposer, scratch pools) on the device, and runs the evaluator's ring call sequences as one cgo call each.
// An evaluator built on this shim replaces the bodies of MulRelin / Rescale / RotateColumns / RotateHoisted by
// these calls; everything else (scale bookkeeping, level checks, panics) stays Go.
type CkksPlan struct {
	contextQ, contextP *Context
	h                  *C.lr_ckks_plan
}

func NewCkksPlan(contextQ, contextP *Context, maxBatch int) *CkksPlan {
	p := &CkksPlan{contextQ: contextQ, contextP: contextP}
	call(func() C.int { return C.lr_ckks_plan_create_ex(contextQ.h, contextP.h, C.int(maxBatch), DefaultOptions.ptr(), &p.h) })
	runtime.SetFinalizer(p, func(p *CkksPlan) { C.lr_ckks_plan_destroy(p.h) })
	return p
}

// Stats: diagnostics of the plan's small-batch paths (lr_ckks_plan_stats): forks, grouped digit extensions.
func (p *CkksPlan) Stats() (forks, groupedExtensions uint64) {
	var f, g C.uint64_t
	call(func() C.int { return C.lr_ckks_plan_stats(p.h, &f, &g) })
	return uint64(f), uint64(g)
}

// SwitchingKeyImage lays SwitchingKey.evakey ([beta][2]*ring.Poly over QP, ckks/keygen.go:68-70) out as one device
// polynomial with batch = 2*beta, which is what the key-switching entry points take.
func (p *CkksPlan) SwitchingKeyImage(evakey [][2]*Poly) *Poly {
	limbs := len(evakey[0][0].Coeffs)
	img := &Poly{resident: true, dLimbs: limbs}
	call(func() C.int { return C.lr_poly_alloc(p.contextQ.h, C.int(limbs), C.int(2*len(evakey)), &img.d) }) // limbs beyond |Q| follow contextP: see lr_ckks_switch_keys
	for i := range evakey {
		for k := 0; k < 2; k++ {
			src := evakey[i][k]
			src.hostView()
			src.uploadTo(img.d, 2*i+k)
		}
	}
	runtime.SetFinalizer(img, func(q *Poly) { C.lr_poly_free(q.d) })
	return img
}

// SwitchKeysInPlace = evaluator.switchKeysInPlace (ckks/evaluator.go:1475).
func (p *CkksPlan) SwitchKeysInPlace(level uint64, cx, evakey, p0, p1 *Poly) {
	p.contextQ.use(cx)
	p.contextQ.want(p0, p1)
	call(func() C.int { return C.lr_ckks_switch_keys(p.h, C.int(level), cx.d, evakey.d, p0.d, p1.d) })
	done(p0, p1)
}

// MulRelin = evaluator.MulRelin (ckks/evaluator.go:1016-1133) on the ring level.  ct0 / ct1: value slices of the two operands
// (two polynomials for a ciphertext, one for a plaintext); evakey: the key image, or nil for the degree-2 result
// (:1061-1066), in which case ctOut holds three polynomials.
func (p *CkksPlan) MulRelin(level uint64, ct0, ct1 []*Poly, evakey *Poly, ctOut []*Poly) {
	q := p.contextQ
	q.use(ct0...)
	q.use(ct1...)
	q.want(ctOut...)
	switch {
	case len(ct0)+len(ct1) == 3: // plaintext x ciphertext, :1113-1131
		pt, ct := ct0, ct1
		if len(ct0) == 2 {
			pt, ct = ct1, ct0
		}
		call(func() C.int { return C.lr_ckks_mul_plain(p.h, C.int(level), pt[0].d, ct[0].d, ct[1].d, ctOut[0].d, ctOut[1].d) })
	case evakey == nil:
		call(func() C.int {
			return C.lr_ckks_mul_norelin(p.h, C.int(level), ct0[0].d, ct0[1].d, ct1[0].d, ct1[1].d, ctOut[0].d, ctOut[1].d, ctOut[2].d)
		})
	default:
		call(func() C.int {
			return C.lr_ckks_mulrelin(p.h, C.int(level), ct0[0].d, ct0[1].d, ct1[0].d, ct1[1].d, evakey.d, ctOut[0].d, ctOut[1].d)
		})
	}
	done(ctOut...)
}

// EncryptPk = pkEncryptor.encrypt, the branch through the special primes, after the sampling (ckks/encryptor.go:205-234).
// u, pk, e: polynomials of contextQP (|Q|+|P| limbs); the sampling itself stays in the reference's Go code.
func (p *CkksPlan) EncryptPk(level uint64, u *Poly, pk, e [2]*Poly, plaintext *Poly, ctOut [2]*Poly) {
	q := p.contextQ
	q.use(u, pk[0], pk[1], e[0], e[1], plaintext)
	q.want(ctOut[0], ctOut[1])
	call(func() C.int {
		return C.lr_ckks_encrypt_pk(p.h, C.int(level), u.d, pk[0].d, pk[1].d, e[0].d, e[1].d, plaintext.d, ctOut[0].d, ctOut[1].d)
	})
	done(ctOut[0], ctOut[1])
}

// Decrypt = decryptor.Decrypt (ckks/decryptor.go:53-78).
func (p *CkksPlan) Decrypt(level uint64, ct []*Poly, sk, ptOut *Poly) {
	q := p.contextQ
	q.use(ct...)
	q.use(sk)
	q.want(ptOut)
	n := len(ct)
	raw := C.malloc(C.size_t(n) * C.size_t(unsafe.Sizeof(uintptr(0))))
	defer C.free(raw)
	arr := polyArray(raw, n)
	for i := range ct {
		arr[i] = ct[i].d
	}
	call(func() C.int { return C.lr_ckks_decrypt(p.h, C.int(level), (**C.lr_poly)(raw), C.int(n-1), sk.d, ptOut.d) })
	done(ptOut)
}

// Rescale = one iteration of evaluator.Rescale's loop (ckks/evaluator.go:958-960) on both components.
func (p *CkksPlan) Rescale(ct [2]*Poly) {
	p.contextQ.use(ct[0], ct[1])
	call(func() C.int { return C.lr_ckks_rescale(p.h, ct[0].d, ct[1].d) })
	for _, q := range ct {
		q.Coeffs = q.Coeffs[:len(q.Coeffs)-1]
	}
	done(ct[0], ct[1])
}

// PermuteNTT = evaluator.permuteNTT (ckks/evaluator.go:1448): RotateColumns with the key of that rotation, Conjugate.
func (p *CkksPlan) PermuteNTT(level uint64, ct0 [2]*Poly, galEl uint64, rotkey *Poly, ctOut [2]*Poly) {
	p.contextQ.use(ct0[0], ct0[1])
	p.contextQ.want(ctOut[0], ctOut[1])
	call(func() C.int {
		return C.lr_ckks_rotate(p.h, C.int(level), ct0[0].d, ct0[1].d, C.uint64_t(galEl), rotkey.d, ctOut[0].d, ctOut[1].d)
	})
	done(ctOut[0], ctOut[1])
}

// RotateHoisted = evaluator.RotateHoisted (ckks/evaluator.go:1252) for the rotations galEls[r] with keys rotkeys[r].
func (p *CkksPlan) RotateHoisted(level uint64, ct0 [2]*Poly, galEls []uint64, rotkeys []*Poly, ctOuts [][2]*Poly) {
	p.contextQ.use(ct0[0], ct0[1])
	n := len(galEls)
	for r := 0; r < n; r++ {
		p.contextQ.want(ctOuts[r][0], ctOuts[r][1])
	}
	sz := C.size_t(n) * C.size_t(unsafe.Sizeof(uintptr(0)))
	keys := (**C.lr_poly)(C.malloc(sz))
	o0 := (**C.lr_poly)(C.malloc(sz))
	o1 := (**C.lr_poly)(C.malloc(sz))
	defer C.free(unsafe.Pointer(keys))
	defer C.free(unsafe.Pointer(o0))
	defer C.free(unsafe.Pointer(o1))
	ks, a0, a1 := polyArray(unsafe.Pointer(keys), n), polyArray(unsafe.Pointer(o0), n), polyArray(unsafe.Pointer(o1), n)
	for r := 0; r < n; r++ {
		ks[r], a0[r], a1[r] = rotkeys[r].d, ctOuts[r][0].d, ctOuts[r][1].d
	}
	call(func() C.int {
		return C.lr_ckks_rotate_hoisted(p.h, C.int(level), ct0[0].d, ct0[1].d, C.int(n), (*C.uint64_t)(unsafe.Pointer(&galEls[0])), keys, o0, o1)
	})
	for r := 0; r < n; r++ {
		done(ctOuts[r][0], ctOuts[r][1])
	}
}

// polyArray views n handle slots of C memory as a Go slice (the pre-go-1.17 spelling of unsafe.Slice)
func polyArray(raw unsafe.Pointer, n int) []*C.lr_poly {
	return (*[1 << 28]*C.lr_poly)(raw)[:n:n]
}

// BfvSwitchKeys = bfv evaluator.switchKeys (bfv/evaluator.go:736-812) on the key-switch plan over (contextQ, contextP): the objects
// bfv.NewEvaluator builds for it (decomposer, baseconverterQ1P, keyswitchpool, :100-112) are the ones the CKKS plan holds.
// Coefficient domain in and out; evakey: SwitchingKeyImage of SwitchingKey.evakey.
func (p *CkksPlan) BfvSwitchKeys(cx, evakey, p0, p1 *Poly) {
	p.contextQ.use(cx)
	p.contextQ.want(p0, p1)
	call(func() C.int { return C.lr_bfv_switch_keys(p.h, cx.d, evakey.d, p0.d, p1.d) })
	done(p0, p1)
}

// BfvRelinearize = bfv evaluator.relinearize for a degree-2 ciphertext (bfv/evaluator.go:480-501): ctOut = (c0 + p0, c1 + p1).
func (p *CkksPlan) BfvRelinearize(ct [3]*Poly, evakey *Poly, ctOut [2]*Poly) {
	p.contextQ.use(ct[0], ct[1], ct[2])
	p.contextQ.want(ctOut[0], ctOut[1])
	call(func() C.int { return C.lr_bfv_relinearize(p.h, ct[0].d, ct[1].d, ct[2].d, evakey.d, ctOut[0].d, ctOut[1].d) })
	done(ctOut[0], ctOut[1])
}

// BfvPermute = bfv evaluator.permute (bfv/evaluator.go:711-735): Context.Permute of both components by the Galois element, switchKeys
// of the second with the rotation's key, Add + Copy -- the body of RotateRows (:670) and RotateColumns (:579); ctOut may be ct0.
func (p *CkksPlan) BfvPermute(ct0 [2]*Poly, generator uint64, switchKey *Poly, ctOut [2]*Poly) {
	p.contextQ.use(ct0[0], ct0[1])
	p.contextQ.want(ctOut[0], ctOut[1])
	call(func() C.int {
		return C.lr_bfv_rotate(p.h, ct0[0].d, ct0[1].d, C.uint64_t(generator), switchKey.d, ctOut[0].d, ctOut[1].d)
	})
	done(ctOut[0], ctOut[1])
}

// CkksBatcher merges the MulRelin calls of the evaluators of many goroutines -- upstream's concurrency model is one evaluator per
// goroutine, one ciphertext per call (examples/dbfv/psi/psi.go:215-233) -- into batched device launches (lr_ckks_batcher_* in
// lattigo_ring.h).  One batcher per parameter set, shared by the evaluators; each lane is a plan over its own pair of contexts.
type CkksBatcher struct {
	Q, P   []uint64
	N      uint64
	lanes  []*CkksPlan
	h      *C.lr_ckks_batcher
	mu     sync.Mutex
	images map[*Poly]*Poly // key image per SwitchingKey (its first poly identifies it): ONE handle for all evaluators, so that their calls share batches
}

func NewCkksBatcher(N uint64, Q, P []uint64, maxBatch, lanes int) *CkksBatcher {
	b := &CkksBatcher{N: N, Q: Q, P: P, images: map[*Poly]*Poly{}}
	raw := C.malloc(C.size_t(lanes) * C.size_t(unsafe.Sizeof(uintptr(0))))
	defer C.free(raw)
	arr := (*[1 << 20]*C.lr_ckks_plan)(raw)[:lanes:lanes]
	for i := 0; i < lanes; i++ {
		cq, err := NewContextWithParams(N, Q)
		if err != nil {
			panic(err)
		}
		cp, err := NewContextWithParams(N, P)
		if err != nil {
			panic(err)
		}
		plan := NewCkksPlan(cq, cp, maxBatch)
		b.lanes = append(b.lanes, plan)
		arr[i] = plan.h
	}
	call(func() C.int { return C.lr_ckks_batcher_create((**C.lr_ckks_plan)(raw), C.int(lanes), &b.h) })
	runtime.SetFinalizer(b, func(b *CkksBatcher) { C.lr_ckks_batcher_destroy(b.h) })
	return b
}

// KeyImage: the device image of a switching key, uploaded once and shared by every caller.
func (b *CkksBatcher) KeyImage(evakey [][2]*Poly) *Poly {
	b.mu.Lock()
	defer b.mu.Unlock()
	if img, ok := b.images[evakey[0][0]]; ok {
		return img
	}
	img := b.lanes[0].SwitchingKeyImage(evakey)
	b.lanes[0].contextQ.Sync()
	b.images[evakey[0][0]] = img
	return img
}

// MulRelin = evaluator.MulRelin (ckks/evaluator.go:1016) of two degree-1 ciphertexts with key; callerQ: the calling evaluator's
// contextQ (its polys are bound to it).  Blocks until this call's result is complete; safe from any number of goroutines.
func (b *CkksBatcher) MulRelin(callerQ *Context, level uint64, ct0, ct1 []*Poly, evakey *Poly, ctOut []*Poly) {
	callerQ.use(ct0...)
	callerQ.use(ct1...)
	callerQ.want(ctOut...)
	call(func() C.int {
		return C.lr_ckks_batcher_mulrelin(b.h, C.int(level), ct0[0].d, ct0[1].d, ct1[0].d, ct1[1].d, evakey.d, ctOut[0].d, ctOut[1].d)
	})
	done(ctOut...)
}

// PermuteNTT = evaluator.permuteNTT (ckks/evaluator.go:1448) through the batcher: calls with the same (level, galEl, key) in flight
// together run as one batched rotation.
func (b *CkksBatcher) PermuteNTT(callerQ *Context, level uint64, ct0 [2]*Poly, galEl uint64, rotkey *Poly, ctOut [2]*Poly) {
	callerQ.use(ct0[0], ct0[1])
	callerQ.want(ctOut[0], ctOut[1])
	call(func() C.int {
		return C.lr_ckks_batcher_rotate(b.h, C.int(level), ct0[0].d, ct0[1].d, C.uint64_t(galEl), rotkey.d, ctOut[0].d, ctOut[1].d)
	})
	done(ctOut[0], ctOut[1])
}

// Stats: launches so far, polys they carried, the largest batch.
func (b *CkksBatcher) Stats() (batches, products uint64, largest int) {
	var nb, np C.uint64_t
	var l C.int
	call(func() C.int { return C.lr_ckks_batcher_stats(b.h, &nb, &np, &l) })
	return uint64(nb), uint64(np), int(l)
}

// BfvPlan: what bfv.NewEvaluator builds for Mul (bfv/evaluator.go:89-112) and tensorAndRescale (:278-464).
type BfvPlan struct {
	contextQ *Context
	h        *C.lr_bfv_plan
}

func NewBfvPlan(contextQ, contextQMul *Context, t uint64, maxBatch int) *BfvPlan {
	p := &BfvPlan{contextQ: contextQ}
	call(func() C.int {
		return C.lr_bfv_plan_create_ex(contextQ.h, contextQMul.h, C.uint64_t(t), C.int(maxBatch), DefaultOptions.ptr(), &p.h)
	})
	runtime.SetFinalizer(p, func(p *BfvPlan) { C.lr_bfv_plan_destroy(p.h) })
	return p
}

// Mul = evaluator.Mul for two degree-1 ciphertexts (bfv/evaluator.go:467 -> tensorAndRescale).
func (p *BfvPlan) Mul(ct0, ct1 [2]*Poly, ctOut [3]*Poly) {
	p.contextQ.use(ct0[0], ct0[1], ct1[0], ct1[1])
	p.contextQ.want(ctOut[0], ctOut[1], ctOut[2])
	call(func() C.int {
		return C.lr_bfv_mul(p.h, ct0[0].d, ct0[1].d, ct1[0].d, ct1[1].d, ctOut[0].d, ctOut[1].d, ctOut[2].d)
	})
	done(ctOut[0], ctOut[1], ctOut[2])
}
