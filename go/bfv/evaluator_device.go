// Replacement bodies for github.com/ldsec/lattigo/bfv (v1.3.1), evaluator.go: this file is added to the package, the module's ring
// package is replaced by go/ring of this repository (INTEGRATION.md section 3), and the upstream bodies of the methods defined here are
// DELETED from evaluator.go (same receivers and signatures: Go has no virtual dispatch, see go/ckks/evaluator_device.go).
//
// NOT COMPILED IN THIS REPOSITORY'S PIPELINE (no Go toolchain in the image); statically checked by tests/test_go_shim.py.
//
// The patch to upstream bfv/evaluator.go, line numbers of v1.3.1:
//
//	delete  Mul          :467-470   -> below: degree 1 x degree 1 is ONE call, BfvPlan.Mul (extension to QMul, transforms, tensor, division by
//	                                   Q with the float-corrected extension, centring, extension back, times t); other degrees go
//	                                   through upstream's tensorAndRescale (:278-464, kept) on host views -- its copy loop into
//	                                   polyBig (:430-436) indexes Coeffs
//	delete  relinearize  :480-501   -> below: degree 2 is ONE call, CkksPlan.BfvRelinearize (key switch + the two Adds); higher degrees
//	                                   keep upstream's loop over switchKeys
//	delete  switchKeys   :736-812   -> below: CkksPlan.BfvSwitchKeys (the per-modulus loops over Coeffs at :776-792 live inside the pipeline)
//	delete  permute      :711-735   -> below: ONE call, CkksPlan.BfvPermute (Context.Permute x2, the key switch, Add, Copy); RotateRows,
//	                                   RotateColumns and rotateColumnsPow2 (:579-681, kept) reach the device through it
//	keep    Relinearize :512, SwitchKeys :539, the rotation front ends :560-709, InnerSum :683, Add / Sub / Neg / MulScalar (Context
//	        methods)
package bfv

import (
	"sync"

	"github.com/ldsec/lattigo/ring"
)

type deviceState struct {
	mul  *ring.BfvPlan
	ks   *ring.CkksPlan // decomposer, baseconverterQ1P and key-switch pools of :89-112, on the device
	keys map[*SwitchingKey]*ring.Poly
}

var deviceStates sync.Map // *evaluator -> *deviceState

// deviceBatcher, when set, takes the degree-1 x degree-1 Mul and the degree-2 Relinearize calls of EVERY evaluator over the same moduli:
// upstream's own pooled workload (examples/dbfv/psi/psi.go:215-233: one evaluator per goroutine, Mul + Relinearize per task) leaves the
// device mostly idle per call; the batcher runs the calls that are in flight together as one batched pipeline (ring.BfvBatcher).
var deviceBatcher *ring.BfvBatcher

// EnableDeviceBatcher builds the shared batcher for a parameter set: maxBatch ciphertexts per launch, `lanes` launches in flight (2).
func EnableDeviceBatcher(params *Parameters, maxBatch, lanes int) {
	c := newBFVContext(params)
	deviceBatcher = ring.NewBfvBatcher(c.n, c.contextQ.Modulus, c.contextP.Modulus, c.contextQMul.Modulus, params.T, maxBatch, lanes)
}

func (evaluator *evaluator) batcher() *ring.BfvBatcher {
	b := deviceBatcher
	q := evaluator.bfvContext.contextQ
	if b == nil || evaluator.baseconverterQ1P == nil || b.N != q.N || len(b.Q) != len(q.Modulus) || b.T != evaluator.params.T {
		return nil
	}
	for i, qi := range q.Modulus {
		if b.Q[i] != qi {
			return nil
		}
	}
	return b
}

func (evaluator *evaluator) dev() *deviceState {
	if s, ok := deviceStates.Load(evaluator); ok {
		return s.(*deviceState)
	}
	ctx := evaluator.bfvContext
	s := &deviceState{keys: map[*SwitchingKey]*ring.Poly{}}
	s.mul = ring.NewBfvPlan(ctx.contextQ, ctx.contextQMul, evaluator.params.T, 1)
	if evaluator.baseconverterQ1P != nil {
		s.ks = ring.NewCkksPlan(ctx.contextQ, ctx.contextP, 1)
	}
	actual, _ := deviceStates.LoadOrStore(evaluator, s)
	return actual.(*deviceState)
}

// ReleaseDevice drops the evaluator's device state (plans, key images) and its entry in deviceStates; see the ckks overlay.
func (evaluator *evaluator) ReleaseDevice() {
	if s, ok := deviceStates.Load(evaluator); ok {
		st := s.(*deviceState)
		for k := range st.keys {
			delete(st.keys, k)
		}
		st.mul, st.ks = nil, nil
		deviceStates.Delete(evaluator)
	}
}

func (evaluator *evaluator) keyImage(k *SwitchingKey) *ring.Poly {
	s := evaluator.dev()
	if img, ok := s.keys[k]; ok {
		return img
	}
	img := s.ks.SwitchingKeyImage(k.evakey)
	s.keys[k] = img
	return img
}

func (evaluator *evaluator) resident(ps ...*ring.Poly) {
	q := evaluator.bfvContext.contextQ
	for _, p := range ps {
		p.Pin(q)
	}
}

// Mul (:467): degree-1 x degree-1 on the device; anything else through upstream's tensorAndRescale on host views.
func (evaluator *evaluator) Mul(op0 *Ciphertext, op1 Operand, ctOut *Ciphertext) {
	el0, el1, elOut := evaluator.getElemAndCheckBinary(op0, op1, ctOut, op0.Degree()+op1.Degree())
	if el0.Degree() == 1 && el1.Degree() == 1 {
		evaluator.resident(el0.value[0], el0.value[1], el1.value[0], el1.value[1], elOut.value[0], elOut.value[1], elOut.value[2])
		if b := evaluator.batcher(); b != nil {
			b.Mul(evaluator.bfvContext.contextQ, [2]*ring.Poly{el0.value[0], el0.value[1]}, [2]*ring.Poly{el1.value[0], el1.value[1]},
				[3]*ring.Poly{elOut.value[0], elOut.value[1], elOut.value[2]})
			return
		}
		evaluator.dev().mul.Mul([2]*ring.Poly{el0.value[0], el0.value[1]}, [2]*ring.Poly{el1.value[0], el1.value[1]},
			[3]*ring.Poly{elOut.value[0], elOut.value[1], elOut.value[2]})
		return
	}
	for _, p := range el0.value {
		p.HostView()
	}
	for _, p := range el1.value {
		p.HostView()
	}
	evaluator.tensorAndRescale(el0, el1, elOut)
	for _, p := range elOut.value {
		p.HostWritten()
	}
}

// switchKeys (:736).
func (evaluator *evaluator) switchKeys(cx *ring.Poly, evakey *SwitchingKey, p0, p1 *ring.Poly) {
	evaluator.resident(cx, p0, p1)
	evaluator.dev().ks.BfvSwitchKeys(cx, evaluator.keyImage(evakey), p0, p1)
}

// relinearize (:480).
func (evaluator *evaluator) relinearize(ct0 *Ciphertext, evakey *EvaluationKey, ctOut *Ciphertext) {
	context := evaluator.bfvContext.contextQ
	if ct0.Degree() == 2 {
		evaluator.resident(ct0.value[0], ct0.value[1], ct0.value[2], ctOut.value[0], ctOut.value[1])
		if b := evaluator.batcher(); b != nil {
			b.Relinearize(context, [3]*ring.Poly{ct0.value[0], ct0.value[1], ct0.value[2]}, b.KeyImage(evakey.evakey[0].evakey),
				[2]*ring.Poly{ctOut.value[0], ctOut.value[1]})
			ctOut.SetValue(ctOut.value[:2])
			return
		}
		evaluator.dev().ks.BfvRelinearize([3]*ring.Poly{ct0.value[0], ct0.value[1], ct0.value[2]}, evaluator.keyImage(evakey.evakey[0]),
			[2]*ring.Poly{ctOut.value[0], ctOut.value[1]})
		ctOut.SetValue(ctOut.value[:2])
		return
	}
	if ctOut != ct0 {
		context.Copy(ct0.value[0], ctOut.value[0])
		context.Copy(ct0.value[1], ctOut.value[1])
	}
	p0, p1 := evaluator.keyswitchpool[2], evaluator.keyswitchpool[3]
	for deg := uint64(ct0.Degree()); deg > 1; deg-- {
		evaluator.switchKeys(ct0.value[deg], evakey.evakey[deg-2], p0, p1)
		context.Add(ctOut.value[0], p0, ctOut.value[0])
		context.Add(ctOut.value[1], p1, ctOut.value[1])
	}
	ctOut.SetValue(ctOut.value[:2])
}

// permute (:711): the Galois automorphism on both components, the key switch of the second, Add and Copy, as one call.
func (evaluator *evaluator) permute(ct0 *Ciphertext, generator uint64, switchKey *SwitchingKey, ctOut *Ciphertext) {
	evaluator.resident(ct0.value[0], ct0.value[1], ctOut.value[0], ctOut.value[1])
	evaluator.dev().ks.BfvPermute([2]*ring.Poly{ct0.value[0], ct0.value[1]}, generator, evaluator.keyImage(switchKey),
		[2]*ring.Poly{ctOut.value[0], ctOut.value[1]})
}
