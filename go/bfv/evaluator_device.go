// Overlay for github.com/ldsec/lattigo/bfv (v1.3.1): drop this file into the package next to the upstream evaluator.go, with the
// module's ring package replaced by go/ring of this repository (INTEGRATION.md section 3).
//
// NOT COMPILED IN THIS REPOSITORY'S PIPELINE (no Go toolchain in the image); statically checked by tests/test_go_shim.py.
//
// Method by method (upstream line numbers in bfv/evaluator.go):
//
//	Mul / tensorAndRescale :278-470   degree-1 x degree-1: ONE call, BfvPlan.Mul (extension to QMul, transforms, tensor, division by Q
//	                                  with the float-corrected extension, centring, extension back, times t); other degrees: upstream
//	relinearize            :480-501   degree 2: ONE call, CkksPlan.BfvRelinearize (key switch + the two Adds); higher degrees: the
//	                                  upstream loop over switchKeys below
//	switchKeys             :736-812   CkksPlan.BfvSwitchKeys (the per-modulus loops over Coeffs at :776-792 are inside the pipeline)
//
// Direct Coeffs indexing elsewhere in the upstream evaluator: tensorAndRescale's copy into polyBig (:430-436) is inside BfvPlan.Mul for
// the degree-1 case and runs on the host, bracketed by HostView / HostWritten, for the others.  Add / Sub / Neg / MulScalar go through
// Context methods and need nothing here; rotations (:560-730) use switchKeys through this type.
package bfv

import (
	"github.com/ldsec/lattigo/ring"
)

type deviceEvaluator struct {
	*evaluator
	mul  *ring.BfvPlan
	ks   *ring.CkksPlan // decomposer, baseconverterQ1P and key-switch pools of :100-112, on the device
	keys map[*SwitchingKey]*ring.Poly
}

// NewDeviceEvaluator = NewEvaluator (:89) + the two plans.
func NewDeviceEvaluator(params *Parameters) Evaluator {
	base := NewEvaluator(params).(*evaluator)
	ctx := base.bfvContext
	ev := &deviceEvaluator{evaluator: base, keys: map[*SwitchingKey]*ring.Poly{}}
	ev.mul = ring.NewBfvPlan(ctx.contextQ, ctx.contextQMul, params.T, 1)
	if len(params.Pi) != 0 {
		ev.ks = ring.NewCkksPlan(ctx.contextQ, ctx.contextP, 1)
	}
	return ev
}

func (eval *deviceEvaluator) keyImage(k *SwitchingKey) *ring.Poly {
	if img, ok := eval.keys[k]; ok {
		return img
	}
	img := eval.ks.SwitchingKeyImage(k.evakey)
	eval.keys[k] = img
	return img
}

func (eval *deviceEvaluator) resident(ps ...*ring.Poly) {
	q := eval.bfvContext.contextQ
	for _, p := range ps {
		p.Pin(q)
	}
}

// Mul (:467): degree-1 x degree-1 on the device; anything else through upstream's tensorAndRescale on host views.
func (eval *deviceEvaluator) Mul(op0 *Ciphertext, op1 Operand, ctOut *Ciphertext) {
	el0, el1, elOut := eval.getElemAndCheckBinary(op0, op1, ctOut, op0.Degree()+op1.Degree())
	if el0.Degree() == 1 && el1.Degree() == 1 {
		eval.resident(el0.value[0], el0.value[1], el1.value[0], el1.value[1], elOut.value[0], elOut.value[1], elOut.value[2])
		eval.mul.Mul([2]*ring.Poly{el0.value[0], el0.value[1]}, [2]*ring.Poly{el1.value[0], el1.value[1]},
			[3]*ring.Poly{elOut.value[0], elOut.value[1], elOut.value[2]})
		return
	}
	for _, p := range el0.value {
		p.HostView()
	}
	for _, p := range el1.value {
		p.HostView()
	}
	eval.evaluator.Mul(op0, op1, ctOut)
	for _, p := range elOut.value {
		p.HostWritten()
	}
}

// switchKeys (:736).
func (eval *deviceEvaluator) switchKeys(cx *ring.Poly, evakey *SwitchingKey, p0, p1 *ring.Poly) {
	eval.resident(cx, p0, p1)
	eval.ks.BfvSwitchKeys(cx, eval.keyImage(evakey), p0, p1)
}

// relinearize (:480).
func (eval *deviceEvaluator) relinearize(ct0 *Ciphertext, evakey *EvaluationKey, ctOut *Ciphertext) {
	context := eval.bfvContext.contextQ
	if ct0.Degree() == 2 {
		eval.resident(ct0.value[0], ct0.value[1], ct0.value[2], ctOut.value[0], ctOut.value[1])
		eval.ks.BfvRelinearize([3]*ring.Poly{ct0.value[0], ct0.value[1], ct0.value[2]}, eval.keyImage(evakey.evakey[0]),
			[2]*ring.Poly{ctOut.value[0], ctOut.value[1]})
		ctOut.SetValue(ctOut.value[:2])
		return
	}
	if ctOut != ct0 {
		context.Copy(ct0.value[0], ctOut.value[0])
		context.Copy(ct0.value[1], ctOut.value[1])
	}
	p0, p1 := eval.keyswitchpool[2], eval.keyswitchpool[3]
	for deg := uint64(ct0.Degree()); deg > 1; deg-- {
		eval.switchKeys(ct0.value[deg], evakey.evakey[deg-2], p0, p1)
		context.Add(ctOut.value[0], p0, ctOut.value[0])
		context.Add(ctOut.value[1], p1, ctOut.value[1])
	}
	ctOut.SetValue(ctOut.value[:2])
}

// Relinearize (:512): upstream's degree checks, then relinearize above.
func (eval *deviceEvaluator) Relinearize(ct0 *Ciphertext, evakey *EvaluationKey, ctOut *Ciphertext) {
	if int(ct0.Degree()-1) > len(evakey.evakey) {
		panic("cannot Relinearize: input ciphertext degree too large to allow relinearization")
	}
	if ct0.Degree() < 2 {
		if ct0 != ctOut {
			ctOut.Copy(ct0.Element())
		}
	} else {
		eval.relinearize(ct0, evakey, ctOut)
	}
}
