// Overlay for github.com/ldsec/lattigo/ckks (v1.3.1): drop this file into the package next to the upstream evaluator.go, with the
// module's ring package replaced by go/ring of this repository (INTEGRATION.md section 3).  It re-points the evaluator's ring-heavy
// methods at the device pipelines of liblattigo_ring_hip.so; everything the upstream bodies do BEFORE they touch the ring -- operand
// checks, level and scale bookkeeping, the panics -- is kept, by calling the same unexported helpers of the upstream file.
//
// NOT COMPILED IN THIS REPOSITORY'S PIPELINE (no Go toolchain in the image, INTEGRATION.md); tests/test_go_shim.py checks every C
// symbol reached through go/ring against the header, delimiter balance and the go 1.13 language level (no generics, no
// unsafe.Slice, no runtime.Pinner).
//
// Method by method (upstream line numbers in ckks/evaluator.go):
//
//	MulRelin            :1016-1133  checks as upstream, then ONE call: CkksPlan.MulRelin (tensor + key switch + the two Adds)
//	Rescale             :933-968    scale loop as upstream, DivRoundByLastModulusNTT on both components per level: CkksPlan.Rescale
//	switchKeysInPlace   :1475-1559  CkksPlan.SwitchKeysInPlace (the loops over Coeffs at :1519-1534 are inside the device pipeline)
//	permuteNTT          :1448-1468  CkksPlan.PermuteNTT (RotateColumns with the rotation's key, Conjugate)
//	RotateHoisted       :1252-1291  CkksPlan.RotateHoisted (decomposition once, all rotations in one call)
//	decomposeAndSplitNTT:1561-1591  not called any more (its loops at :1580-1586 live inside the pipelines)
//
// Direct Coeffs indexing elsewhere in the upstream evaluator, and what happens to it under this overlay:
//
//	AddConst            :429-445    host loop over half-vectors with two constants: bracketed (HostView in, HostWritten out)
//	MultByConstAndAdd   :588-606    same
//	MultByConst         :712-730    same
//	MultByi / DivByi    :765-779, :814-828  same
//	DropLevel           :910        re-slices Coeffs (metadata only): the device image is sized by cap(Coeffs) and keeps its stride
//	switchKeyHoisted    :1356-1360  replaced (RotateHoisted above)
//
// The bracketed methods stay correct on resident ciphertexts at the price of one PCIe round trip each (DESIGN.md section 4: ~1 ms per
// polynomial at N = 2^15); they are the constant-by-ciphertext operations, not the path BASELINE.json measures.  Add / Sub / Neg /
// MulByPow2 / Reduce go through Context methods and need nothing here.
package ckks

import (
	"errors"
	"math"

	"github.com/ldsec/lattigo/ring"
	"github.com/ldsec/lattigo/utils"
)

// deviceEvaluator is the upstream evaluator plus the device plan and the per-key images of the switching keys.
type deviceEvaluator struct {
	*evaluator
	plan *ring.CkksPlan
	keys map[*SwitchingKey]*ring.Poly // SwitchingKeyImage per key, uploaded once
}

// NewDeviceEvaluator = NewEvaluator (:78) + the plan.  maxBatch is 1: an evaluator holds one ciphertext at a time, like upstream;
// callers that want the batched throughput of DESIGN.md section 6 drive ring.CkksPlan themselves on arrays of ciphertexts.
func NewDeviceEvaluator(params *Parameters) Evaluator {
	base := NewEvaluator(params).(*evaluator)
	ev := &deviceEvaluator{evaluator: base, keys: map[*SwitchingKey]*ring.Poly{}}
	if len(params.Pi) != 0 {
		ev.plan = ring.NewCkksPlan(base.ckksContext.contextQ, base.ckksContext.contextP, 1)
	}
	return ev
}

func (eval *deviceEvaluator) keyImage(k *SwitchingKey) *ring.Poly {
	if img, ok := eval.keys[k]; ok {
		return img
	}
	img := eval.plan.SwitchingKeyImage(k.evakey)
	eval.keys[k] = img
	return img
}

// resident pins the value polynomials of an element: from here on only Pin / Sync / Unpin move them across PCIe.
func (eval *deviceEvaluator) resident(el *ckksElement) {
	q := eval.ckksContext.contextQ
	for _, p := range el.value {
		p.Pin(q)
	}
}

// MulRelin (:1016): upstream's checks, then the whole ring sequence as one device call.
func (eval *deviceEvaluator) MulRelin(op0, op1 Operand, evakey *EvaluationKey, ctOut *Ciphertext) {
	el0, el1, elOut := eval.getElemAndCheckBinary(op0, op1, ctOut, utils.MaxUint64(op0.Degree(), op1.Degree()))
	level := utils.MinUint64(utils.MinUint64(el0.Level(), el1.Level()), elOut.Level())
	if ctOut.Level() > level {
		eval.DropLevel(elOut.Ciphertext(), elOut.Level()-level)
	}
	if el0.Degree() > 1 || el1.Degree() > 1 {
		panic("cannot MulRelin: input elements must be of degree 0 or 1")
	}
	if !el0.IsNTT() {
		panic("cannot MulRelin: op0 must be in NTT")
	}
	if !el1.IsNTT() {
		panic("cannot MulRelin: op1 must be in NTT")
	}
	elOut.SetScale(el0.Scale() * el1.Scale())

	var key *ring.Poly
	if evakey != nil {
		key = eval.keyImage(evakey.evakey)
	}
	if el0.Degree()+el1.Degree() == 2 && evakey == nil {
		elOut.Resize(eval.params, 2) // degree-2 result, :1061-1066 / :1107-1111
	}
	eval.resident(el0)
	eval.resident(el1)
	eval.resident(elOut)
	// the pipeline works on its own temporaries, so ctOut may be either operand (upstream routes that case through ringpool, :1071-1076)
	eval.plan.MulRelin(level, el0.value, el1.value, key, elOut.value)
}

// Rescale (:933): the scale loop is upstream's; each iteration divides both components by the last modulus on the device.
func (eval *deviceEvaluator) Rescale(ct0 *Ciphertext, threshold float64, ctOut *Ciphertext) (err error) {
	ringContext := eval.ckksContext.contextQ
	if ct0.Level() == 0 {
		return errors.New("cannot Rescale: input Ciphertext already at level 0")
	}
	if ct0.Level() != ctOut.Level() {
		panic("cannot Rescale: degrees of receiver Ciphertext and input Ciphertext do not match")
	}
	if ct0.Scale() >= (threshold*float64(ringContext.Modulus[ctOut.Level()]))/2 {
		if !ct0.IsNTT() {
			panic("cannot Rescale: input Ciphertext not in NTT")
		}
		ctOut.Copy(ct0.Element())
		eval.resident(ctOut.Element())
		for ctOut.Scale() >= (threshold*float64(ringContext.Modulus[ctOut.Level()]))/2 && ctOut.Level() != 0 {
			ctOut.DivScale(float64(ringContext.Modulus[ctOut.Level()]))
			if len(ctOut.value) == 2 {
				eval.plan.Rescale([2]*ring.Poly{ctOut.value[0], ctOut.value[1]}) // re-slices Coeffs like :33 of ring_scaling.go
			} else {
				for i := range ctOut.value { // degree 0 or 2: component by component, still on the device
					ringContext.DivRoundByLastModulusNTT(ctOut.value[i])
				}
			}
		}
	} else {
		ctOut.Copy(ct0.Element())
	}
	return nil
}

// switchKeysInPlace (:1475).
func (eval *deviceEvaluator) switchKeysInPlace(level uint64, cx *ring.Poly, evakey *SwitchingKey, p0, p1 *ring.Poly) {
	q := eval.ckksContext.contextQ
	cx.Pin(q)
	p0.Pin(q)
	p1.Pin(q)
	eval.plan.SwitchKeysInPlace(level, cx, eval.keyImage(evakey), p0, p1)
}

// permuteNTT (:1448): both components permuted, the second key-switched, the additions folded into the pipeline's last pass.
func (eval *deviceEvaluator) permuteNTT(ct0 *Ciphertext, galEl uint64, evakey *SwitchingKey, ctOut *Ciphertext) {
	eval.resident(ct0.Element())
	eval.resident(ctOut.Element())
	level := utils.MinUint64(ct0.Level(), ctOut.Level())
	eval.plan.PermuteNTT(level, [2]*ring.Poly{ct0.value[0], ct0.value[1]}, galEl, eval.keyImage(evakey),
		[2]*ring.Poly{ctOut.value[0], ctOut.value[1]})
}

// RotateColumns (:1201) for a rotation whose key exists; the power-of-two fallback (:1226-1243) composes such rotations and is
// inherited from upstream unchanged (it calls permuteNTT through this type).
func (eval *deviceEvaluator) RotateColumns(ct0 *Ciphertext, k uint64, evakey *RotationKeys, ctOut *Ciphertext) {
	if ct0.Degree() != 1 || ctOut.Degree() != 1 {
		panic("cannot RotateColumns: input and output Ciphertext must be of degree 1")
	}
	k &= ((eval.ckksContext.n >> 1) - 1)
	if k == 0 {
		ctOut.Copy(ct0.Element())
		return
	}
	if evakey.evakeyRotColLeft[k] == nil {
		eval.evaluator.RotateColumns(ct0, k, evakey, ctOut) // composition out of power-of-two rotations, or upstream's panic
		return
	}
	ctOut.SetScale(ct0.Scale())
	eval.permuteNTT(ct0, ring.ModExp(GaloisGen, k, 2*eval.ckksContext.n), evakey.evakeyRotColLeft[k], ctOut)
}

// Conjugate (:1431).
func (eval *deviceEvaluator) Conjugate(ct0 *Ciphertext, evakey *RotationKeys, ctOut *Ciphertext) {
	if ct0.Degree() != 1 || ctOut.Degree() != 1 {
		panic("cannot Conjugate: input and output Ciphertext must be of degree 1")
	}
	if evakey.evakeyConjugate == nil {
		panic("cannot Conjugate: rows rotation key not generated")
	}
	ctOut.SetScale(ct0.Scale())
	eval.permuteNTT(ct0, 2*eval.ckksContext.n-1, evakey.evakeyConjugate, ctOut)
}

// RotateHoisted (:1252): one decomposition, every rotation's gather + inner product + ModDown in one device call.
func (eval *deviceEvaluator) RotateHoisted(ct0 *Ciphertext, rotations []uint64, rotkeys *RotationKeys) (cOut map[uint64]*Ciphertext) {
	cOut = make(map[uint64]*Ciphertext)
	eval.resident(ct0.Element())
	var gens []uint64
	var keys []*ring.Poly
	var outs [][2]*ring.Poly
	for _, i := range rotations {
		i &= ((eval.ckksContext.n >> 1) - 1)
		if _, seen := cOut[i]; seen {
			continue
		}
		if i == 0 {
			cOut[i] = ct0.CopyNew().Ciphertext()
			continue
		}
		if rotkeys.permuteNTTLeftIndex[i] == nil {
			panic("cannot switchKeyHoisted: specific rotation has not been generated")
		}
		ct := NewCiphertext(eval.params, 1, ct0.Level(), ct0.Scale())
		eval.resident(ct.Element())
		cOut[i] = ct
		gens = append(gens, ring.ModExp(GaloisGen, i, 2*eval.ckksContext.n))
		keys = append(keys, eval.keyImage(rotkeys.evakeyRotColLeft[i]))
		outs = append(outs, [2]*ring.Poly{ct.value[0], ct.value[1]})
	}
	if len(gens) > 0 {
		eval.plan.RotateHoisted(ct0.Level(), [2]*ring.Poly{ct0.value[0], ct0.value[1]}, gens, keys, outs)
	}
	return
}

// hostLoop runs one of the upstream methods that index Coeffs directly (list in the header) on possibly resident ciphertexts.
func (eval *deviceEvaluator) hostLoop(in, out *Ciphertext, body func()) {
	for _, p := range in.value {
		p.HostView()
	}
	if out != in {
		for _, p := range out.value {
			p.HostView()
		}
	}
	body()
	for _, p := range out.value {
		p.HostWritten()
	}
}

func (eval *deviceEvaluator) AddConst(ct0 *Ciphertext, constant interface{}, ctOut *Ciphertext) {
	eval.hostLoop(ct0, ctOut, func() { eval.evaluator.AddConst(ct0, constant, ctOut) })
}

func (eval *deviceEvaluator) MultByConstAndAdd(ct0 *Ciphertext, constant interface{}, ctOut *Ciphertext) {
	eval.hostLoop(ct0, ctOut, func() { eval.evaluator.MultByConstAndAdd(ct0, constant, ctOut) })
}

func (eval *deviceEvaluator) MultByConst(ct0 *Ciphertext, constant interface{}, ctOut *Ciphertext) {
	eval.hostLoop(ct0, ctOut, func() { eval.evaluator.MultByConst(ct0, constant, ctOut) })
}

func (eval *deviceEvaluator) MultByi(ct0 *Ciphertext, ctOut *Ciphertext) {
	eval.hostLoop(ct0, ctOut, func() { eval.evaluator.MultByi(ct0, ctOut) })
}

func (eval *deviceEvaluator) DivByi(ct0 *Ciphertext, ctOut *Ciphertext) {
	eval.hostLoop(ct0, ctOut, func() { eval.evaluator.DivByi(ct0, ctOut) })
}

// levelsFor mirrors :1508 (beta at a level); kept for callers that size their own key images.
func (eval *deviceEvaluator) levelsFor(level uint64) uint64 {
	return uint64(math.Ceil(float64(level+1) / float64(eval.params.Alpha())))
}
