// Replacement bodies for github.com/ldsec/lattigo/ckks (v1.3.1), evaluator.go: this file is added to the package, the module's ring
// package is replaced by go/ring of this repository (INTEGRATION.md section 3), and the upstream bodies of the methods defined here are
// DELETED from evaluator.go (Go has no virtual dispatch: an embedding wrapper would leave Power, EvaluatePoly*, EvaluateCheby*, the
// ...New wrappers and the power-of-two rotation composer calling the upstream bodies, whose loops over Coeffs read stale host data on
// device-resident ciphertexts).  Same receivers, same signatures, so every upstream caller -- polynomial_evaluation.go, chebyshev
// interpolation, the examples -- reaches the device pipelines unchanged.
//
// NOT COMPILED IN THIS REPOSITORY'S PIPELINE (no Go toolchain in the image, INTEGRATION.md); tests/test_go_shim.py checks every plan /
// poly method used here against go/ring, every upstream identifier used here against the reference package, delimiter balance and the
// go 1.13 language level (no generics, no unsafe.Slice, no runtime.Pinner).
//
// The patch to upstream ckks/evaluator.go, line numbers of v1.3.1:
//
//	delete  MulRelin             :1016-1133   -> below: upstream's checks, then ONE call, CkksPlan.MulRelin (tensor + key switch + Adds)
//	delete  Relinearize          :1144-1162   -> below (key switch with the two Adds folded into its last pass)
//	delete  SwitchKeys           :1176-1190   -> below
//	delete  Rescale              :933-968     -> below: upstream's scale loop; per level both components, CkksPlan.Rescale
//	delete  RotateHoisted        :1252-1289   -> below: one decomposition, every rotation in one call, CkksPlan.RotateHoisted
//	delete  switchKeyHoisted     :1291-1392   (its loops over Coeffs at :1356-1360 live inside the pipeline)
//	delete  permuteNTT           :1452-1472   -> below, same signature: CkksPlan.PermuteNTT (RotateColumns, Conjugate and the
//	                                             power-of-two composer rotateColumnsPow2 :1402-1427 call it and stay as they are)
//	delete  switchKeysInPlace    :1475-1558   -> below (the loops over Coeffs at :1519-1534 live inside the pipeline)
//	delete  decomposeAndSplitNTT :1561-1591   (no caller left; its loops at :1580-1586 live inside the pipelines)
//	patch   AddConst :373, MultByConstAndAdd :451, MultByConst :622, MultByi :746, DivByi :795 -- their element loops index Coeffs
//	        with one constant for the coefficients below n/2 and one for the rest (:429-445, :588-606, :712-730, :765-779, :814-828).
//	        The constant computation per limb (scaleUpExact, MRed by nttPsi[i][1], MForm) stays as it is; inside the limb loop the
//	        two `for j` / `for u` blocks are replaced by `lo[i], hi[i] = <constant of the first block>, <constant of the second>`
//	        (lo, hi := make([]uint64, level+1) before the loop), and after the limb loop ONE line runs the element loops on the
//	        device: `eval.halfScalar(op, level, ct0, ctOut, lo, hi, first)` (below) with op 0 / first = true for AddConst (value[0]
//	        only, CRed(x + s)), op 2 for MultByConstAndAdd (CRed(out + MRed(x, s))), op 1 for MultByConst, MultByi, DivByi
//	        (MRed(x, s)).  Polynomial evaluation (EvaluatePoly*, EvaluateCheby*) calls these between its MulRelin / Rescale steps: with
//	        the patch a whole evaluation stays on the device.  (Unpatched, the first statement `defer eval.hostLoop(ct0, ctOut)()` keeps
//	        the upstream loops correct on resident ciphertexts at the price of one PCIe round trip per call.)
//	keep    DropLevel :901 (it re-slices Coeffs: metadata; the device image is sized by cap(Coeffs) and keeps its stride), RescaleMany
//	        :971 (Context.DivRoundByLastModulusManyNTT is a device call), Add / Sub / Neg / MulByPow2 / Reduce / ScaleUp (Context methods)
package ckks

import (
	"errors"
	"sync"

	"github.com/ldsec/lattigo/ring"
	"github.com/ldsec/lattigo/utils"
)

// deviceState is what the evaluator struct would carry if a file could add fields to it: the plan (decomposer, base converter and
// pools of ckks/evaluator.go:81-112 on the device) and the device images of the switching keys, uploaded once per key.
type deviceState struct {
	plan *ring.CkksPlan
	keys map[*SwitchingKey]*ring.Poly
}

var deviceStates sync.Map // *evaluator -> *deviceState

func (eval *evaluator) dev() *deviceState {
	if s, ok := deviceStates.Load(eval); ok {
		return s.(*deviceState)
	}
	s := &deviceState{keys: map[*SwitchingKey]*ring.Poly{}}
	if eval.baseconverter != nil { // special primes present (:91-97)
		s.plan = ring.NewCkksPlan(eval.ckksContext.contextQ, eval.ckksContext.contextP, 1)
	}
	actual, _ := deviceStates.LoadOrStore(eval, s)
	return actual.(*deviceState)
}

// deviceBatcher, when set, takes the ciphertext x ciphertext MulRelin calls of EVERY evaluator over the same moduli: upstream's
// concurrency model is one evaluator per goroutine and one ciphertext per call (examples/dbfv/psi/psi.go:215-233), which leaves the
// device mostly idle per call; the batcher runs the calls that are in flight together as one batched pipeline
// (ring.CkksBatcher, lr_ckks_batcher_* in lattigo_ring.h).  Set it once, before the goroutines start.
var deviceBatcher *ring.CkksBatcher

// EnableDeviceBatcher builds the shared batcher for a parameter set: maxBatch polys per launch, `lanes` launches in flight (2).
func EnableDeviceBatcher(params *Parameters, maxBatch, lanes int) {
	c := newContext(params)
	deviceBatcher = ring.NewCkksBatcher(c.n, c.contextQ.Modulus, c.contextP.Modulus, maxBatch, lanes)
}

func (eval *evaluator) batcher() *ring.CkksBatcher {
	b := deviceBatcher
	q := eval.ckksContext.contextQ
	if b == nil || eval.baseconverter == nil || b.N != q.N || len(b.Q) != len(q.Modulus) {
		return nil
	}
	for i, qi := range q.Modulus {
		if b.Q[i] != qi {
			return nil
		}
	}
	return b
}

// ReleaseDevice drops what this evaluator holds on the device -- its plan (pools, decomposer tables) and its key images -- and its
// entry in deviceStates; the handles' finalizers free the device memory.  Call it when the goroutine that owns the evaluator is done
// (the evaluator struct cannot carry a finalizer of its own: the map holds it alive).  A later call on the evaluator builds a new state.
func (eval *evaluator) ReleaseDevice() {
	if s, ok := deviceStates.Load(eval); ok {
		st := s.(*deviceState)
		for k := range st.keys {
			delete(st.keys, k)
		}
		st.plan = nil
		deviceStates.Delete(eval)
	}
}

func (eval *evaluator) keyImage(k *SwitchingKey) *ring.Poly {
	s := eval.dev()
	if img, ok := s.keys[k]; ok {
		return img
	}
	img := s.plan.SwitchingKeyImage(k.evakey)
	s.keys[k] = img
	return img
}

// resident pins polynomials: from here on only Pin / Sync / Unpin move them across PCIe.
func (eval *evaluator) resident(ps ...*ring.Poly) {
	q := eval.ckksContext.contextQ
	for _, p := range ps {
		p.Pin(q)
	}
}

// hostLoop brackets an upstream method that indexes Coeffs directly: `defer eval.hostLoop(ct0, ctOut)()`.
func (eval *evaluator) hostLoop(in, out *Ciphertext) func() {
	for _, p := range in.value {
		p.HostView()
	}
	if out != in {
		for _, p := range out.value {
			p.HostView()
		}
	}
	return func() {
		for _, p := range out.value {
			p.HostWritten()
		}
	}
}

// halfScalar is the device form of the element loops of AddConst / MultByConstAndAdd / MultByConst / MultByi / DivByi (patch list above):
// lo[i] / hi[i] are the constants upstream computes for limb i and the two halves of the coefficient vector.
func (eval *evaluator) halfScalar(op int, level uint64, ct0, ctOut *Ciphertext, lo, hi []uint64, first bool) {
	context := eval.ckksContext.contextQ
	for u := range ct0.value {
		if first && u > 0 {
			break
		}
		eval.resident(ct0.value[u], ctOut.value[u])
		context.HalfScalarOp(op, level, ct0.value[u], lo, hi, ctOut.value[u])
	}
}

// galoisElement recovers the Galois element from the NTT permutation table upstream passes around (ring.PermuteNTTIndex,
// ring/ring_galois.go:29-53: index[i] = bitrev(((g * (2 bitrev(i) + 1)) mod 2N - 1) / 2), so g = 2 bitrev(index[0]) + 1).
func (eval *evaluator) galoisElement(index []uint64) uint64 {
	logN := eval.ckksContext.logN
	var r uint64
	for b := uint64(0); b < logN; b++ {
		r |= ((index[0] >> b) & 1) << (logN - 1 - b)
	}
	return 2*r + 1
}

// MulRelin (:1016): upstream's checks, then the whole ring sequence as one device call.
func (eval *evaluator) MulRelin(op0, op1 Operand, evakey *EvaluationKey, ctOut *Ciphertext) {
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
	if el0.Degree()+el1.Degree() == 2 && evakey == nil {
		elOut.Resize(eval.params, 2) // degree-2 result, :1061-1066 / :1107-1111
	}
	eval.resident(el0.value...)
	eval.resident(el1.value...)
	eval.resident(elOut.value...)
	// the pipeline works on its own temporaries, so ctOut may be either operand (upstream routes that case through ringpool, :1071-1076)
	if b := eval.batcher(); b != nil && evakey != nil && el0.Degree() == 1 && el1.Degree() == 1 {
		b.MulRelin(eval.ckksContext.contextQ, level, el0.value, el1.value, b.KeyImage(evakey.evakey.evakey), elOut.value)
		return
	}
	// (after the batcher branch: an evaluator whose products go through the batcher never uploads a private image of the key --
	// 360 MB at PN16QP1761, times the goroutines)
	var key *ring.Poly
	if evakey != nil {
		key = eval.keyImage(evakey.evakey)
	}
	eval.dev().plan.MulRelin(level, el0.value, el1.value, key, elOut.value)
}

// Relinearize (:1144).
func (eval *evaluator) Relinearize(ct0 *Ciphertext, evakey *EvaluationKey, ctOut *Ciphertext) {
	if ct0.Degree() != 2 {
		panic("cannot Relinearize: input Ciphertext is not of degree 2")
	}
	if ctOut != ct0 {
		ctOut.SetScale(ct0.Scale())
	}
	level := utils.MinUint64(ct0.Level(), ctOut.Level())
	context := eval.ckksContext.contextQ
	eval.switchKeysInPlace(level, ct0.value[2], evakey.evakey, eval.poolQ[1], eval.poolQ[2])
	eval.resident(ct0.value[0], ct0.value[1], ctOut.value[0], ctOut.value[1])
	context.AddLvl(level, ct0.value[0], eval.poolQ[1], ctOut.value[0])
	context.AddLvl(level, ct0.value[1], eval.poolQ[2], ctOut.value[1])
	ctOut.Resize(eval.params, 1)
}

// SwitchKeys (:1176).
func (eval *evaluator) SwitchKeys(ct0 *Ciphertext, switchingKey *SwitchingKey, ctOut *Ciphertext) {
	if ct0.Degree() != 1 || ctOut.Degree() != 1 {
		panic("cannot SwitchKeys: input and output Ciphertext must be of degree 1")
	}
	level := utils.MinUint64(ct0.Level(), ctOut.Level())
	context := eval.ckksContext.contextQ
	eval.switchKeysInPlace(level, ct0.value[1], switchingKey, eval.poolQ[1], eval.poolQ[2])
	eval.resident(ct0.value[0], ctOut.value[0], ctOut.value[1])
	context.AddLvl(level, ct0.value[0], eval.poolQ[1], ctOut.value[0])
	context.CopyLvl(level, eval.poolQ[2], ctOut.value[1])
}

// Rescale (:933): the scale loop is upstream's; each iteration divides both components by the last modulus on the device.
func (eval *evaluator) Rescale(ct0 *Ciphertext, threshold float64, ctOut *Ciphertext) (err error) {
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
		eval.resident(ctOut.value...)
		for ctOut.Scale() >= (threshold*float64(ringContext.Modulus[ctOut.Level()]))/2 && ctOut.Level() != 0 {
			ctOut.DivScale(float64(ringContext.Modulus[ctOut.Level()]))
			if len(ctOut.value) == 2 && eval.dev().plan != nil {
				eval.dev().plan.Rescale([2]*ring.Poly{ctOut.value[0], ctOut.value[1]}) // re-slices Coeffs like ring_scaling.go:33
			} else {
				for i := range ctOut.value { // degree 0 or 2, or no special primes: component by component, still on the device
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
func (eval *evaluator) switchKeysInPlace(level uint64, cx *ring.Poly, evakey *SwitchingKey, p0, p1 *ring.Poly) {
	eval.resident(cx, p0, p1)
	eval.dev().plan.SwitchKeysInPlace(level, cx, eval.keyImage(evakey), p0, p1)
}

// permuteNTT (:1452): both components permuted, the second key-switched, the addition and the copy folded into the pipeline's last
// pass.  RotateColumns, Conjugate and rotateColumnsPow2 call it with the index table of the rotation; the device gathers from the
// Galois element, which the table determines.
func (eval *evaluator) permuteNTT(ct0 *Ciphertext, index []uint64, evakey *SwitchingKey, ctOut *Ciphertext) {
	eval.resident(ct0.value[0], ct0.value[1], ctOut.value[0], ctOut.value[1])
	level := utils.MinUint64(ct0.Level(), ctOut.Level())
	if b := eval.batcher(); b != nil { // the rotations of the evaluators in flight together run as one batched rotation per (level, element, key)
		b.PermuteNTT(eval.ckksContext.contextQ, level, [2]*ring.Poly{ct0.value[0], ct0.value[1]}, eval.galoisElement(index),
			b.KeyImage(evakey.evakey), [2]*ring.Poly{ctOut.value[0], ctOut.value[1]})
		return
	}
	eval.dev().plan.PermuteNTT(level, [2]*ring.Poly{ct0.value[0], ct0.value[1]}, eval.galoisElement(index), eval.keyImage(evakey),
		[2]*ring.Poly{ctOut.value[0], ctOut.value[1]})
}

// RotateHoisted (:1252): one decomposition, every rotation's gather + inner product + ModDown in one device call.
func (eval *evaluator) RotateHoisted(ct0 *Ciphertext, rotations []uint64, rotkeys *RotationKeys) (cOut map[uint64]*Ciphertext) {
	cOut = make(map[uint64]*Ciphertext)
	eval.resident(ct0.value[0], ct0.value[1])
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
		eval.resident(ct.value[0], ct.value[1])
		cOut[i] = ct
		gens = append(gens, eval.galoisElement(rotkeys.permuteNTTLeftIndex[i]))
		keys = append(keys, eval.keyImage(rotkeys.evakeyRotColLeft[i]))
		outs = append(outs, [2]*ring.Poly{ct.value[0], ct.value[1]})
	}
	if len(gens) > 0 {
		eval.dev().plan.RotateHoisted(ct0.Level(), [2]*ring.Poly{ct0.value[0], ct0.value[1]}, gens, keys, outs)
	}
	return
}
