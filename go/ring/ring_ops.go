package ring

// #include "lattigo_ring.h"
import "C"

import (
	"math/big"
	"math/bits"
	"unsafe"
)

// ew is the one forwarding point of the coefficient-wise family (ring/ring.go): op on limbs [0, level].
func (c *Context) ew(op C.int, level uint64, a, b, o *Poly, scalars []uint64) {
	c.use(a, b)
	if readsOut(op) {
		c.use(o)
	} else {
		c.want(o)
	}
	var bd *C.lr_poly
	if b != nil {
		bd = b.d
	}
	var sp *C.uint64_t
	if scalars != nil {
		sp = (*C.uint64_t)(unsafe.Pointer(&scalars[0]))
	}
	call(func() C.int { return C.lr_ewise(c.h, op, C.int(level), a.d, bd, o.d, sp) })
	done(o)
}

func readsOut(op C.int) bool {
	switch op {
	case C.LR_MUL_COEFFS_AND_ADD, C.LR_MUL_COEFFS_AND_ADD_NOMOD, C.LR_MUL_MONT_AND_ADD, C.LR_MUL_MONT_AND_ADD_NOMOD,
		C.LR_MUL_MONT_CONSTANT_AND_ADD_NOMOD, C.LR_MUL_MONT_AND_SUB, C.LR_MUL_MONT_AND_SUB_NOMOD:
		return true
	}
	return false
}

// residues of a big scalar, one per modulus (what MulScalarBigint / AddScalarBigint compute per limb).
func (c *Context) residues(s *big.Int) []uint64 {
	out := make([]uint64, len(c.Modulus))
	t := new(big.Int)
	for i, q := range c.Modulus {
		out[i] = t.Mod(s, new(big.Int).SetUint64(q)).Uint64()
	}
	return out
}

func (c *Context) repeat(v uint64) []uint64 {
	out := make([]uint64, len(c.Modulus))
	for i := range out {
		out[i] = v
	}
	return out
}

// ring/ring.go:10-143
func (c *Context) Add(p1, p2, p3 *Poly)                       { c.ew(C.LR_ADD, c.top(), p1, p2, p3, nil) }
func (c *Context) AddLvl(level uint64, p1, p2, p3 *Poly)      { c.ew(C.LR_ADD, level, p1, p2, p3, nil) }
func (c *Context) AddNoMod(p1, p2, p3 *Poly)                  { c.ew(C.LR_ADD_NOMOD, c.top(), p1, p2, p3, nil) }
func (c *Context) AddNoModLvl(level uint64, p1, p2, p3 *Poly) { c.ew(C.LR_ADD_NOMOD, level, p1, p2, p3, nil) }
func (c *Context) Sub(p1, p2, p3 *Poly)                       { c.ew(C.LR_SUB, c.top(), p1, p2, p3, nil) }
func (c *Context) SubLvl(level uint64, p1, p2, p3 *Poly)      { c.ew(C.LR_SUB, level, p1, p2, p3, nil) }
func (c *Context) SubNoMod(p1, p2, p3 *Poly)                  { c.ew(C.LR_SUB_NOMOD, c.top(), p1, p2, p3, nil) }
func (c *Context) SubNoModLvl(level uint64, p1, p2, p3 *Poly) { c.ew(C.LR_SUB_NOMOD, level, p1, p2, p3, nil) }
func (c *Context) Neg(p1, p2 *Poly)                           { c.ew(C.LR_NEG, c.top(), p1, nil, p2, nil) }
func (c *Context) NegLvl(level uint64, p1, p2 *Poly)          { c.ew(C.LR_NEG, level, p1, nil, p2, nil) }
func (c *Context) Reduce(p1, p2 *Poly)                        { c.ew(C.LR_REDUCE, c.top(), p1, nil, p2, nil) }
func (c *Context) ReduceLvl(level uint64, p1, p2 *Poly)       { c.ew(C.LR_REDUCE, level, p1, nil, p2, nil) }

// Mod / AND / OR / XOR (ring/ring.go:146-184): word-level helpers outside the hot path, host loops over Coeffs.
func (c *Context) hostMap(p1, p2 *Poly, f func(uint64) uint64) {
	p1.hostView()
	for i := range c.Modulus {
		a, b := p1.Coeffs[i], p2.Coeffs[i]
		for j := uint64(0); j < c.N; j++ {
			b[j] = f(a[j])
		}
	}
	p2.hostWritten()
}

func (c *Context) Mod(p1 *Poly, m uint64, p2 *Poly) {
	u := BRedParams(m)
	c.hostMap(p1, p2, func(x uint64) uint64 { return BRedAdd(x, m, u) })
}
func (c *Context) AND(p1 *Poly, m uint64, p2 *Poly) { c.hostMap(p1, p2, func(x uint64) uint64 { return x & m }) }
func (c *Context) OR(p1 *Poly, m uint64, p2 *Poly)  { c.hostMap(p1, p2, func(x uint64) uint64 { return x | m }) }
func (c *Context) XOR(p1 *Poly, m uint64, p2 *Poly) { c.hostMap(p1, p2, func(x uint64) uint64 { return x ^ m }) }

// ring/ring.go:187-355
func (c *Context) MulCoeffs(p1, p2, p3 *Poly)            { c.ew(C.LR_MUL_COEFFS, c.top(), p1, p2, p3, nil) }
func (c *Context) MulCoeffsAndAdd(p1, p2, p3 *Poly)      { c.ew(C.LR_MUL_COEFFS_AND_ADD, c.top(), p1, p2, p3, nil) }
func (c *Context) MulCoeffsAndAddNoMod(p1, p2, p3 *Poly) { c.ew(C.LR_MUL_COEFFS_AND_ADD_NOMOD, c.top(), p1, p2, p3, nil) }
func (c *Context) MulCoeffsConstant(p1, p2, p3 *Poly)    { c.ew(C.LR_MUL_COEFFS_CONSTANT, c.top(), p1, p2, p3, nil) }
func (c *Context) MulCoeffsMontgomery(p1, p2, p3 *Poly)  { c.ew(C.LR_MUL_MONT, c.top(), p1, p2, p3, nil) }
func (c *Context) MulCoeffsMontgomeryLvl(level uint64, p1, p2, p3 *Poly) {
	c.ew(C.LR_MUL_MONT, level, p1, p2, p3, nil)
}
func (c *Context) MulCoeffsMontgomeryAndAdd(p1, p2, p3 *Poly) { c.ew(C.LR_MUL_MONT_AND_ADD, c.top(), p1, p2, p3, nil) }
func (c *Context) MulCoeffsMontgomeryAndAddLvl(level uint64, p1, p2, p3 *Poly) {
	c.ew(C.LR_MUL_MONT_AND_ADD, level, p1, p2, p3, nil)
}
func (c *Context) MulCoeffsMontgomeryAndAddNoMod(p1, p2, p3 *Poly) {
	c.ew(C.LR_MUL_MONT_AND_ADD_NOMOD, c.top(), p1, p2, p3, nil)
}
func (c *Context) MulCoeffsMontgomeryAndAddNoModLvl(level uint64, p1, p2, p3 *Poly) {
	c.ew(C.LR_MUL_MONT_AND_ADD_NOMOD, level, p1, p2, p3, nil)
}
func (c *Context) MulCoeffsMontgomeryConstantAndAddNoModLvl(level uint64, p1, p2, p3 *Poly) {
	c.ew(C.LR_MUL_MONT_CONSTANT_AND_ADD_NOMOD, level, p1, p2, p3, nil)
}
func (c *Context) MulCoeffsMontgomeryAndSub(p1, p2, p3 *Poly) { c.ew(C.LR_MUL_MONT_AND_SUB, c.top(), p1, p2, p3, nil) }
func (c *Context) MulCoeffsMontgomeryAndSubNoMod(p1, p2, p3 *Poly) {
	c.ew(C.LR_MUL_MONT_AND_SUB_NOMOD, c.top(), p1, p2, p3, nil)
}
func (c *Context) MulCoeffsMontgomeryConstant(p1, p2, p3 *Poly) { c.ew(C.LR_MUL_MONT_CONSTANT, c.top(), p1, p2, p3, nil) }

// MulPoly / MulPolyMontgomery (ring/ring.go:358,371): NTT both operands, multiply, InvNTT.
func (c *Context) MulPoly(p1, p2, p3 *Poly) {
	a, b := c.NewPoly(), c.NewPoly()
	c.NTT(p1, a)
	c.NTT(p2, b)
	c.MulCoeffs(a, b, p3)
	c.InvNTT(p3, p3)
}

func (c *Context) MulPolyMontgomery(p1, p2, p3 *Poly) {
	a, b := c.NewPoly(), c.NewPoly()
	c.NTT(p1, a)
	c.NTT(p2, b)
	c.MulCoeffsMontgomery(a, b, p3)
	c.InvNTT(p3, p3)
}

// MulPolyNaive / MulPolyNaiveMontgomery (ring/ring.go:383,413): the schoolbook negacyclic convolution the reference's tests
// compare the NTT product with; O(N^2) on the host.
func (c *Context) MulPolyNaive(p1, p2, p3 *Poly) {
	m := p1.CopyNew()
	c.MForm(m, m)
	c.MulPolyNaiveMontgomery(m, p2, p3)
}

func (c *Context) MulPolyNaiveMontgomery(p1, p2, p3 *Poly) {
	x, y := p1.CopyNew(), p2.CopyNew()
	n := c.N
	for l, q := range c.Modulus {
		a, b, acc := x.Coeffs[l], y.Coeffs[l], p3.Coeffs[l]
		qInv := c.mredParams[l]
		for j := range acc {
			acc[j] = 0
		}
		for i := uint64(0); i < n; i++ {
			for j := uint64(0); j < i; j++ { // wrapped terms come back negated
				acc[j] = CRed(acc[j]+(q-MRed(a[i], b[n-i+j], q, qInv)), q)
			}
			for j := i; j < n; j++ {
				acc[j] = CRed(acc[j]+MRed(a[i], b[j-i], q, qInv), q)
			}
		}
	}
	p3.hostWritten()
}

// AddScalar / SubScalar and their Bigint forms write into p1, not p2 (ring/ring.go:469,482,492,505:
// p1tmp, p2tmp := p1.Coeffs[i], p1.Coeffs[i]); harmless for the in-place calls of bfv/evaluator.go:457,459 and reproduced.
func (c *Context) AddScalar(p1 *Poly, scalar uint64, p2 *Poly) {
	c.ew(C.LR_ADD_SCALAR_LIMBS, c.top(), p1, nil, p1, c.repeat(scalar))
}
func (c *Context) SubScalar(p1 *Poly, scalar uint64, p2 *Poly) {
	c.ew(C.LR_SUB_SCALAR_LIMBS, c.top(), p1, nil, p1, c.repeat(scalar))
}
func (c *Context) AddScalarBigint(p1 *Poly, scalar *big.Int, p2 *Poly) {
	c.ew(C.LR_ADD_SCALAR_LIMBS, c.top(), p1, nil, p1, c.residues(scalar))
}
func (c *Context) SubScalarBigint(p1 *Poly, scalar *big.Int, p2 *Poly) {
	c.ew(C.LR_SUB_SCALAR_LIMBS, c.top(), p1, nil, p1, c.residues(scalar))
}

// ring/ring.go:513-656
func (c *Context) MulScalar(p1 *Poly, scalar uint64, p2 *Poly) {
	c.ew(C.LR_MUL_SCALAR, c.top(), p1, nil, p2, []uint64{scalar})
}
func (c *Context) MulScalarLvl(level uint64, p1 *Poly, scalar uint64, p2 *Poly) {
	c.ew(C.LR_MUL_SCALAR, level, p1, nil, p2, []uint64{scalar})
}
func (c *Context) MulScalarBigint(p1 *Poly, scalar *big.Int, p2 *Poly) {
	c.ew(C.LR_MUL_SCALAR_LIMBS, c.top(), p1, nil, p2, c.residues(scalar))
}
func (c *Context) MulScalarBigintLvl(level uint64, p1 *Poly, scalar *big.Int, p2 *Poly) {
	c.ew(C.LR_MUL_SCALAR_LIMBS, level, p1, nil, p2, c.residues(scalar))
}
func (c *Context) MForm(p1, p2 *Poly)                  { c.ew(C.LR_MFORM, c.top(), p1, nil, p2, nil) }
func (c *Context) MFormLvl(level uint64, p1, p2 *Poly) { c.ew(C.LR_MFORM, level, p1, nil, p2, nil) }
func (c *Context) InvMForm(p1, p2 *Poly)               { c.ew(C.LR_INV_MFORM, c.top(), p1, nil, p2, nil) }

func (c *Context) MulByPow2New(p1 *Poly, pow2 uint64) *Poly {
	p2 := c.NewPoly()
	c.MulByPow2(p1, pow2, p2)
	return p2
}
func (c *Context) MulByPow2(p1 *Poly, pow2 uint64, p2 *Poly) {
	c.ew(C.LR_MUL_BY_POW2, c.top(), p1, nil, p2, []uint64{pow2})
}
func (c *Context) MulByPow2Lvl(level uint64, p1 *Poly, pow2 uint64, p2 *Poly) {
	c.ew(C.LR_MUL_BY_POW2, level, p1, nil, p2, []uint64{pow2})
}

// MultByMonomial / MultByMonomialNew (ring/ring.go:656,663); p1 == p2 is allowed.
func (c *Context) MultByMonomial(p1 *Poly, monomialDeg uint64, p2 *Poly) {
	c.use(p1)
	c.want(p2)
	call(func() C.int { return C.lr_mult_by_monomial(c.h, p1.d, C.uint64_t(monomialDeg), p2.d) })
	done(p2)
}
// Shift (ring/ring.go:575): p2 = p1 rotated left by n positions.  The reference re-slices p2.Coeffs; here p2's device image is written.
func (c *Context) Shift(p1 *Poly, n uint64, p2 *Poly) {
	c.use(p1)
	c.want(p2)
	call(func() C.int { return C.lr_shift(c.h, p1.d, C.uint64_t(n), p2.d) })
	done(p2)
}

// Rotate (ring/ring.go:775): the reference writes into p1 whatever p2 is (`p1tmp, p2tmp := p1.Coeffs[i], p1.Coeffs[i]`, :791).
func (c *Context) Rotate(p1 *Poly, n uint64, p2 *Poly) {
	c.use(p1)
	call(func() C.int { return C.lr_rotate(c.h, p1.d, C.uint64_t(n)) })
	done(p1)
}

// Exp (ring/ring.go:440-464): as written upstream its observable effect is p1 <- NTT(p1) and, by its last statement, p2 <- InvNTT(p1)
// (the powers it computes in between are overwritten).  No caller in the module; kept so that the identifier exists.
func (c *Context) Exp(p1 *Poly, e uint64, p2 *Poly) {
	c.NTT(p1, p1)
	c.InvNTT(p1, p2)
}

func (c *Context) MultByMonomialNew(p1 *Poly, monomialDeg uint64) *Poly {
	p2 := c.NewPoly()
	c.MultByMonomial(p1, monomialDeg, p2)
	return p2
}

// MulByVectorMontgomery / ...AndAddNoMod (ring/ring.go:726,737): one host vector against every limb; host loops.
func (c *Context) MulByVectorMontgomery(p1 *Poly, vector []uint64, p2 *Poly) {
	p1.hostView()
	for i, q := range c.Modulus {
		a, b, qInv := p1.Coeffs[i], p2.Coeffs[i], c.mredParams[i]
		for j := uint64(0); j < c.N; j++ {
			b[j] = MRed(a[j], vector[j], q, qInv)
		}
	}
	p2.hostWritten()
}

func (c *Context) MulByVectorMontgomeryAndAddNoMod(p1 *Poly, vector []uint64, p2 *Poly) {
	p1.hostView()
	p2.hostView()
	for i, q := range c.Modulus {
		a, b, qInv := p1.Coeffs[i], p2.Coeffs[i], c.mredParams[i]
		for j := uint64(0); j < c.N; j++ {
			b[j] += MRed(a[j], vector[j], q, qInv)
		}
	}
	p2.hostWritten()
}

// BitReverse (ring/ring.go:749): bit-reversal permutation of the coefficients, in place when p1 == p2; host loop.
func (c *Context) BitReverse(p1, p2 *Poly) {
	p1.hostView()
	shift := uint(64 - (bits.Len64(c.N) - 1))
	for i := range c.Modulus {
		a, b := p1.Coeffs[i], p2.Coeffs[i]
		for j := uint64(0); j < c.N; j++ {
			k := bits.Reverse64(j) >> shift
			if p1 != p2 {
				b[k] = a[j]
			} else if j < k {
				b[j], b[k] = b[k], b[j]
			}
		}
	}
	p2.hostWritten()
}

// HalfScalarOp runs the element loop of the constant-by-ciphertext methods of ckks.Evaluator on the device (ckks/evaluator.go:429-445,
// 588-606, 712-730, 765-779, 814-828): for the limbs 0..level, p2[i][j] = OP(p1[i][j], lo[i]) for j < N/2 and OP(p1[i][j], hi[i])
// above, with op 0 = CRed(x + s), 1 = MRed(x, s), 2 = CRed(p2 + MRed(x, s)).  The overlay go/ckks/evaluator_device.go calls it where
// the upstream methods loop over Coeffs.
func (c *Context) HalfScalarOp(op int, level uint64, p1 *Poly, lo, hi []uint64, p2 *Poly) {
	c.use(p1)
	if op == 2 {
		c.use(p2)
	} else {
		c.want(p2)
	}
	plo := (*C.uint64_t)(unsafe.Pointer(&lo[0]))
	phi := (*C.uint64_t)(unsafe.Pointer(&hi[0]))
	call(func() C.int { return C.lr_half_scalar_op(c.h, C.int(op), C.int(level), p1.d, plo, phi, p2.d) })
	done(p2)
}
