package ring

// #include "lattigo_ring.h"
import "C"

import (
	"math/big"
	"unsafe"
)

// ew is the one forwarding point of the coefficient-wise family (ring/ring.go): op on limbs [0, level].
func (c *Context) ew(op C.int, level uint64, a, b, o *Poly, scalars []uint64) {
	in(a, b)
	if readsOut(op) {
		in(o)
	}
	var bd *C.lr_poly
	if b != nil {
		bd = b.d
	}
	var sp *C.uint64_t
	if scalars != nil {
		sp = (*C.uint64_t)(unsafe.Pointer(&scalars[0]))
	}
	check(C.lr_ewise(c.h, op, C.int(level), a.d, bd, o.d, sp))
	out(o)
}

func readsOut(op C.int) bool {
	switch op {
	case C.LR_MUL_COEFFS_AND_ADD, C.LR_MUL_COEFFS_AND_ADD_NOMOD, C.LR_MUL_MONT_AND_ADD, C.LR_MUL_MONT_AND_ADD_NOMOD,
		C.LR_MUL_MONT_CONSTANT_AND_ADD_NOMOD, C.LR_MUL_MONT_AND_SUB, C.LR_MUL_MONT_AND_SUB_NOMOD:
		return true
	}
	return false
}

func (c *Context) top() uint64 { return uint64(len(c.Modulus) - 1) }

// residues of a big scalar, one per modulus (what MulScalarBigint / AddScalarBigint compute per limb).
func (c *Context) residues(s *big.Int) []uint64 {
	out := make([]uint64, len(c.Modulus))
	t := new(big.Int)
	for i, q := range c.Modulus {
		out[i] = t.Mod(s, new(big.Int).SetUint64(q)).Uint64()
	}
	return out
}

// ring/ring.go:10-143
func (c *Context) Add(p1, p2, p3 *Poly)                        { c.ew(C.LR_ADD, c.top(), p1, p2, p3, nil) }
func (c *Context) AddLvl(level uint64, p1, p2, p3 *Poly)       { c.ew(C.LR_ADD, level, p1, p2, p3, nil) }
func (c *Context) AddNoMod(p1, p2, p3 *Poly)                   { c.ew(C.LR_ADD_NOMOD, c.top(), p1, p2, p3, nil) }
func (c *Context) AddNoModLvl(level uint64, p1, p2, p3 *Poly)  { c.ew(C.LR_ADD_NOMOD, level, p1, p2, p3, nil) }
func (c *Context) Sub(p1, p2, p3 *Poly)                        { c.ew(C.LR_SUB, c.top(), p1, p2, p3, nil) }
func (c *Context) SubLvl(level uint64, p1, p2, p3 *Poly)       { c.ew(C.LR_SUB, level, p1, p2, p3, nil) }
func (c *Context) SubNoMod(p1, p2, p3 *Poly)                   { c.ew(C.LR_SUB_NOMOD, c.top(), p1, p2, p3, nil) }
func (c *Context) SubNoModLvl(level uint64, p1, p2, p3 *Poly)  { c.ew(C.LR_SUB_NOMOD, level, p1, p2, p3, nil) }
func (c *Context) Neg(p1, p2 *Poly)                            { c.ew(C.LR_NEG, c.top(), p1, nil, p2, nil) }
func (c *Context) NegLvl(level uint64, p1, p2 *Poly)           { c.ew(C.LR_NEG, level, p1, nil, p2, nil) }
func (c *Context) Reduce(p1, p2 *Poly)                         { c.ew(C.LR_REDUCE, c.top(), p1, nil, p2, nil) }
func (c *Context) ReduceLvl(level uint64, p1, p2 *Poly)        { c.ew(C.LR_REDUCE, level, p1, nil, p2, nil) }

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

// ring/ring.go:469-656
func (c *Context) MForm(p1, p2 *Poly)                  { c.ew(C.LR_MFORM, c.top(), p1, nil, p2, nil) }
func (c *Context) MFormLvl(level uint64, p1, p2 *Poly) { c.ew(C.LR_MFORM, level, p1, nil, p2, nil) }
func (c *Context) InvMForm(p1, p2 *Poly)               { c.ew(C.LR_INV_MFORM, c.top(), p1, nil, p2, nil) }
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

// AddScalarBigint / SubScalarBigint write into p1, not p2 (ring/ring.go:482,505: p1tmp, p2tmp := p1.Coeffs[i], p1.Coeffs[i]);
// harmless for the in-place calls of bfv/evaluator.go:457,459 and reproduced as such.
func (c *Context) AddScalarBigint(p1 *Poly, scalar *big.Int, p2 *Poly) {
	c.ew(C.LR_ADD_SCALAR_LIMBS, c.top(), p1, nil, p1, c.residues(scalar))
}
func (c *Context) SubScalarBigint(p1 *Poly, scalar *big.Int, p2 *Poly) {
	c.ew(C.LR_SUB_SCALAR_LIMBS, c.top(), p1, nil, p1, c.residues(scalar))
}
func (c *Context) MulByPow2New(p1 *Poly, pow2 uint64) *Poly {
	p2 := c.NewPoly()
	c.MulByPow2(p1, pow2, p2)
	return p2
}
func (c *Context) MulByPow2(p1 *Poly, pow2 uint64, p2 *Poly) {
	c.ew(C.LR_MUL_BY_POW2, c.top(), p1, nil, p2, []uint64{pow2})
}

// MultByMonomial (ring/ring.go:663); p1 and p2 must differ here (the reference goes through a temporary).
func (c *Context) MultByMonomial(p1 *Poly, monomialDeg uint64, p2 *Poly) {
	in(p1)
	check(C.lr_mult_by_monomial(c.h, p1.d, C.uint64_t(monomialDeg), p2.d))
	out(p2)
}

// Copy / CopyLvl (ring/ring_object.go:85,98).
func (c *Context) Copy(p0, p1 *Poly)                  { c.ew(C.LR_COPY, c.top(), p0, nil, p1, nil) }
func (c *Context) CopyLvl(level uint64, p0, p1 *Poly) { c.ew(C.LR_COPY, level, p0, nil, p1, nil) }
