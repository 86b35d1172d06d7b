package ring

// Host-side scalar primitives of ring/modular_reduction.go (the evaluators call them per coefficient on host
// slices, e.g. ring.MRed 15x and ring.CRed 10x in ckks/evaluator.go).  Formulas: SURVEY.md appendix A.1.  The device
// kernels use the same definitions (csrc/lr_arith.hpp); these never cross cgo.

import (
	"math/big"
	"math/bits"
)

// MRedParams: q^-1 mod 2^64 as q^(2^63 - 1) by square-and-multiply (modular_reduction.go:53).  The inverse is the
// POSITIVE one, which fixes the sign convention of MRed below.
func MRedParams(q uint64) (qInv uint64) {
	qInv = 1
	sq := q
	for i := 0; i < 63; i++ {
		qInv *= sq
		sq *= sq
	}
	return
}

// BRedParams: floor(2^128 / q) as {high word, low word} (modular_reduction.go:97).
func BRedParams(q uint64) []uint64 {
	u := new(big.Int).Lsh(big.NewInt(1), 128)
	u.Quo(u, new(big.Int).SetUint64(q))
	lo := new(big.Int).And(u, new(big.Int).SetUint64(^uint64(0))).Uint64()
	hi := new(big.Int).Rsh(u, 64).Uint64()
	return []uint64{hi, lo}
}

// MForm: a * 2^64 mod q with the Barrett constant u (modular_reduction.go:15).
func MForm(a, q uint64, u []uint64) (r uint64) {
	r = MFormConstant(a, q, u)
	if r >= q {
		r -= q
	}
	return
}

// MFormConstant: the same in [0, 2q) (:25).
func MFormConstant(a, q uint64, u []uint64) uint64 {
	top, _ := bits.Mul64(a, u[1])
	return -(a*u[0] + top) * q
}

// InvMForm: a * 2^-64 mod q (:34).
func InvMForm(a, q, qInv uint64) (r uint64) {
	r = InvMFormConstant(a, q, qInv)
	if r >= q {
		r -= q
	}
	return
}

// InvMFormConstant: in [0, 2q) (:44).
func InvMFormConstant(a, q, qInv uint64) uint64 {
	h, _ := bits.Mul64(a*qInv, q)
	return q - h
}

// MRed: x * y * 2^-64 mod q (:70).
func MRed(x, y, q, qInv uint64) (r uint64) {
	r = MRedConstant(x, y, q, qInv)
	if r >= q {
		r -= q
	}
	return
}

// MRedConstant: in [0, 2q) (:83).
func MRedConstant(x, y, q, qInv uint64) uint64 {
	hi, lo := bits.Mul64(x, y)
	h, _ := bits.Mul64(lo*qInv, q)
	return hi - h + q
}

// BRedAdd: x mod q for any 64-bit x (:112).
func BRedAdd(x, q uint64, u []uint64) (r uint64) {
	r = BRedAddConstant(x, q, u)
	if r >= q {
		r -= q
	}
	return
}

// BRedAddConstant: in [0, 2q) (:123).
func BRedAddConstant(x, q uint64, u []uint64) uint64 {
	est, _ := bits.Mul64(x, u[0])
	return x - est*q
}

// BRed: x * y mod q for arbitrary 64-bit x, y (:133).
func BRed(x, y, q uint64, u []uint64) (r uint64) {
	r = BRedConstant(x, y, q, u)
	if r >= q {
		r -= q
	}
	return
}

// BRedConstant: in [0, 2q) (:172).  Quotient estimate = the words of (x*y) * u above 2^128, with the reference's
// carry chain (the low x low product only contributes its high word; the carry of the second cross sum is kept,
// its sum is not).
func BRedConstant(x, y, q uint64, u []uint64) uint64 {
	phi, plo := bits.Mul64(x, y)
	llhi, _ := bits.Mul64(plo, u[1])
	c1hi, c1lo := bits.Mul64(plo, u[0])
	mid, carry := bits.Add64(c1lo, llhi, 0)
	acc := c1hi + carry
	c2hi, c2lo := bits.Mul64(phi, u[1])
	_, carry = bits.Add64(c2lo, mid, 0)
	est := phi*u[0] + acc + c2hi + carry
	return plo - est*q
}

// CRed: a mod q for a in [0, 2q) (:211).
func CRed(a, q uint64) uint64 {
	if a >= q {
		return a - q
	}
	return a
}
