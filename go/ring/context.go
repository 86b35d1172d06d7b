package ring

// #include "lattigo_ring.h"
import "C"

import (
	"bytes"
	"encoding/gob"
	"errors"
	"math/big"
	"math/bits"
	"runtime"
	"unsafe"
)

// Context mirrors ring.Context (ring/ring_context.go:18-51): same exported fields, same unexported fields (the sampler
// files kept from upstream read mask and the ternary matrices), plus the device handle.  The host-format tables
// (Montgomery form, bit-reversed order) are the library's copies of what GenNTTParams computes in the reference.
type Context struct {
	N       uint64
	Modulus []uint64

	mask      []uint64
	allowsNTT bool

	ModulusBigint *big.Int

	bredParams [][]uint64
	mredParams []uint64

	rescaleParams [][]uint64

	matrixTernary           [][]uint64
	matrixTernaryMontgomery [][]uint64

	psiMont    []uint64
	psiInvMont []uint64

	nttPsi    [][]uint64
	nttPsiInv [][]uint64
	nttNInv   []uint64

	h      *C.lr_context
	device int
}

// DefaultDevice is the GPU new contexts bind to; one process per GPU sets it once at start-up (INTEGRATION.md).
var DefaultDevice = 0

// NewContext (ring/ring_context.go:54).
func NewContext() *Context { return &Context{device: DefaultDevice} }

// NewContextWithParams = NewContext + SetParameters + GenNTTParams (ring/ring_context.go:60-66).
func NewContextWithParams(N uint64, Moduli []uint64) (*Context, error) {
	c := NewContext()
	c.SetParameters(N, Moduli)
	return c, c.GenNTTParams()
}

// NewContextWithParamsOnDevice binds the context (and everything created through it) to one GPU.
func NewContextWithParamsOnDevice(N uint64, Moduli []uint64, device int) (*Context, error) {
	c := &Context{device: device}
	c.SetParameters(N, Moduli)
	return c, c.GenNTTParams()
}

// SetParameters (ring/ring_context.go:68-127): everything that does not need the moduli to be NTT-friendly.
func (c *Context) SetParameters(N uint64, Modulus []uint64) {
	if (N&(N-1)) != 0 && N != 0 {
		panic("invalid ring degree (must be a power of 2)") // :71-73
	}
	c.allowsNTT = false
	c.N = N
	L := len(Modulus)
	c.Modulus = append([]uint64{}, Modulus...)
	c.mask = make([]uint64, L)
	c.ModulusBigint = big.NewInt(1)
	c.bredParams = make([][]uint64, L)
	c.mredParams = make([]uint64, L)
	c.matrixTernary = make([][]uint64, L)
	c.matrixTernaryMontgomery = make([][]uint64, L)
	for i, qi := range c.Modulus {
		c.mask[i] = (1 << uint64(bits.Len64(qi))) - 1 // :84
		c.ModulusBigint.Mul(c.ModulusBigint, new(big.Int).SetUint64(qi))
		c.bredParams[i] = BRedParams(qi)
		if (qi&(qi-1)) != 0 && qi != 0 { // :104-106
			c.mredParams[i] = MRedParams(qi)
		}
		c.matrixTernary[i] = []uint64{0, 1, qi - 1}
		c.matrixTernaryMontgomery[i] = []uint64{0, MForm(1, qi, c.bredParams[i]), MForm(qi-1, qi, c.bredParams[i])}
	}
}

// GenNTTParams (ring/ring_context.go:129-209): checks that every modulus is a prime congruent to 1 mod 2N, computes the
// psi tables -- here inside lr_context_create, with the reference's primitiveRoot search -- uploads them, and keeps the
// host-format copies for the getters.
func (c *Context) GenNTTParams() error {
	if c.N == 0 || len(c.Modulus) == 0 {
		panic("error : invalid context parameters (missing)") // :131-133
	}
	if c.h != nil {
		C.lr_context_destroy(c.h)
		c.h = nil
	}
	rc, msg := status(func() C.int {
		return C.lr_context_create_ex(C.uint64_t(c.N), (*C.uint64_t)(unsafe.Pointer(&c.Modulus[0])), C.int(len(c.Modulus)), C.int(c.device), DefaultOptions.ptr(), &c.h)
	})
	switch rc {
	case C.LR_OK:
	case C.LR_ERR_NOT_NTT_FRIENDLY:
		c.allowsNTT = false
		return errors.New("warning : provided modulus does not allow NTT") // :141-146
	case C.LR_ERR_INVALID_DEGREE:
		panic("invalid ring degree (must be a power of 2)")
	default:
		return errors.New(msg)
	}
	runtime.SetFinalizer(c, func(c *Context) { C.lr_context_destroy(c.h) })
	L, n := len(c.Modulus), int(c.N)
	c.rescaleParams = make([][]uint64, L-1) // :148-158: rescaleParams[j-1][i], i < j
	flat := c.table(C.LR_TAB_RESCALE, L*L)
	for j := 1; j < L; j++ {
		c.rescaleParams[j-1] = flat[(j-1)*L : (j-1)*L+j]
	}
	c.psiMont = c.table(C.LR_TAB_PSI_MONT, L)
	c.psiInvMont = c.table(C.LR_TAB_PSI_INV_MONT, L)
	c.nttNInv = c.table(C.LR_TAB_NTT_N_INV, L)
	c.nttPsi = rows(c.table(C.LR_TAB_NTT_PSI, L*n), L, n)
	c.nttPsiInv = rows(c.table(C.LR_TAB_NTT_PSI_INV, L*n), L, n)
	c.allowsNTT = true
	return nil
}

func (c *Context) table(which C.int, n int) []uint64 {
	out := make([]uint64, n)
	call(func() C.int { return C.lr_context_get_table(c.h, which, (*C.uint64_t)(unsafe.Pointer(&out[0])), C.size_t(n)) })
	return out
}

func rows(flat []uint64, count, per int) [][]uint64 {
	out := make([][]uint64, count)
	for i := range out {
		out[i] = flat[i*per : (i+1)*per]
	}
	return out
}

// smallContext / MarshalBinary / UnmarshalBinary (ring/ring_context.go:211-245): N and the moduli through encoding/gob.
type smallContext struct {
	N       uint64
	Modulus []uint64
}

func (c *Context) MarshalBinary() ([]byte, error) {
	var buf bytes.Buffer
	if err := gob.NewEncoder(&buf).Encode(smallContext{c.N, c.Modulus}); err != nil {
		return nil, err
	}
	return buf.Bytes(), nil
}

func (c *Context) UnmarshalBinary(data []byte) error {
	var p smallContext
	if err := gob.NewDecoder(bytes.NewReader(data)).Decode(&p); err != nil {
		return err
	}
	c.SetParameters(p.N, p.Modulus)
	c.GenNTTParams()
	return nil
}

// Getters (ring/ring_context.go:248-285).
func (c *Context) AllowsNTT() bool           { return c.allowsNTT }
func (c *Context) GetBredParams() [][]uint64 { return c.bredParams }
func (c *Context) GetMredParams() []uint64   { return c.mredParams }
func (c *Context) GetPsi() []uint64          { return c.psiMont }
func (c *Context) GetPsiInv() []uint64       { return c.psiInvMont }
func (c *Context) GetNttPsi() [][]uint64     { return c.nttPsi }
func (c *Context) GetNttPsiInv() [][]uint64  { return c.nttPsiInv }
func (c *Context) GetNttNInv() []uint64      { return c.nttNInv }

// NewPoly / NewPolyLvl (ring/ring_context.go:288,300): host storage only; the device image is created on first use.
func (c *Context) NewPoly() *Poly { return NewPoly(c.N, uint64(len(c.Modulus))) }

func (c *Context) NewPolyLvl(level uint64) *Poly { return NewPoly(c.N, level+1) }

// SetCoefficientsInt64 / Uint64 / String / Bigint(Lvl), PolyToString, PolyToBigint, Equal, EqualLvl
// (ring/ring_context.go:312-456): host-side conversions between residues and integers.  They work on Coeffs; a
// device-resident polynomial is synchronised first and pushed back afterwards.
func (c *Context) SetCoefficientsInt64(coeffs []int64, p1 *Poly) {
	for i, v := range coeffs {
		for j, qi := range c.Modulus {
			p1.Coeffs[j][i] = CRed(uint64(v%int64(qi)+int64(qi)), qi)
		}
	}
	p1.hostWritten()
}

func (c *Context) SetCoefficientsUint64(coeffs []uint64, p1 *Poly) {
	for i, v := range coeffs {
		for j, qi := range c.Modulus {
			p1.Coeffs[j][i] = v % qi
		}
	}
	p1.hostWritten()
}

func (c *Context) SetCoefficientsString(coeffs []string, p1 *Poly) {
	vals := make([]*big.Int, len(coeffs))
	for i, s := range coeffs {
		v, ok := new(big.Int).SetString(s, 10)
		if !ok {
			panic("SetCoefficientsString: not a base-10 integer")
		}
		vals[i] = v
	}
	c.SetCoefficientsBigintLvl(uint64(len(c.Modulus)-1), vals, p1)
}

func (c *Context) SetCoefficientsBigint(coeffs []*big.Int, p1 *Poly) {
	c.SetCoefficientsBigintLvl(uint64(len(c.Modulus)-1), coeffs, p1)
}

func (c *Context) SetCoefficientsBigintLvl(level uint64, coeffs []*big.Int, p1 *Poly) {
	t := new(big.Int)
	for i := uint64(0); i <= level; i++ {
		qi := new(big.Int).SetUint64(c.Modulus[i])
		for j, v := range coeffs {
			p1.Coeffs[i][j] = t.Mod(v, qi).Uint64()
		}
	}
	p1.hostWritten()
}

func (c *Context) PolyToString(p1 *Poly) []string {
	vals := make([]*big.Int, c.N)
	c.PolyToBigint(p1, vals)
	out := make([]string, c.N)
	for i := range out {
		out[i] = vals[i].String()
	}
	return out
}

// PolyToBigint (ring/ring_context.go:384-423): CRT reconstruction of every coefficient in [0, Q).
func (c *Context) PolyToBigint(p1 *Poly, coeffsBigint []*big.Int) {
	p1.hostView()
	level := len(p1.Coeffs) - 1
	Q := big.NewInt(1)
	for i := 0; i <= level; i++ {
		Q.Mul(Q, new(big.Int).SetUint64(c.Modulus[i]))
	}
	crt := make([]*big.Int, level+1)
	for i := 0; i <= level; i++ {
		qi := new(big.Int).SetUint64(c.Modulus[i])
		rest := new(big.Int).Quo(Q, qi)
		inv := new(big.Int).ModInverse(rest, qi)
		crt[i] = rest.Mul(rest, inv)
	}
	t := new(big.Int)
	for x := uint64(0); x < c.N; x++ {
		acc := new(big.Int)
		for i := 0; i <= level; i++ {
			acc.Add(acc, t.Mul(new(big.Int).SetUint64(p1.Coeffs[i][x]), crt[i]))
		}
		coeffsBigint[x] = acc.Mod(acc, Q)
	}
}

// Equal / EqualLvl (ring/ring_context.go:426-456): both operands are reduced first, then compared limb by limb.
func (c *Context) Equal(p1, p2 *Poly) bool { return c.EqualLvl(uint64(len(c.Modulus)-1), p1, p2) }

func (c *Context) EqualLvl(level uint64, p1, p2 *Poly) bool {
	c.ReduceLvl(level, p1, p1)
	c.ReduceLvl(level, p2, p2)
	p1.hostView()
	p2.hostView()
	for i := uint64(0); i <= level; i++ {
		for j := uint64(0); j < c.N; j++ {
			if p1.Coeffs[i][j] != p2.Coeffs[i][j] {
				return false
			}
		}
	}
	return true
}

// Sync waits for the device work queued by this context's handles.
func (c *Context) Sync() { call(func() C.int { return C.lr_context_sync(c.h) }) }

// SetStream runs this context's launches on a caller-owned hipStream_t (nil: the device's shared stream).  Handles built over
// two contexts need the same stream on both (include/lattigo_ring.h, lr_context_set_stream).
func (c *Context) SetStream(hipStream unsafe.Pointer) {
	call(func() C.int { return C.lr_context_set_stream(c.h, hipStream) })
}
