package ring

// #include "lattigo_ring.h"
import "C"

import "unsafe"

// GenGaloisParams (ring/ring_galois.go:9): the powers gen^0 .. gen^(n/2 - 1) modulo 2n.
func GenGaloisParams(n, gen uint64) (galElRotCol []uint64) {
	m, mask := n>>1, (n<<1)-1
	galElRotCol = make([]uint64, m)
	galElRotCol[0] = 1
	for i := uint64(1); i < m; i++ {
		galElRotCol[i] = (galElRotCol[i-1] * gen) & mask
	}
	return
}

// PermuteNTTIndex (ring/ring_galois.go:29): the gather index of the automorphism X -> X^(gen^power) in the NTT domain.
func PermuteNTTIndex(gen, power, N uint64) []uint64 {
	index := make([]uint64, N)
	call(func() C.int {
		return C.lr_permute_ntt_index(C.uint64_t(gen), C.uint64_t(power), C.uint64_t(N), (*C.uint64_t)(unsafe.Pointer(&index[0])))
	})
	return index
}

// PermuteNTT (ring/ring_galois.go:55) and PermuteNTTWithIndex (:89) are package-level in the reference: no Context, hence no
// device, is at hand.  They are plain gathers over Coeffs on the host -- "Careful, not inplace!" as in the reference.  On the
// device the same permutation is Context.PermuteNTTLvl below and, fused with the key switch, CkksPlan.PermuteNTT.
func PermuteNTT(polIn *Poly, gen uint64, polOut *Poly) {
	N := uint64(len(polIn.Coeffs[0]))
	PermuteNTTWithIndex(polIn, PermuteNTTIndex(gen, 1, N), polOut)
}

func PermuteNTTWithIndex(polIn *Poly, index []uint64, polOut *Poly) {
	polIn.hostView()
	for j := range polIn.Coeffs {
		src, dst := polIn.Coeffs[j], polOut.Coeffs[j]
		for i, k := range index {
			dst[i] = src[k]
		}
	}
	polOut.hostWritten()
}

// PermuteNTTLvl: the device form of PermuteNTT on limbs 0..level (lr_permute_ntt); gen is the Galois element itself.
func (c *Context) PermuteNTTLvl(level uint64, polIn *Poly, gen uint64, polOut *Poly) {
	c.use(polIn)
	c.want(polOut)
	call(func() C.int { return C.lr_permute_ntt(c.h, C.int(level), polIn.d, C.uint64_t(gen), polOut.d) })
	done(polOut)
}

// Permute (ring/ring_galois.go:106), coefficient domain, all limbs of the context; not in place.
func (c *Context) Permute(polIn *Poly, gen uint64, polOut *Poly) {
	c.use(polIn)
	c.want(polOut)
	call(func() C.int { return C.lr_permute(c.h, polIn.d, C.uint64_t(gen), polOut.d) })
	done(polOut)
}
