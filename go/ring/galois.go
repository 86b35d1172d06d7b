package ring

// #include "lattigo_ring.h"
import "C"

import "unsafe"

// GaloisGen is the generator of the rotation group used by ckks/bfv (ring/ring_galois.go:9).
const GaloisGen uint64 = 5

// PermuteNTTIndex (ring/ring_galois.go:29): the gather index of the automorphism X -> X^(gen^power) in the NTT domain.
func PermuteNTTIndex(gen, power, N uint64) []uint64 {
	index := make([]uint64, N)
	check(C.lr_permute_ntt_index(C.uint64_t(gen), C.uint64_t(power), C.uint64_t(N), (*C.uint64_t)(unsafe.Pointer(&index[0]))))
	return index
}

// PermuteNTT (ring/ring_galois.go:55) as a method: the package-level function of the reference has no context to
// find the device through; polIn and polOut must differ, as in the reference.
func (c *Context) PermuteNTT(polIn *Poly, gen uint64, polOut *Poly) {
	in(polIn)
	check(C.lr_permute_ntt(c.h, C.int(len(polIn.Coeffs)-1), polIn.d, C.uint64_t(gen), polOut.d))
	out(polOut)
}

// Permute (ring/ring_galois.go:106), coefficient domain.
func (c *Context) Permute(polIn *Poly, gen uint64, polOut *Poly) {
	in(polIn)
	check(C.lr_permute(c.h, polIn.d, C.uint64_t(gen), polOut.d))
	out(polOut)
}
