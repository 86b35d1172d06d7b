package ring

// #include <stdlib.h>
// #include "lattigo_ring.h"
import "C"

import (
	"crypto/rand"
	"encoding/binary"
	"errors"
	"runtime"
	"unsafe"
)

// Poly mirrors ring.Poly (ring/ring_object.go:11-13): Coeffs is the host view; d is the device image, created the
// first time a Context method touches the polynomial.
type Poly struct {
	Coeffs [][]uint64

	d        *C.lr_poly
	dLimbs   int  // limbs the device image was allocated with
	resident bool // device-resident mode: Coeffs is stale until Sync / Unpin
}

// NewPoly (ring/ring_object.go:16): N coefficients under nbModuli moduli, all zero.
func NewPoly(N, nbModuli uint64) *Poly {
	p := &Poly{Coeffs: make([][]uint64, nbModuli)}
	for i := range p.Coeffs {
		p.Coeffs[i] = make([]uint64, N)
	}
	return p
}

// NewPolyUniform (ring/ring_object.go:26): every word drawn from crypto/rand, NOT reduced modulo anything.
func NewPolyUniform(N, nbModuli uint64) *Poly {
	p := NewPoly(N, nbModuli)
	buf := make([]byte, N<<3)
	for i := range p.Coeffs {
		if _, err := rand.Read(buf); err != nil {
			panic("crypto rand error")
		}
		for j := range p.Coeffs[i] {
			p.Coeffs[i][j] = binary.BigEndian.Uint64(buf[j<<3 : (j+1)<<3])
		}
	}
	return p
}

func (p *Poly) GetDegree() int    { return len(p.Coeffs[0]) } // ring/ring_object.go:50
func (p *Poly) GetLenModuli() int { return len(p.Coeffs) }    // :55

// bind creates the device image (once).  It is sized by cap(Coeffs): a rescale re-slices Coeffs (ring_scaling.go:33) and
// the image keeps the original stride, like lr_poly_set_limbs documents.
func (c *Context) bind(ps ...*Poly) {
	for _, p := range ps {
		if p == nil || p.d != nil {
			continue
		}
		limbs := cap(p.Coeffs)
		q := p
		call(func() C.int { return C.lr_poly_alloc(c.h, C.int(limbs), 1, &q.d) })
		p.dLimbs = limbs
		runtime.SetFinalizer(p, func(x *Poly) { C.lr_poly_free(x.d) })
	}
}

// Host <-> device copies go one limb at a time: a Go pointer may be passed to C for the duration of a call, but it may not be
// stored in C memory (an array of limb pointers), and runtime.Pinner -- which would allow that -- is go 1.21 while the reference
// module declares go 1.13 (/go.mod:3).  Each Coeffs[i] is a []uint64 (no Go pointers inside), so &Coeffs[i][0] may cross as it is.
func (p *Poly) uploadTo(d *C.lr_poly, batchIndex int) {
	for i := range p.Coeffs {
		limb := C.int(i)
		src := (*C.uint64_t)(unsafe.Pointer(&p.Coeffs[i][0]))
		call(func() C.int { return C.lr_poly_upload_limb(d, C.int(batchIndex), limb, src) })
	}
}

func (p *Poly) upload() {
	call(func() C.int { return C.lr_poly_set_limbs(p.d, C.int(len(p.Coeffs))) })
	p.uploadTo(p.d, 0)
}

func (p *Poly) download() {
	for i := range p.Coeffs {
		limb := C.int(i)
		dst := (*C.uint64_t)(unsafe.Pointer(&p.Coeffs[i][0]))
		call(func() C.int { return C.lr_poly_download_limb(p.d, 0, limb, dst) })
	}
}

// Pin makes the device image authoritative: methods stop copying this polynomial across PCIe.
func (p *Poly) Pin(c *Context) *Poly {
	if !p.resident {
		c.bind(p)
		p.upload()
		p.resident = true
	}
	return p
}

// Sync refreshes Coeffs from the device image of a resident polynomial; Unpin also leaves resident mode.
func (p *Poly) Sync() {
	if p.resident {
		p.download()
	}
}
func (p *Poly) Unpin() { p.Sync(); p.resident = false }

// HostView / HostWritten are what the evaluator overlays (go/ckks/evaluator_device.go, go/bfv/evaluator_device.go) put around the
// upstream loops that index Coeffs directly: HostView before a loop reads a possibly resident polynomial, HostWritten after a loop
// has written one.  Both are no-ops for polynomials that are not resident.
func (p *Poly) HostView()    { p.hostView() }
func (p *Poly) HostWritten() { p.hostWritten() }

// hostView / hostWritten bracket host-side code that reads / has written Coeffs of a possibly resident polynomial.
func (p *Poly) hostView() { p.Sync() }
func (p *Poly) hostWritten() {
	if p.resident {
		p.upload()
	}
}

// use / done bracket one method call on the literal drop-in path: inputs are bound and uploaded, outputs are
// bound before the call and downloaded after it.
func (c *Context) use(ps ...*Poly) {
	c.bind(ps...)
	for _, p := range ps {
		if p != nil && !p.resident {
			p.upload()
		}
	}
}
func (c *Context) want(ps ...*Poly) {
	c.bind(ps...)
	for _, p := range ps {
		if p != nil {
			call(func() C.int { return C.lr_poly_set_limbs(p.d, C.int(len(p.Coeffs))) })
		}
	}
}
func done(ps ...*Poly) {
	for _, p := range ps {
		if p != nil && !p.resident {
			p.download()
		}
	}
}

// Zero (ring/ring_object.go:60).
func (p *Poly) Zero() {
	for i := range p.Coeffs {
		for j := range p.Coeffs[i] {
			p.Coeffs[i][j] = 0
		}
	}
	if p.d != nil {
		call(func() C.int { return C.lr_poly_zero(p.d) })
	}
}

// CopyNew (ring/ring_object.go:67), Poly.Copy (:109), SetCoefficients (:122), GetCoefficients (:137): host copies.
func (p *Poly) CopyNew() *Poly {
	p.hostView()
	q := &Poly{Coeffs: make([][]uint64, len(p.Coeffs))}
	for i := range p.Coeffs {
		q.Coeffs[i] = append([]uint64{}, p.Coeffs[i]...)
	}
	return q
}

func (p *Poly) Copy(p1 *Poly) {
	if p == p1 {
		return
	}
	p1.hostView()
	for i := range p1.Coeffs {
		copy(p.Coeffs[i], p1.Coeffs[i])
	}
	p.hostWritten()
}

func (p *Poly) SetCoefficients(coeffs [][]uint64) {
	for i := range coeffs {
		copy(p.Coeffs[i], coeffs[i])
	}
	p.hostWritten()
}

func (p *Poly) GetCoefficients() [][]uint64 {
	p.hostView()
	out := make([][]uint64, len(p.Coeffs))
	for i := range p.Coeffs {
		out[i] = append([]uint64{}, p.Coeffs[i]...)
	}
	return out
}

// Context.Copy / CopyLvl (ring/ring_object.go:85,98).
func (c *Context) Copy(p0, p1 *Poly) { c.CopyLvl(uint64(len(c.Modulus)-1), p0, p1) }

func (c *Context) CopyLvl(level uint64, p0, p1 *Poly) {
	if p0 != p1 {
		c.ew(C.LR_COPY, level, p0, nil, p1, nil)
	}
}

// Wire format (ring/ring_object.go:146-270): byte 0 = log2 N, byte 1 = number of moduli, then limb-major big-endian
// words.  MarshalBinary / UnmarshalBinary of a polynomial that has a device image are produced and consumed by the
// device (the byte swap runs there); the slice-level helpers are host loops.
func (p *Poly) GetDataLen(WithMetadata bool) uint64 { // :178
	n := uint64(len(p.Coeffs)*len(p.Coeffs[0])) << 3
	if WithMetadata {
		n += 2
	}
	return n
}

func WriteCoeffsTo(pointer, N, numberModuli uint64, coeffs [][]uint64, data []byte) (uint64, error) { // :146
	for i := uint64(0); i < numberModuli; i++ {
		for j := uint64(0); j < N; j++ {
			v := coeffs[i][j]
			for k := 0; k < 8; k++ {
				data[pointer+uint64(k)] = byte(v >> uint(56-8*k))
			}
			pointer += 8
		}
	}
	return pointer, nil
}

func DecodeCoeffs(pointer, N, numberModuli uint64, coeffs [][]uint64, data []byte) (uint64, error) { // :197
	for i := uint64(0); i < numberModuli; i++ {
		for j := uint64(0); j < N; j++ {
			var v uint64
			for k := 0; k < 8; k++ {
				v = v<<8 | uint64(data[pointer+uint64(k)])
			}
			coeffs[i][j] = v
			pointer += 8
		}
	}
	return pointer, nil
}

func DecodeCoeffsNew(pointer, N, numberModuli uint64, coeffs [][]uint64, data []byte) (uint64, error) { // :209
	for i := uint64(0); i < numberModuli; i++ {
		coeffs[i] = make([]uint64, N)
	}
	return DecodeCoeffs(pointer, N, numberModuli, coeffs, data)
}

func (p *Poly) WriteTo(data []byte) (uint64, error) { // :159
	p.hostView()
	N, L := uint64(len(p.Coeffs[0])), uint64(len(p.Coeffs))
	if uint64(len(data)) < p.GetDataLen(true) {
		return 0, errors.New("Data array is too small to write ring.Poly")
	}
	data[0] = uint8(bitLen(N) - 1)
	data[1] = uint8(L)
	return WriteCoeffsTo(2, N, L, p.Coeffs, data)
}

func (p *Poly) WriteCoeffs(data []byte) (uint64, error) { // :172
	p.hostView()
	return WriteCoeffsTo(0, uint64(len(p.Coeffs[0])), uint64(len(p.Coeffs)), p.Coeffs, data)
}

func bitLen(x uint64) (n uint64) {
	for ; x != 0; x >>= 1 {
		n++
	}
	return
}

func (p *Poly) MarshalBinary() ([]byte, error) { // :222
	data := make([]byte, p.GetDataLen(true))
	if p.d == nil {
		_, err := p.WriteTo(data)
		return data, err
	}
	if !p.resident {
		p.upload()
	}
	var n C.size_t
	err := callErr(func() C.int {
		return C.lr_poly_marshal(p.d, 0, (*C.uint8_t)(unsafe.Pointer(&data[0])), C.size_t(len(data)), &n)
	})
	return data[:int(n)], err
}

func (p *Poly) UnmarshalBinary(data []byte) error { // :252
	N := uint64(1) << data[0]
	L := uint64(data[1])
	if uint64(len(data)-2) != (N*L)<<3 {
		return errors.New("error : invalid polynomial encoding") // :262-264
	}
	if p.Coeffs == nil || uint64(len(p.Coeffs)) != L {
		p.Coeffs = make([][]uint64, L)
		for i := range p.Coeffs {
			p.Coeffs[i] = make([]uint64, N)
		}
	}
	if p.d == nil || !p.resident {
		_, err := DecodeCoeffs(2, N, L, p.Coeffs, data)
		return err
	}
	return callErr(func() C.int {
		return C.lr_poly_unmarshal(p.d, 0, (*C.uint8_t)(unsafe.Pointer(&data[0])), C.size_t(len(data)))
	})
}

func (p *Poly) DecodePolyNew(data []byte) (uint64, error) { // :232
	N := uint64(1) << data[0]
	L := uint64(data[1])
	p.Coeffs = make([][]uint64, L)
	return DecodeCoeffsNew(2, N, L, p.Coeffs, data)
}
