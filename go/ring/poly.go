package ring

// #include <stdlib.h>
// #include "lattigo_ring.h"
import "C"

import (
	"runtime"
	"unsafe"
)

// Poly mirrors ring.Poly (ring/ring_object.go:11-13): Coeffs is the host view, d the device image.
type Poly struct {
	Coeffs [][]uint64

	d        *C.lr_poly
	resident bool // device-resident mode: Coeffs is stale until Sync / Unpin
}

func (p *Poly) GetDegree() int    { return len(p.Coeffs[0]) } // ring/ring_object.go:50
func (p *Poly) GetLenModuli() int { return len(p.Coeffs) }    // :55

// limbPtrs builds the C array of per-limb pointers.  A Go pointer passed to C may not point at memory that
// holds Go pointers, so Coeffs itself cannot cross: the array lives in C memory and each limb is pinned.
func (p *Poly) limbPtrs(pin *runtime.Pinner) (**C.uint64_t, func()) {
	n := len(p.Coeffs)
	raw := C.malloc(C.size_t(n) * C.size_t(unsafe.Sizeof(uintptr(0))))
	arr := unsafe.Slice((**C.uint64_t)(raw), n)
	for i := range p.Coeffs {
		pin.Pin(&p.Coeffs[i][0])
		arr[i] = (*C.uint64_t)(unsafe.Pointer(&p.Coeffs[i][0]))
	}
	return (**C.uint64_t)(raw), func() { C.free(raw) }
}

func (p *Poly) upload() {
	var pin runtime.Pinner
	defer pin.Unpin()
	ptrs, free := p.limbPtrs(&pin)
	defer free()
	check(C.lr_poly_set_limbs(p.d, C.int(len(p.Coeffs))))
	check(C.lr_poly_upload(p.d, 0, ptrs, C.int(len(p.Coeffs))))
}

func (p *Poly) download() {
	var pin runtime.Pinner
	defer pin.Unpin()
	ptrs, free := p.limbPtrs(&pin)
	defer free()
	check(C.lr_poly_download(p.d, 0, ptrs, C.int(len(p.Coeffs))))
}

// Pin makes the device image authoritative: methods stop copying this polynomial across PCIe.
func (p *Poly) Pin() *Poly {
	if !p.resident {
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

// in / out bracket one method call on the literal drop-in path.
func in(ps ...*Poly) {
	for _, p := range ps {
		if p != nil && !p.resident {
			p.upload()
		}
	}
}
func out(ps ...*Poly) {
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
	check(C.lr_poly_zero(p.d))
}

// MarshalBinary / UnmarshalBinary (ring/ring_object.go:222,252): the big-endian image is produced and consumed by the
// device, so a resident polynomial never takes the host detour.
func (p *Poly) MarshalBinary() ([]byte, error) {
	in(p)
	data := make([]byte, 2+(len(p.Coeffs)*len(p.Coeffs[0]))<<3)
	var n C.size_t
	rc := C.lr_poly_marshal(p.d, 0, (*C.uint8_t)(unsafe.Pointer(&data[0])), C.size_t(len(data)), &n)
	return data[:int(n)], statusErr(rc)
}

func (p *Poly) UnmarshalBinary(data []byte) error {
	if rc := C.lr_poly_unmarshal(p.d, 0, (*C.uint8_t)(unsafe.Pointer(&data[0])), C.size_t(len(data))); rc != C.LR_OK {
		return statusErr(rc)
	}
	out(p)
	return nil
}
