// Package ring is the drop-in shim for github.com/ldsec/lattigo/ring (v1.3.1): the same exported identifiers,
// method bodies forwarded through cgo to liblattigo_ring_hip.so (include/lattigo_ring.h).
//
// NOT COMPILED IN THIS REPOSITORY'S PIPELINE: neither the build container nor the GPU box has a Go toolchain
// (INTEGRATION.md).  The executable mirror of this file set is lattigo-fhe-by-go_amd/ring.py, which the parity
// tests drive through the same C ABI.
//
// Memory model.  A Poly keeps the reference's host view (Coeffs [][]uint64) and owns a device image.  By default
// every method is a literal drop-in: inputs are uploaded, the kernel runs, outputs are downloaded, so code that
// reads or writes Coeffs directly (the evaluators do, e.g. ckks/evaluator.go:1519-1534) keeps working.
// Poly.Pin() switches a polynomial to device-resident mode: uploads and downloads then happen only on Pin /
// Unpin / Sync, which is the mode the throughput numbers of DESIGN.md are measured in.
package ring

// #cgo CFLAGS:  -I${SRCDIR}/../../include
// #cgo LDFLAGS: -L${SRCDIR}/../../lattigo-fhe-by-go_amd -llattigo_ring_hip -Wl,-rpath,${SRCDIR}/../../lattigo-fhe-by-go_amd
// #include <stdlib.h>
// #include "lattigo_ring.h"
import "C"

import (
	"errors"
	"fmt"
)

// check panics on a non-zero status, like the reference panics on misuse (index out of range,
// ckks/evaluator.go:1027-1036); constructors translate the two documented statuses into their Go forms.
func check(rc C.int) {
	if rc != C.LR_OK {
		panic(fmt.Sprintf("lattigo_ring: status %d: %s", int(rc), C.GoString(C.lr_last_error_string())))
	}
}

func statusErr(rc C.int) error {
	if rc == C.LR_OK {
		return nil
	}
	return errors.New(C.GoString(C.lr_last_error_string()))
}
