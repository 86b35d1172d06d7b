package ring

// #include "lattigo_ring.h"
import "C"

// Multi-device use from ONE Go process (SURVEY 8(e); upstream's parallel model is a pool of goroutines, one evaluator each,
// examples/dbfv/psi/psi.go:215-233): every goroutine creates its contexts on its own GPU (NewContextWithParamsOnDevice), works on its
// block of independent ciphertexts, and hands finished polynomials to the root device with the calls below -- direct peer copies over
// xGMI on a copy stream per device pair, ordered on the devices behind the producer's kernels.  No second process, no collective library.

// CopyFrom copies src (resident on srcCtx's device) into dst (resident on c's device).  Asynchronous: the copy waits, on the device, for what
// srcCtx has enqueued so far and overlaps what it is given next; call WaitPeerCopies on c before c's work reads dst.
func (c *Context) CopyFrom(dst *Poly, srcCtx *Context, src *Poly) {
	srcCtx.use(src)
	dst.Pin(c)
	c.want(dst)
	call(func() C.int { return C.lr_poly_copy_peer(c.h, dst.d, 0, srcCtx.h, src.d, 0, 1) })
}

// WaitPeerCopies makes c's stream wait (on the device) for every peer copy into c's device enqueued so far.
func (c *Context) WaitPeerCopies() {
	call(func() C.int { return C.lr_context_wait_peer_copies(c.h) })
}

// GatherTo brings srcs[i] (on srcCtxs[i]'s device) to dst[i] on c's device, all of them, and waits for the copies on c's stream: the
// gather of a sharded batch's results to the root in global unit order.
func (c *Context) GatherTo(dst []*Poly, srcCtxs []*Context, srcs []*Poly) {
	if len(dst) != len(srcs) || len(srcCtxs) != len(srcs) {
		panic("GatherTo: dst, srcCtxs and srcs must have the same length")
	}
	for i := range srcs {
		c.CopyFrom(dst[i], srcCtxs[i], srcs[i])
	}
	c.WaitPeerCopies()
}
