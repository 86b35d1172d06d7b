package ring

// #include "lattigo_ring.h"
import "C"

// Options is lr_options (lattigo_ring.h): the configuration of the device library -- kernel selection switches (every alternative gives
// the same bits) and launch-shape thresholds.  NewOptions returns the defaults (lr_options_init); Set names a field as the header does.
// A nil *Options anywhere means the defaults.  The LR_* environment variables are a test-only override of the same fields.
type Options struct{ c C.lr_options }

// DefaultOptions, when set, is what NewContextWithParams, NewCkksPlan and NewBfvPlan create their handles with.
var DefaultOptions *Options

func NewOptions() *Options {
	o := &Options{}
	call(func() C.int { return C.lr_options_init(&o.c) })
	return o
}

// Set assigns one field by its header name (e.g. "no_pair", "fork_below_workgroups") and returns o for chaining.
func (o *Options) Set(field string, value int64) *Options {
	switch field {
	case "no_asm":
		o.c.no_asm = C.int32_t(value)
	case "no_fp":
		o.c.no_fp = C.int32_t(value)
	case "ntt_mode":
		o.c.ntt_mode = C.int32_t(value)
	case "asm_variant":
		o.c.asm_variant = C.int32_t(value)
	case "asm14_1024":
		o.c.asm14_1024 = C.int32_t(value)
	case "no_wide14_small":
		o.c.no_wide14_small = C.int32_t(value)
	case "wide14_max_items":
		o.c.wide14_max_items = C.int32_t(value)
	case "ntt_split15":
		o.c.ntt_split15 = C.int32_t(value)
	case "split15_max_workgroups":
		o.c.split15_max_workgroups = C.int32_t(value)
	case "no_invfuse":
		o.c.no_invfuse = C.int32_t(value)
	case "no_grid_padding":
		o.c.no_grid_padding = C.int32_t(value)
	case "ntt_stagger":
		o.c.ntt_stagger = C.int32_t(value)
	case "ntt_persist":
		o.c.ntt_persist = C.int32_t(value)
	case "ntt_timeline":
		o.c.ntt_timeline = C.int32_t(value)
	case "no_epilogue":
		o.c.no_epilogue = C.int32_t(value)
	case "no_int_epilogue":
		o.c.no_int_epilogue = C.int32_t(value)
	case "rescale_unfused":
		o.c.rescale_unfused = C.int32_t(value)
	case "rescale_unpaired":
		o.c.rescale_unpaired = C.int32_t(value)
	case "pair_max_workgroups":
		o.c.pair_max_workgroups = C.int32_t(value)
	case "ext_narrow":
		o.c.ext_narrow = C.int32_t(value)
	case "ext_ieee_div":
		o.c.ext_ieee_div = C.int32_t(value)
	case "no_ext_chunks":
		o.c.no_ext_chunks = C.int32_t(value)
	case "no_staging":
		o.c.no_staging = C.int32_t(value)
	case "no_exttop":
		o.c.no_exttop = C.int32_t(value)
	case "no_invtop":
		o.c.no_invtop = C.int32_t(value)
	case "no_ext_group":
		o.c.no_ext_group = C.int32_t(value)
	case "keymac_narrow":
		o.c.keymac_narrow = C.int32_t(value)
	case "no_pair":
		o.c.no_pair = C.int32_t(value)
	case "no_fork":
		o.c.no_fork = C.int32_t(value)
	case "fork_below_workgroups":
		o.c.fork_below_workgroups = C.int32_t(value)
	case "bfv_no_ext_epilogue":
		o.c.bfv_no_ext_epilogue = C.int32_t(value)
	case "bfv_no_gather":
		o.c.bfv_no_gather = C.int32_t(value)
	case "bfv_gather_below":
		o.c.bfv_gather_below = C.int64_t(value)
	default:
		panic("lr_options has no field " + field)
	}
	return o
}

func (o *Options) ptr() *C.lr_options {
	if o == nil {
		return nil
	}
	return &o.c
}

// Options of a live context: what it ended up with after the test-only environment override (lr_context_get_options).
func (c *Context) Options() *Options {
	o := &Options{}
	call(func() C.int { return C.lr_context_get_options(c.h, &o.c) })
	return o
}
