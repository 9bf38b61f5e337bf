/* cfx_dev.h - developer entry points of libcfx_dev.so (the same sources as libcfx.so compiled with -DCFX_DEV_PROBES:
 * `python -m compactfusion_amd.build --dev-probes`).  NOT part of the product ABI: libcfx.so exports none of these symbols, its kernels carry
 * no probe argument and no probe branch (csrc/cfx_internal.h: `Probe` is an empty type there), and compactfusion_amd/_lib.py does not bind
 * them.  Users: tools/*_stamps.py, tools/fused_probe.py (timelines of the layer launches and of the slab-resident low-rank chain) and
 * tests/tagwrap_child.py (walks a context across the wrap of its launch tags). */
#ifndef CFX_DEV_H
#define CFX_DEV_H
#include "cfx.h"
#ifdef __cplusplus
extern "C" {
#endif

/* When `buf` is non-NULL every workgroup of a layer launch / compress launch / k_lrs launch writes 16 u64 words at buf + 16 * workgroup:
 * phase times on the 100 MHz wall clock (word 7 of the layer kernels: the workgroup's role).  NULL switches it off. */
int cfx_dev_stamps(cfx_ctx* ctx, void* buf);
/* The layer launches tag what they hand over with numbers a context gives out in sequence (24 bits for the 1-bit / 2-bit layer, 31 for the
 * int4 / int8 layer); where a sequence wraps - 16.7 million, 2.1 billion launches in - the tagged arenas are zeroed and the numbers start
 * over.  This sets the two counters (after a device synchronisation) so that a test can walk a context across the wrap. */
int cfx_dev_set_launch_tags(cfx_ctx* ctx, unsigned abs_seq, unsigned mml_seq);
/* Early exits of the one-launch compress kernel (tools/fused_probe.py): 1 = stop after publishing, 2 = after the tickets, 3 = empty grid,
 * 4 = loads only; 0 = off. */
int cfx_dev_set_probe(cfx_ctx* ctx, int mode);

#ifdef __cplusplus
}
#endif
#endif
