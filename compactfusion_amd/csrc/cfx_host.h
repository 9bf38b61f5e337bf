// Host-side internals shared by the translation units of libcfx.so's streaming codecs (cfx_api.hip: context, C-ABI, dispatch; cfx_absmean.hip,
// cfx_minmax.hip, cfx_topk.hip: each family's kernels AND the code that launches them).  Not part of the ABI.
#ifndef CFX_HOST_H
#define CFX_HOST_H
#include "cfx_device.h"

// One compress call after validation: what cfx_api.hip's compress_impl has checked and packed, handed to the family that launches it.
struct CompressCall {
    cfx_ctx* ctx;
    int codec, N, C, param, flags, batch;
    const cfx_comp_item* items;
    int n_ride, n_gated;
    const cfx_decomp_item* gated;
    void* stream;
    CfxXGate* xg;                    // exchange-layer op: the gated items wait on an external gate (see compress_impl)
    BatchC b;
    BatchD rd, gd;
    u64* ws;
    size_t wstride;
    int CB;
    bool upd, capturing;             // capturing: the stream is under hipGraph capture - no one-launch layer form (include/cfx.h)
    // the tile geometry of the abs-mean and min/max families' compress launches (compress_impl, before the dispatch)
    bool fused;
    unsigned* tick;
    unsigned slot;
    int stream_cus, R, P;
};
extern "C" {
CFX_HIDDEN int cfx_i_topk_compress(CompressCall& cc);
CFX_HIDDEN int cfx_i_absmean_compress(CompressCall& cc);
CFX_HIDDEN int cfx_i_minmax_compress(CompressCall& cc);
// reconstruction launches of a validated batch (decompress_impl's dispatch); `pre` / `pre_val`: an optional flag word the kernel waits for
CFX_HIDDEN int cfx_i_absmean_decompress(cfx_ctx* ctx, int codec, int N, int C, int batch, const BatchD& b, int R, void* stream, unsigned* pre, unsigned pre_val);
CFX_HIDDEN int cfx_i_minmax_decompress(cfx_ctx* ctx, int codec, int N, int C, int batch, const BatchD& b, int R, void* stream, unsigned* pre, unsigned pre_val);
CFX_HIDDEN int cfx_i_topk_decompress(cfx_ctx* ctx, int N, int C, int param, int batch, const BatchD& b, void* stream, unsigned* pre, unsigned pre_val);
// cfx_api.hip
CFX_HIDDEN int cfx_i_auto_rows(const cfx_ctx* ctx, int N, int C, int batch, bool stats);
CFX_HIDDEN int cfx_i_fused_rows(const cfx_ctx* ctx, int N, int C, int batch, int cus);
CFX_HIDDEN unsigned cfx_i_ticket_slot(cfx_ctx* ctx, void* stream);
CFX_HIDDEN void cfx_i_fill_p2p(cfx_ctx* ctx, CfxXGate* xg, P2PInline& p);
CFX_HIDDEN int cfx_i_decompress_checked(cfx_ctx* ctx, int codec, int N, int C, int param, int batch, const cfx_decomp_item* items, void* stream,
                                        unsigned* pre, unsigned pre_val);
}  // extern "C"
// (short names the family files were written with)
#define auto_rows cfx_i_auto_rows
#define fused_rows cfx_i_fused_rows
#define ticket_slot cfx_i_ticket_slot
#define fill_p2p cfx_i_fill_p2p
#define stream_cu_count cfx_i_stream_cus
#define decompress_impl cfx_i_decompress_checked
#ifndef CFX_API_TU
#define shape_ok cfx_i_shape_ok
#define ws_words cfx_i_ws_words
#endif
#endif
