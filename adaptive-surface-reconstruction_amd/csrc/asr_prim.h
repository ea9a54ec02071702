// asr_prim.h -- host helpers shared by the .hip translation units: counter flags, rocPRIM
// radix sort / scan wrappers running on the context stream with arena temporaries.
#pragma once
#include <rocprim/rocprim.hpp>

#include "asr_common.h"

namespace asr_prim {

// Device counters come in blocks of 64 ints out of a pool that is zeroed once: a pass that needs fresh counters takes the
// next block (fresh_flags) instead of clearing the one block there used to be -- twenty fill launches per geometry build,
// each a dependent launch in one of its two chains.  The pool is cleared again, on the context's stream, when it is used up.
constexpr int FLAG_BLOCK = 64, FLAG_BLOCKS = 512;
static inline int ensure_flags(asr_hip_context* ctx) {
    if (!ctx->d_flags_base) {
        ASR_HIP_CHECK(ctx, hipMalloc((void**)&ctx->d_flags_base, (size_t)FLAG_BLOCK * FLAG_BLOCKS * sizeof(int)));
        ASR_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_flags_base, 0, (size_t)FLAG_BLOCK * FLAG_BLOCKS * sizeof(int), ctx->stream));
        ctx->d_flags = ctx->d_flags_base;
        ctx->flags_next = 1;
    }
    return ASR_HIP_OK;
}
// ctx->d_flags = a block of zeroed counters (what hipMemsetAsync(ctx->d_flags, 0, ...) used to give)
static inline int fresh_flags(asr_hip_context* ctx) {
    ASR_TRY(ensure_flags(ctx));
    if (ctx->flags_next >= FLAG_BLOCKS) {
        ASR_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_flags_base, 0, (size_t)FLAG_BLOCK * FLAG_BLOCKS * sizeof(int), ctx->stream));
        ctx->flags_next = 0;
    }
    ctx->d_flags = ctx->d_flags_base + (size_t)FLAG_BLOCK * ctx->flags_next++;
    return ASR_HIP_OK;
}
static inline int read_flags(asr_hip_context* ctx, int* host) {
    ASR_HIP_CHECK(ctx, hipMemcpyAsync(host, ctx->d_flags, 16 * sizeof(int), hipMemcpyDeviceToHost,
                                      ctx->stream));
    ASR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return ASR_HIP_OK;
}
static inline int set_flag(asr_hip_context* ctx, int which, int value) {
    ASR_HIP_CHECK(ctx, hipMemcpyAsync(ctx->d_flags + which, &value, sizeof(int),
                                      hipMemcpyHostToDevice, ctx->stream));
    ASR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return ASR_HIP_OK;
}

static inline int sort_keys(asr_hip_context* ctx, Arena& arena, const u64* in, u64* out, i64 n, int end_bit = 64) {
    if (n <= 0) return ASR_HIP_OK;
    size_t tb = 0;
    ASR_HIP_CHECK(ctx, rocprim::radix_sort_keys(nullptr, tb, in, out, (size_t)n, 0, end_bit,
                                                ctx->stream));
    void* tmp = arena.alloc(tb ? tb : 256);
    if (!tmp) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
    ASR_HIP_CHECK(ctx, rocprim::radix_sort_keys(tmp, tb, in, out, (size_t)n, 0, end_bit,
                                                ctx->stream));
    return ASR_HIP_OK;
}
// stable LSD radix sort on key bits [begin_bit, end_bit)
template <class K, class V>
static inline int sort_pairs(asr_hip_context* ctx, Arena& arena, const K* kin, K* kout, const V* vin, V* vout,
               i64 n, int end_bit, int begin_bit = 0) {
    if (n <= 0) return ASR_HIP_OK;
    size_t tb = 0;
    ASR_HIP_CHECK(ctx, rocprim::radix_sort_pairs(nullptr, tb, kin, kout, vin, vout, (size_t)n, begin_bit,
                                                 end_bit, ctx->stream));
    void* tmp = arena.alloc(tb ? tb : 256);
    if (!tmp) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
    ASR_HIP_CHECK(ctx, rocprim::radix_sort_pairs(tmp, tb, kin, kout, vin, vout, (size_t)n, begin_bit,
                                                 end_bit, ctx->stream));
    return ASR_HIP_OK;
}
// out[0..n] = exclusive scan of in[0..n] (in has n+1 entries, in[n] ignored by callers that set it 0)
static inline int scan_counts(asr_hip_context* ctx, Arena& arena, const i64* in, i64* out, i64 n_plus_1) {
    size_t tb = 0;
    ASR_HIP_CHECK(ctx, rocprim::exclusive_scan(nullptr, tb, in, out, i64(0), (size_t)n_plus_1,
                                               rocprim::plus<i64>(), ctx->stream));
    void* tmp = arena.alloc(tb ? tb : 256);
    if (!tmp) ASR_FAIL(ctx, ASR_HIP_EHIP, "arena allocation failed");
    ASR_HIP_CHECK(ctx, rocprim::exclusive_scan(tmp, tb, in, out, i64(0), (size_t)n_plus_1,
                                               rocprim::plus<i64>(), ctx->stream));
    return ASR_HIP_OK;
}
static inline int read_i64(asr_hip_context* ctx, const i64* dev, i64* host) {
    ASR_HIP_CHECK(ctx, hipMemcpyAsync(host, dev, sizeof(i64), hipMemcpyDeviceToHost, ctx->stream));
    ASR_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return ASR_HIP_OK;
}

static inline int bits_for(i64 n) {
    int b = 1;
    while ((i64(1) << b) < n) ++b;
    return b;
}


}  // namespace asr_prim
