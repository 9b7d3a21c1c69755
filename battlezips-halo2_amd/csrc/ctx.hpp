// Context, workspace and timing plumbing behind the C ABI (include/bzh2.h).
#pragma once
#include <hip/hip_runtime.h>

#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/bzh2.h"

struct bzh_bases {
    int curve = 0;
    size_t n = 0;
    uint32_t* d_xy = nullptr;  // affine x||y, Montgomery form: n points, or pre_nwin rows of n
    int device = 0;
    int pre_c = 0;     // != 0: d_xy holds the window table, row w = 2^(pre_c * w) * G_i
    int pre_nwin = 0;
};

struct bzh_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::mutex mu;
    std::string last_error;
    // grow-only device workspaces (calls on a ctx are serialised and stream-ordered)
    static constexpr int kWsSlots = 6;  // 0-2 msm / scans, 3 host staging, 4-5 ipa
    void* ws[kWsSlots] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t ws_bytes[kWsSlots] = {0, 0, 0, 0, 0, 0};
    // profiling
    bool profiling = false;
    struct Span {
        int cls;
        hipEvent_t a, b;
    };
    std::vector<Span> spans;
    std::vector<hipEvent_t> event_pool;
    double acc_ms[BZH_T_COUNT] = {0};
    uint64_t acc_n[BZH_T_COUNT] = {0};
    double alg_bytes[BZH_T_COUNT] = {0};  // algorithmic bytes of the launches timed while profiling
    int num_cu = 256;
    // pinned upload ring: small host->device copies stay asynchronous (a pageable hipMemcpyAsync waits for the copy)
    char* pin = nullptr;
    size_t pin_bytes = 0, pin_off = 0;
};

#define BZH_HIP_TRY(ctx, expr)                                                                    \
    do {                                                                                          \
        hipError_t e__ = (expr);                                                                  \
        if (e__ != hipSuccess) {                                                                  \
            (ctx)->last_error = std::string(#expr) + ": " + hipGetErrorString(e__);               \
            return (e__ == hipErrorOutOfMemory) ? BZH_E_OOM : BZH_E_HIP;                          \
        }                                                                                         \
    } while (0)

namespace bzh {

// grow-only workspace slot
inline int ws_ensure(bzh_ctx* ctx, int slot, size_t bytes, void** out) {
    if (ctx->ws_bytes[slot] < bytes) {
        if (ctx->ws[slot]) {
            BZH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            BZH_HIP_TRY(ctx, hipFree(ctx->ws[slot]));
            ctx->ws[slot] = nullptr;
            ctx->ws_bytes[slot] = 0;
        }
        size_t want = bytes + (bytes >> 3);
        BZH_HIP_TRY(ctx, hipMalloc(&ctx->ws[slot], want));
        ctx->ws_bytes[slot] = want;
    }
    *out = ctx->ws[slot];
    return BZH_OK;
}

// host -> device copy of a small, short-lived host buffer through the pinned ring: returns at once, stream-ordered
inline int h2d_small(bzh_ctx* ctx, void* dst, const void* src, size_t bytes) {
    constexpr size_t kRing = (size_t)8 << 20;
    if (!bytes) return BZH_OK;
    if (bytes > kRing / 8) {
        BZH_HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        return BZH_OK;
    }
    if (!ctx->pin) {
        BZH_HIP_TRY(ctx, hipHostMalloc((void**)&ctx->pin, kRing, hipHostMallocDefault));
        ctx->pin_bytes = kRing;
        ctx->pin_off = 0;
    }
    const size_t need = (bytes + 63) & ~(size_t)63;
    if (ctx->pin_off + need > ctx->pin_bytes) {  // wrap: everything queued from the ring has to have landed
        BZH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        ctx->pin_off = 0;
    }
    memcpy(ctx->pin + ctx->pin_off, src, bytes);
    BZH_HIP_TRY(ctx, hipMemcpyAsync(dst, ctx->pin + ctx->pin_off, bytes, hipMemcpyHostToDevice, ctx->stream));
    ctx->pin_off += need;
    return BZH_OK;
}

struct ScopedTimer {
    bzh_ctx* ctx;
    int cls;
    hipEvent_t a = nullptr, b = nullptr;
    ScopedTimer(bzh_ctx* c, int k) : ctx(c), cls(k) {
        if (!ctx->profiling) return;
        a = take();
        b = take();
        if (a && b) (void)hipEventRecord(a, ctx->stream);
    }
    ~ScopedTimer() {
        if (!ctx->profiling || !a || !b) return;
        (void)hipEventRecord(b, ctx->stream);
        ctx->spans.push_back({cls, a, b});
    }
    hipEvent_t take() {
        if (!ctx->event_pool.empty()) {
            hipEvent_t e = ctx->event_pool.back();
            ctx->event_pool.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        return e;
    }
};

// implemented in msm.hip / ntt.hip
int msm_run(bzh_ctx* ctx, const bzh_bases* bases, const uint32_t* d_scalars, size_t n, size_t batch, int form,
            uint32_t* d_out_xyz);
int msm_run_paired(bzh_ctx* ctx, const bzh_bases* bases, const uint32_t* d_scalars, size_t n_pair, unsigned log_m, size_t batch,
                   int form, uint32_t* d_out_xyz);
int ntt_run(bzh_ctx* ctx, int field, uint32_t* d_data, unsigned log_n, size_t batch, const uint64_t* omega,
            const uint64_t* coset_shift, int inverse, int form);
int bases_to_montgomery(bzh_ctx* ctx, int curve, uint32_t* d_xy, size_t n);
int bases_precompute(bzh_ctx* ctx, bzh_bases* bases, int window_bits);
// polyops.hip (device pointers, Montgomery form)
int field_convert(bzh_ctx* ctx, int field, uint32_t* d, size_t count, int to_mont);
int poly_batch_invert(bzh_ctx* ctx, int field, uint32_t* d, size_t count);
int poly_prefix_product(bzh_ctx* ctx, int field, uint32_t* d, size_t n, size_t batch);
int poly_eval(bzh_ctx* ctx, int field, const uint32_t* coeffs, size_t n, size_t batch, const uint32_t* xs, size_t x_stride,
              uint32_t* out);
int poly_inner_product(bzh_ctx* ctx, int field, const uint32_t* a, const uint32_t* b, size_t n, size_t batch, uint32_t* out);
int poly_fold(bzh_ctx* ctx, int field, const uint32_t* in, size_t half, size_t batch, const uint32_t* u, size_t u_stride,
              uint32_t* out);
int poly_vec_mul(bzh_ctx* ctx, int field, uint32_t* a, const uint32_t* b, size_t count);
int poly_kate_division(bzh_ctx* ctx, int field, const uint32_t* d_c, size_t n, size_t batch, const uint32_t* d_xs, uint32_t* d_q);
// exprvm.hip
int expr_eval(bzh_ctx* ctx, int field, const void* d_prog, int nops, const uint32_t* const* d_cols, const size_t* d_strides,
              const uint32_t* d_consts, size_t const_stride, size_t size, int result_slot, size_t batch, uint32_t* d_out);
// ipa.hip
int random_field(bzh_ctx* ctx, int field, const uint32_t* d_raw, size_t count, uint32_t* d_out);
int ipa_open(bzh_ctx* ctx, const bzh_bases* bases, const uint32_t* d_polys, size_t batch, const uint64_t* blinds,
             const uint64_t* x3s, const uint8_t* rng_bytes, size_t rng_stride, bzh_transcript* const* trs, uint64_t* out_v);
int ipa_verify(bzh_ctx* ctx, const bzh_bases* bases, const uint64_t* commitment_xy, const uint64_t* x3, const uint64_t* v,
               const uint8_t* proof, size_t proof_len, bzh_transcript* tr, const uint64_t* g0_u_w_xy);

}  // namespace bzh
