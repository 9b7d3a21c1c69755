// Context, workspace and timing plumbing behind the C ABI (include/bzh2.h).
#pragma once
#include <hip/hip_runtime.h>
#include <sched.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
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
    // Several tables side by side (one per proof: the IPA's collapsed generators, msm_collapse_table): the rows of the table
    // array are row_stride points apart (0: n) and vector v of a batch reads the columns [v * vec_col_stride, + n)
    size_t row_stride = 0;
    size_t vec_col_stride = 0;
    // Window tables of the Pasta curves carry a second copy for k_msm_accumulate's unsaturated-limb loop: point i as 20 words --
    // x then y in 9 x 29-bit limbs (x 2^261, below 2 p, carried), 2 words of padding -- so that the loop loads its operands
    // ready-made instead of re-slicing and folding two coordinates per addition (csrc/fe29.cuh).  Same indexing as d_xy.
    uint32_t* d_xy29 = nullptr;
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
    unsigned long long* d_add_counter = nullptr;   // device counter of bucket additions (non-zero window digits), while profiling
    bool msm_attr_set[3] = {false, false, false};  // hipFuncSetAttribute(MaxDynamicSharedMemorySize) done on this ctx's device, per curve
    // pinned upload ring: small host->device copies stay asynchronous (a pageable hipMemcpyAsync waits for the copy)
    char* pin = nullptr;
    size_t pin_bytes = 0, pin_off = 0;
    char* pin_big = nullptr;  // grow-only pinned buffer for transfers above 1 MiB
    size_t pin_big_bytes = 0, pin_big_off = 0;
    struct PendingD2H {
        void* dst;
        const char* slot;
        size_t bytes;
    };
    std::vector<PendingD2H> pending_d2h;
    uint32_t* ped_tbl = nullptr;   // bzh_pedersen_commit_batch's direct-lookup table of V and R (csrc/pedersen.hip), built on first use
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

// Host threads one call may start (witness synthesis, the lookup's sort helpers, the verifier's per-proof pool, Params::new's
// hash-to-curve): $BZH_HOST_THREADS when set -- a multi-rank launcher gives every rank its share of the node's cores there
// (bench.py: affinity count / world size) --, else the calling thread's CPU affinity count (std::thread::hardware_concurrency
// counts the machine's cores, not the cgroup / taskset share).  Read per call: a launcher may set it after loading the library.
inline unsigned host_thread_budget() {
    if (const char* e = getenv("BZH_HOST_THREADS")) {
        const int v = atoi(e);
        if (v > 0) return (unsigned)v;
    }
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) {
        const int c = CPU_COUNT(&set);
        if (c > 0) return (unsigned)c;
    }
    return 1u;
}

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

// ---------------------------------------------------------------------------
// Host <-> device staging through pinned memory and copy KERNELS.  The runtime's own copy paths (SDMA / blit
// through hipMemcpyAsync) showed 3x slower transfers and multi-millisecond stalls on small copies in every
// process after the first one on a box ("transfers"); a kernel that reads or writes mapped pinned
// host memory has launch-like latency and is stream-ordered like everything else.
//   h2d_small  short-lived host buffer -> device, asynchronous (ring of pinned slots, or the big buffer)
//   d2h_async  device -> host destination, completed by d2h_finish (one stream sync for any number of them)
// ---------------------------------------------------------------------------
static __global__ void __launch_bounds__(256) k_xfer16(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n16) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
static __global__ void __launch_bounds__(256) k_xfer4(uint32_t* __restrict__ dst, const uint32_t* __restrict__ src, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
// stream-ordered copy between two device-visible addresses (device memory or mapped pinned host memory)
inline int xfer_launch(bzh_ctx* ctx, void* dst, const void* src, size_t bytes, hipMemcpyKind fallback) {
    if (!bytes) return BZH_OK;
    const uintptr_t al = (uintptr_t)dst | (uintptr_t)src | (uintptr_t)bytes;
    if ((al & 15) == 0) {
        const size_t n16 = bytes / 16;
        const unsigned blocks = (unsigned)std::min<size_t>((n16 + 255) / 256, 2048);
        hipLaunchKernelGGL(k_xfer16, dim3(blocks), dim3(256), 0, ctx->stream, (uint4*)dst, (const uint4*)src, n16);
    } else if ((al & 3) == 0) {
        const size_t n4 = bytes / 4;
        const unsigned blocks = (unsigned)std::min<size_t>((n4 + 255) / 256, 2048);
        hipLaunchKernelGGL(k_xfer4, dim3(blocks), dim3(256), 0, ctx->stream, (uint32_t*)dst, (const uint32_t*)src, n4);
    } else {
        BZH_HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, fallback, ctx->stream));
        return BZH_OK;
    }
    BZH_HIP_TRY(ctx, hipGetLastError());
    return BZH_OK;
}
inline int pin_big_ensure(bzh_ctx* ctx, size_t bytes) {
    if (ctx->pin_big_bytes >= bytes) return BZH_OK;
    BZH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (auto& p : ctx->pending_d2h) memcpy(p.dst, p.slot, p.bytes);
    ctx->pending_d2h.clear();
    if (ctx->pin_big) BZH_HIP_TRY(ctx, hipHostFree(ctx->pin_big));
    ctx->pin_big = nullptr;
    ctx->pin_big_bytes = 0;
    const size_t want = bytes + (bytes >> 2) + 4096;
    BZH_HIP_TRY(ctx, hipHostMalloc((void**)&ctx->pin_big, want, hipHostMallocDefault));
    ctx->pin_big_bytes = want;
    ctx->pin_big_off = 0;
    return BZH_OK;
}
// slot of `bytes` in the big pinned buffer; waits for the stream when the buffer has to be recycled
inline int pin_big_take(bzh_ctx* ctx, size_t bytes, char** out) {
    const size_t need = (bytes + 255) & ~(size_t)255;
    if (ctx->pin_big_bytes < need) {
        int rc = pin_big_ensure(ctx, need);
        if (rc) return rc;
    }
    if (ctx->pin_big_off + need > ctx->pin_big_bytes) {
        BZH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (auto& p : ctx->pending_d2h) memcpy(p.dst, p.slot, p.bytes);  // staged results leave before their slots are reused
        ctx->pending_d2h.clear();
        ctx->pin_big_off = 0;
    }
    *out = ctx->pin_big + ctx->pin_big_off;
    ctx->pin_big_off += need;
    return BZH_OK;
}
// make room for a group of big slots that have to stay valid together (no recycling in between)
inline int pin_big_reserve(bzh_ctx* ctx, size_t total) {
    total += 4096;
    int rc = pin_big_ensure(ctx, total);
    if (rc) return rc;
    if (ctx->pin_big_off + total > ctx->pin_big_bytes) {
        BZH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (auto& p : ctx->pending_d2h) memcpy(p.dst, p.slot, p.bytes);
        ctx->pending_d2h.clear();
        ctx->pin_big_off = 0;
    }
    return BZH_OK;
}
inline int pin_ring_take(bzh_ctx* ctx, size_t bytes, char** out) {
    constexpr size_t kRing = (size_t)8 << 20;
    if (!ctx->pin) {
        BZH_HIP_TRY(ctx, hipHostMalloc((void**)&ctx->pin, kRing, hipHostMallocDefault));
        ctx->pin_bytes = kRing;
        ctx->pin_off = 0;
    }
    const size_t need = (bytes + 255) & ~(size_t)255;
    if (ctx->pin_off + need > ctx->pin_bytes) {  // wrap: everything queued from the ring has to have landed
        BZH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (auto& p : ctx->pending_d2h) memcpy(p.dst, p.slot, p.bytes);
        ctx->pending_d2h.clear();
        ctx->pin_off = 0;
    }
    *out = ctx->pin + ctx->pin_off;
    ctx->pin_off += need;
    return BZH_OK;
}
inline int h2d_small(bzh_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (!bytes) return BZH_OK;
    char* slot = nullptr;
    int rc = bytes > ((size_t)1 << 20) ? pin_big_take(ctx, bytes, &slot) : pin_ring_take(ctx, bytes, &slot);
    if (rc) return rc;
    memcpy(slot, src, bytes);
    return xfer_launch(ctx, dst, slot, bytes, hipMemcpyHostToDevice);
}
// pinned staging slot the caller fills itself (saves one host copy for large uploads), then h2d_commit
inline int h2d_stage(bzh_ctx* ctx, size_t bytes, char** slot) {
    return bytes > ((size_t)1 << 20) ? pin_big_take(ctx, bytes, slot) : pin_ring_take(ctx, bytes, slot);
}
inline int h2d_commit(bzh_ctx* ctx, void* dst, const char* slot, size_t bytes) {
    return xfer_launch(ctx, dst, slot, bytes, hipMemcpyHostToDevice);
}
inline int d2h_async(bzh_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes) {
    if (!bytes) return BZH_OK;
    char* slot = nullptr;
    int rc = bytes > ((size_t)1 << 20) ? pin_big_take(ctx, bytes, &slot) : pin_ring_take(ctx, bytes, &slot);
    if (rc) return rc;
    rc = xfer_launch(ctx, slot, dev_src, bytes, hipMemcpyDeviceToHost);
    if (rc) return rc;
    ctx->pending_d2h.push_back({host_dst, slot, bytes});
    return BZH_OK;
}
// one stream synchronisation, then the staged results are handed to their destinations
inline int d2h_finish(bzh_ctx* ctx) {
    BZH_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (auto& p : ctx->pending_d2h) memcpy(p.dst, p.slot, p.bytes);
    ctx->pending_d2h.clear();
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
// d_out29 != null (Pasta scalar fields): the evaluations leave the last pass as unsaturated planes (fe29.cuh, 9 * 2^log_n words per
// polynomial) instead of going to d_dst, which may then be null
int ntt_run_padded(bzh_ctx* ctx, int field, uint32_t* d_dst, const uint32_t* d_src, unsigned src_log, unsigned log_n, size_t batch,
                   const uint64_t* omega, const uint64_t* coset_shift, uint32_t* d_out29 = nullptr);
int bases_to_montgomery(bzh_ctx* ctx, int curve, uint32_t* d_xy, size_t n);
int bases_precompute(bzh_ctx* ctx, bzh_bases* bases, int window_bits);
// The IPA's generator collapse on the device.  For every proof b of `batch`: G'[b][i] = sum_{t < cnt} s[b][t] * G[i + t*m], i < m
// = n_srs / cnt, evaluated through the SRS window table `srs` (n_srs + 2 points), followed by U and W (the table's last two
// points), and expanded into a per-proof window table of c_tail-bit rows.  d_s: batch x cnt Montgomery scalars.
// d_table: ceil(256 / c_tail) rows of batch * (m + 2) affine points; d_scratch: msm_collapse_scratch_bytes() bytes.
// `out` is filled to describe the table array to msm_run_paired (it borrows d_table).
size_t msm_collapse_scratch_bytes(const bzh_bases* srs, size_t cnt, size_t batch, int c_tail);
// (d_table29: room for the fe29 copy of the collapsed tables, 20 words per point, or null)
int msm_collapse_table(bzh_ctx* ctx, const bzh_bases* srs, const uint32_t* d_s, size_t cnt, size_t batch, int c_tail, uint32_t* d_table29, uint32_t* d_table,
                       void* d_scratch, bzh_bases* out);
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
              const uint32_t* d_consts, size_t const_stride, size_t size, int result_slot, size_t batch, int nslots, uint32_t* d_out);
// VM v2 (stack-discipline registers + LDS slot file): the prover's quotient evaluator
int expr_eval2(bzh_ctx* ctx, int field, const void* d_prog, int nops, const uint32_t* const* d_cols, const size_t* d_strides,
               const uint32_t* d_consts, size_t const_stride, size_t size, size_t batch, int nlds, uint32_t* d_out);
// ipa.hip
int random_field(bzh_ctx* ctx, int field, const uint32_t* d_raw, size_t count, uint32_t* d_out);
int ipa_open(bzh_ctx* ctx, const bzh_bases* bases, const uint32_t* d_polys, size_t batch, const uint64_t* blinds,
             const uint64_t* x3s, const uint8_t* rng_bytes, size_t rng_stride, bzh_transcript* const* trs, uint64_t* out_v,
             const uint32_t* d_raw_in = nullptr);  // d_raw_in: the draws already on the device (batch x (n + 1 + 2k) x 64 B) instead of rng_bytes
int ipa_verify(bzh_ctx* ctx, const bzh_bases* bases, const uint64_t* commitment_xy, const uint64_t* x3, const uint64_t* v,
               const uint8_t* proof, size_t proof_len, bzh_transcript* tr, const uint64_t* g0_u_w_xy);
int ipa_check_batch(bzh_ctx* ctx, const bzh_bases* bases, size_t batch, size_t nl, const uint64_t* lc_pts, const uint64_t* lc_scal,
                    const uint64_t* cu, int* ok);
bool point_decompress(int curve, const uint8_t* in, uint64_t* xy_canonical);

}  // namespace bzh
