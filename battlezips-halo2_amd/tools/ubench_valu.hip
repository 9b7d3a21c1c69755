// Instruction-throughput microbenchmark for the integer/FP64 VALU ops that bound
// 256-bit modular multiplication on gfx950.  Prints lane-ops per clock per CU.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_valu ubench_valu.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int ITERS = 4096;
constexpr int UNROLL = 8;  // independent chains per lane

#define KERNEL(NAME, DECL, BODY, SINK)                                                     \
    __global__ void __launch_bounds__(256) NAME(uint32_t* out, uint32_t seed) {            \
        DECL;                                                                              \
        for (int it = 0; it < ITERS; it++) {                                               \
            BODY;                                                                          \
        }                                                                                  \
        uint32_t s = SINK;                                                                 \
        if (s == 0x12345678u) out[threadIdx.x] = s;                                        \
    }

#define DECL64 uint64_t a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; uint32_t x = threadIdx.x | 1, y = seed | 3
#define BODY_MAD64 asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n" \
                    "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n" \
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y) : "vcc")
#define SINK64 (uint32_t)(a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7)
KERNEL(k_mad_u64_u32, DECL64, BODY_MAD64, SINK64)

#define OP8(OPSTR)                                                                                                       \
    asm volatile(OPSTR " %0, %8, %0\n " OPSTR " %1, %8, %1\n " OPSTR " %2, %8, %2\n " OPSTR " %3, %8, %3\n " OPSTR        \
                       " %4, %8, %4\n " OPSTR " %5, %8, %5\n " OPSTR " %6, %8, %6\n " OPSTR " %7, %8, %7\n"                \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x))
#define DECL32 uint32_t a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; uint32_t x = threadIdx.x | 1
#define SINK32 (a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7)

#define B_MUL_LO OP8("v_mul_lo_u32")
#define B_MUL_HI OP8("v_mul_hi_u32")
#define B_MUL24 OP8("v_mul_u32_u24")
#define B_MULHI24 OP8("v_mul_hi_u32_u24")
#define B_ADD OP8("v_add_u32")
#define B_XOR OP8("v_xor_b32")
KERNEL(k_mul_lo_u32, DECL32, B_MUL_LO, SINK32)
KERNEL(k_mul_hi_u32, DECL32, B_MUL_HI, SINK32)
KERNEL(k_mul_u32_u24, DECL32, B_MUL24, SINK32)
KERNEL(k_mul_hi_u32_u24, DECL32, B_MULHI24, SINK32)
KERNEL(k_add_u32, DECL32, B_ADD, SINK32)
KERNEL(k_xor_b32, DECL32, B_XOR, SINK32)

#define OP8_3(OPSTR)                                                                                                     \
    asm volatile(OPSTR " %0, %8, %9, %0\n " OPSTR " %1, %8, %9, %1\n " OPSTR " %2, %8, %9, %2\n " OPSTR " %3, %8, %9, %3\n " \
                 OPSTR " %4, %8, %9, %4\n " OPSTR " %5, %8, %9, %5\n " OPSTR " %6, %8, %9, %6\n " OPSTR " %7, %8, %9, %7\n"  \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y))
#define DECL32Y DECL32; uint32_t y = seed | 5
#define B_MAD24 OP8_3("v_mad_u32_u24")
#define B_ADD3 OP8_3("v_add3_u32")
#define B_DOT2 OP8_3("v_dot2_u32_u16")
#define B_DOT4 OP8_3("v_dot4_u32_u8")
#define B_PKMAD OP8_3("v_pk_mad_u16")
#define B_LSHLADD OP8_3("v_lshl_add_u32")
#define B_MAD16 OP8_3("v_mad_u32_u16")
KERNEL(k_mad_u32_u24, DECL32Y, B_MAD24, SINK32)
KERNEL(k_add3_u32, DECL32Y, B_ADD3, SINK32)
KERNEL(k_dot2_u32_u16, DECL32Y, B_DOT2, SINK32)
KERNEL(k_dot4_u32_u8, DECL32Y, B_DOT4, SINK32)
KERNEL(k_mad_u16_pk, DECL32Y, B_PKMAD, SINK32)
KERNEL(k_lshl_add, DECL32Y, B_LSHLADD, SINK32)
KERNEL(k_mad_u32_u16, DECL32Y, B_MAD16, SINK32)

#define B_ADDC asm volatile("v_add_co_u32 %0, vcc, %8, %0\n v_addc_co_u32 %1, vcc, %9, %1, vcc\n v_add_co_u32 %2, vcc, %8, %2\n v_addc_co_u32 %3, vcc, %9, %3, vcc\n" \
                    "v_add_co_u32 %4, vcc, %8, %4\n v_addc_co_u32 %5, vcc, %9, %5, vcc\n v_add_co_u32 %6, vcc, %8, %6\n v_addc_co_u32 %7, vcc, %9, %7, vcc\n" \
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y) : "vcc")
KERNEL(k_addc_pair, DECL32Y, B_ADDC, SINK32)

#define DECLF64 double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; double x = 1.0 + 1e-9 * threadIdx.x, y = 1e-7
#define B_FMA64 asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n" \
                    "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n" \
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y))
#define SINKF (uint32_t)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
KERNEL(k_fma_f64, DECLF64, B_FMA64, SINKF)
#define DECLF32 float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; float x = 1.0f + 1e-6f * threadIdx.x, y = 1e-7f
#define B_FMA32 asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n" \
                    "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n" \
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x), "v"(y))
KERNEL(k_fma_f32, DECLF32, B_FMA32, SINKF)

typedef void (*kfn)(uint32_t*, uint32_t);
struct Case { const char* name; kfn fn; int ops_per_iter; };

// round 4: the 64-bit shift / add forms the compiler picks for the unsaturated product's column hand-over, and their 32-bit stand-ins
#define OP8_64S(OPSTR)                                                                                                   \
    asm volatile(OPSTR " %0, 29, %0\n " OPSTR " %1, 29, %1\n " OPSTR " %2, 29, %2\n " OPSTR " %3, 29, %3\n " OPSTR      \
                       " %4, 29, %4\n " OPSTR " %5, 29, %5\n " OPSTR " %6, 29, %6\n " OPSTR " %7, 29, %7\n"              \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7))
#define B_LSHR64 OP8_64S("v_lshrrev_b64")
#define B_LSHL64 OP8_64S("v_lshlrev_b64")
KERNEL(k_lshrrev_b64, DECL64, B_LSHR64, SINK64)
KERNEL(k_lshlrev_b64, DECL64, B_LSHL64, SINK64)
#define B_LSHLADD64 asm volatile("v_lshl_add_u64 %0, %0, 0, %8\n v_lshl_add_u64 %1, %1, 0, %8\n v_lshl_add_u64 %2, %2, 0, %8\n v_lshl_add_u64 %3, %3, 0, %8\n" \
                    "v_lshl_add_u64 %4, %4, 0, %8\n v_lshl_add_u64 %5, %5, 0, %8\n v_lshl_add_u64 %6, %6, 0, %8\n v_lshl_add_u64 %7, %7, 0, %8\n" \
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(a7 ^ seed))
KERNEL(k_lshl_add_u64, DECL64, B_LSHLADD64, SINK64)
#define B_ALIGNBIT asm volatile("v_alignbit_b32 %0, %8, %0, 29\n v_alignbit_b32 %1, %8, %1, 29\n v_alignbit_b32 %2, %8, %2, 29\n v_alignbit_b32 %3, %8, %3, 29\n" \
                    "v_alignbit_b32 %4, %8, %4, 29\n v_alignbit_b32 %5, %8, %5, 29\n v_alignbit_b32 %6, %8, %6, 29\n v_alignbit_b32 %7, %8, %7, 29\n" \
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(x))
KERNEL(k_alignbit, DECL32, B_ALIGNBIT, SINK32)
#define B_AND OP8("v_and_b32")
#define B_LSHR32 OP8("v_lshrrev_b32")
#define B_SUB OP8("v_sub_u32")
KERNEL(k_and_b32, DECL32, B_AND, SINK32)
KERNEL(k_lshrrev_b32, DECL32, B_LSHR32, SINK32)
KERNEL(k_sub_u32, DECL32, B_SUB, SINK32)
#define B_ANDOR OP8_3("v_and_or_b32")
#define B_BFE OP8_3("v_bfe_u32")
KERNEL(k_and_or, DECL32Y, B_ANDOR, SINK32)
KERNEL(k_bfe, DECL32Y, B_BFE, SINK32)

int main() {
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    double clk_ghz = prop.clockRate / 1e6;
    printf("device %s, %d CUs, clock %.2f GHz\n", prop.name, cus, clk_ghz);
    uint32_t* d_out;
    CHK(hipMalloc(&d_out, 4096));
    Case cases[] = {
        {"v_mad_u64_u32", k_mad_u64_u32, 8}, {"v_mul_lo_u32", k_mul_lo_u32, 8}, {"v_mul_hi_u32", k_mul_hi_u32, 8},
        {"v_mul_u32_u24", k_mul_u32_u24, 8}, {"v_mul_hi_u32_u24", k_mul_hi_u32_u24, 8}, {"v_mad_u32_u24", k_mad_u32_u24, 8},
        {"v_mad_u32_u16", k_mad_u32_u16, 8}, {"v_dot2_u32_u16", k_dot2_u32_u16, 8}, {"v_dot4_u32_u8", k_dot4_u32_u8, 8},
        {"v_pk_mad_u16", k_mad_u16_pk, 8}, {"v_add_u32", k_add_u32, 8}, {"v_xor_b32", k_xor_b32, 8},
        {"v_add3_u32", k_add3_u32, 8}, {"v_lshl_add_u32", k_lshl_add, 8}, {"add_co+addc pair(2 ops)", k_addc_pair, 8},
        {"v_fma_f64", k_fma_f64, 8}, {"v_fma_f32", k_fma_f32, 8},
        {"v_lshrrev_b64", k_lshrrev_b64, 8}, {"v_lshlrev_b64", k_lshlrev_b64, 8}, {"v_lshl_add_u64", k_lshl_add_u64, 8},
        {"v_alignbit_b32", k_alignbit, 8}, {"v_and_b32", k_and_b32, 8}, {"v_lshrrev_b32", k_lshrrev_b32, 8}, {"v_sub_u32", k_sub_u32, 8},
        {"v_and_or_b32", k_and_or, 8}, {"v_bfe_u32", k_bfe, 8},
    };
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    for (int waves_per_simd : {1, 2, 8}) {
        printf("--- %d wave(s) per SIMD ---\n", waves_per_simd);
        for (auto& c : cases) {
            dim3 grid(cus * waves_per_simd), block(256);
            hipLaunchKernelGGL(c.fn, grid, block, 0, 0, d_out, 7u);
            CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(e0));
            for (int r = 0; r < 5; r++) hipLaunchKernelGGL(c.fn, grid, block, 0, 0, d_out, 7u);
            CHK(hipEventRecord(e1));
            CHK(hipEventSynchronize(e1));
            float ms;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            double lane_ops = 5.0 * (double)grid.x * 256 * ITERS * c.ops_per_iter;
            double per_s = lane_ops / (ms * 1e-3);
            printf("%-26s %8.3f ms  %8.2f Tlane-op/s  %7.1f lane-ops/clk/CU (at %.2f GHz)\n", c.name, ms / 5, per_s / 1e12,
                   per_s / cus / (clk_ghz * 1e9), clk_ghz);
        }
    }
    return 0;
}
