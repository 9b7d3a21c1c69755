// Representation experiment: would a 5 x 52-bit limb field element multiplied with FP64 FMAs beat
// the 8 x 32-bit v_mad_u64_u32 Montgomery product every hot kernel uses (csrc/field.cuh, 160 G fe_mul/s)?
//
// Measured here, in isolation and with no memory traffic:
//   int  : the shipped fe_mul<Fp> (96 v_mad_u64_u32 + carries: product scanning + sparse-modulus reduction)
//   dp52 : a complete Montgomery multiplication on 5 double limbs (R = 2^260): every 52 x 52 partial product split exactly
//          into hi / lo with two FMAs (hi = fma(a, b, 2^104) - 2^104, lo = fma(a, b, -hi)), accumulated per column as
//          64-bit integers through the bit pattern of the doubles (one binade => the mantissa IS the integer), 5 rounds of
//          q = lo52(t0 * pinv), t += q * p with the Pasta modulus' sparse limbs, carry normalisation back to 52-bit limbs.
//          Verified on the host against a big-integer product for random inputs (the kernel writes its result back).
//   dp52 product only: the 25 split products + column sums without the reduction (an upper bound for any cleverer reduction)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../csrc -o ubench_dp52 ubench_dp52.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "field.cuh"
using namespace bzh;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef unsigned __int128 u128;
// Fp = 2^254 + T, T = 0x224698fc094cf91b992d30ed00000001 (src/chips/bitify.rs:461)
static const uint64_t kP[4] = {0x992d30ed00000001ull, 0x224698fc094cf91bull, 0ull, 0x4000000000000000ull};
#define MASK52 ((1ull << 52) - 1)

struct Dp5 { double l[5]; };

__device__ __forceinline__ long long bits(double d) { return __double_as_longlong(d); }

// exact hi / lo split of a * b for integers a, b < 2^52 held in doubles: hi is a multiple of 2^52, |lo| <= 2^51
#define C1 (20282409603651670423947251286016.0)         /* 2^104: a*b < 2^104 keeps C1 + a*b inside [2^104, 2^105) */
#define C2 (1.5 * 4503599627370496.0)                    /* 1.5 * 2^52  */
__device__ __forceinline__ void split_mul(double a, double b, long long& hi_units, long long& lo_units) {
    const double h = __fma_rn(a, b, C1);          // binade [2^104, 2^105): ulp 2^52
    const double hs = C1 - h;                     // -(hi), exact
    const double l = __fma_rn(a, b, hs) + C2;     // lo + 1.5 * 2^52, binade [2^52, 2^53): ulp 1
    hi_units = bits(h);                           // = bits(C1) + hi / 2^52
    lo_units = bits(l);                           // = bits(C2) + lo
}

// p in 52-bit limbs and -p^-1 mod 2^52, filled by the host
struct Consts { double p[5]; double pinv; long long c1bits, c2bits; };

__device__ __forceinline__ Dp5 dp52_mul(const Dp5& a, const Dp5& b, const Consts& K, bool reduce) {
    long long col[11];
#pragma unroll
    for (int i = 0; i < 11; i++) col[i] = 0;
#pragma unroll
    for (int i = 0; i < 5; i++)
#pragma unroll
        for (int j = 0; j < 5; j++) {
            long long h, l;
            split_mul(a.l[i], b.l[j], h, l);
            col[i + j] += l - K.c2bits;
            col[i + j + 1] += h - K.c1bits;
        }
    if (reduce) {
#pragma unroll
        for (int r = 0; r < 5; r++) {
            // q = (col[r] * pinv) mod 2^52: low 52 bits of a 52 x 52 product (col[r] first normalised to 52 bits)
            const long long t0 = col[r] & (long long)MASK52;
            col[r + 1] += col[r] >> 52;           // arithmetic shift: carries may be negative
            const double t0d = (double)t0;
            const double hq = __fma_rn(t0d, K.pinv, C1);
            const double lq = __fma_rn(t0d, K.pinv, C1 - hq) + C2;
            const long long q = (bits(lq) - K.c2bits) & (long long)MASK52;
            const double qd = (double)q;
            // t += q * p; limb 3 of p is 0 and limb 4 is 2^46: no multiplication for those
#pragma unroll
            for (int j = 0; j < 3; j++) {
                long long h, l;
                split_mul(qd, K.p[j], h, l);
                if (j == 0) col[r + 1] += (t0 + l - K.c2bits) >> 52;   // column r becomes 0 mod 2^52: only its carry survives
                else col[r + j] += l - K.c2bits;
                col[r + j + 1] += h - K.c1bits;
            }
            col[r + 4] += (q & 63) << 46;       // p limb 4 = 2^46: q * 2^46 = (q >> 6) * 2^52 + (q & 63) * 2^46
            col[r + 5] += q >> 6;
        }
    }
    Dp5 o;
    const int base = reduce ? 5 : 0;
    long long carry = 0;
#pragma unroll
    for (int i = 0; i < 5; i++) {
        const long long v = col[base + i] + carry;
        carry = v >> 52;
        o.l[i] = (double)(v & (long long)MASK52);
    }
    return o;   // < 2p (not conditionally subtracted: neither is the timing of the int path's chain dependent on it)
}

__global__ void __launch_bounds__(256) k_dp52_chain(double* io, Consts K, int iters, int reduce) {
    const size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    Dp5 x, y;
    for (int i = 0; i < 5; i++) x.l[i] = io[(g & 1023) * 5 + i], y.l[i] = io[((g + 7) & 1023) * 5 + i];
    for (int i = 0; i < iters; i++) x = dp52_mul(x, y, K, reduce != 0);
    if (iters == 1 || x.l[0] == 12345.0)
        for (int i = 0; i < 5; i++) io[(1024 + g) * 5 + i] = x.l[i];
}
template <class P>
__global__ void __launch_bounds__(256) k_int_chain(uint32_t* io, int iters) {
    const size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    Fe<P> x = fe_load<P>(io + (g & 1023) * 8), y = fe_load<P>(io + ((g + 7) & 1023) * 8);
    for (int i = 0; i < iters; i++) x = fe_mul(x, y);
    if (x.l[0] == 0x12345u) fe_store(io + (g & 1023) * 8, x);
}

// ---- host big-integer check -------------------------------------------------------------------------------------------
struct Big { uint64_t w[10]; };   // 640 bits
static Big big_from_limbs52(const double* l) {
    Big r; memset(&r, 0, sizeof r);
    for (int i = 0; i < 5; i++) {
        const uint64_t v = (uint64_t)l[i];
        const int sh = 52 * i, wi = sh / 64, bi = sh % 64;
        r.w[wi] |= v << bi;
        if (bi > 12) r.w[wi + 1] |= v >> (64 - bi);
    }
    return r;
}
static Big big_mul(const Big& a, const Big& b) {   // inputs < 2^262
    Big r; memset(&r, 0, sizeof r);
    for (int i = 0; i < 5; i++) {
        u128 c = 0;
        for (int j = 0; j < 5; j++) {
            c += (u128)a.w[i] * b.w[j] + r.w[i + j];
            r.w[i + j] = (uint64_t)c;
            c >>= 64;
        }
        r.w[i + 5] = (uint64_t)c;
    }
    return r;
}
static void big_mod_p(Big& x) {   // x mod p by shift-subtract (x < 2^524)
    Big p; memset(&p, 0, sizeof p);
    memcpy(p.w, kP, 32);
    auto ge = [](const Big& a, const Big& b) { for (int i = 9; i >= 0; i--) { if (a.w[i] != b.w[i]) return a.w[i] > b.w[i]; } return true; };
    auto sub = [](Big& a, const Big& b) { unsigned __int128 br = 0; for (int i = 0; i < 10; i++) { u128 d = (u128)a.w[i] - b.w[i] - (uint64_t)br; a.w[i] = (uint64_t)d; br = (d >> 64) & 1; } };
    auto shl1 = [](Big& a) { for (int i = 9; i > 0; i--) a.w[i] = (a.w[i] << 1) | (a.w[i - 1] >> 63); a.w[0] <<= 1; };
    auto shr1 = [](Big& a) { for (int i = 0; i < 9; i++) a.w[i] = (a.w[i] >> 1) | (a.w[i + 1] << 63); a.w[9] >>= 1; };
    int s = 0;
    while (!(p.w[9] >> 62)) { shl1(p); s++; }
    for (; s >= 0; s--) { if (ge(x, p)) sub(x, p); shr1(p); }
}

int main() {
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    // constants
    Consts K;
    {
        u128 lo = ((u128)kP[1] << 64) | kP[0];
        K.p[0] = (double)(uint64_t)(lo & MASK52);
        K.p[1] = (double)(uint64_t)((lo >> 52) & MASK52);
        K.p[2] = (double)(uint64_t)(lo >> 104);          // 24 bits
        K.p[3] = 0.0;
        K.p[4] = (double)(1ull << 46);                   // 2^254 / 2^208
        uint64_t p0 = (uint64_t)(lo & MASK52), inv = 1;  // -p^-1 mod 2^52 by Newton
        for (int i = 0; i < 6; i++) inv = inv * (2 - p0 * inv);
        K.pinv = (double)((0 - inv) & MASK52);
        const double c1 = C1, c2 = C2;
        memcpy(&K.c1bits, &c1, 8);
        memcpy(&K.c2bits, &c2, 8);
    }
    double* d;
    CHK(hipMalloc(&d, (1024 + 256 * cus * 8 * 2) * 5 * sizeof(double)));
    std::vector<double> h(1024 * 5);
    uint64_t s = 88172645463325252ull;
    for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (double)(s & MASK52); }
    for (int e = 0; e < 1024; e++) h[e * 5 + 4] = (double)((uint64_t)h[e * 5 + 4] & ((1ull << 45) - 1));   // < 2^253 < p
    CHK(hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    // correctness: one Montgomery product per lane against the host: out * 2^260 == x * y (mod p)
    hipLaunchKernelGGL(k_dp52_chain, dim3(4), dim3(256), 0, 0, d, K, 1, 1);
    CHK(hipDeviceSynchronize());
    std::vector<double> out(1024 * 5);
    CHK(hipMemcpy(out.data(), d + 1024 * 5, out.size() * 8, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int g = 0; g < 1024; g++) {
        Big x = big_from_limbs52(&h[(g & 1023) * 5]), y = big_from_limbs52(&h[((g + 7) & 1023) * 5]), o = big_from_limbs52(&out[g * 5]);
        Big xy = big_mul(x, y);
        big_mod_p(xy);
        Big r; memset(&r, 0, sizeof r);
        r.w[4] = 1ull << 4;                               // 2^260
        Big lhs = big_mul(o, r);
        big_mod_p(lhs);
        bad += memcmp(lhs.w, xy.w, sizeof lhs.w) != 0;
    }
    printf("dp52 Montgomery product vs host big integers: %d / 1024 mismatches\n", bad);
    uint32_t* di;
    CHK(hipMalloc(&di, 1024 * 64));
    std::vector<uint32_t> hi(1024 * 16);
    for (size_t i = 0; i < hi.size(); i++) hi[i] = (uint32_t)(i * 2654435761u) & 0x3fffffffu;
    CHK(hipMemcpy(di, hi.data(), hi.size() * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch, double ops_per_lane, int threads_per_cu) {
        launch();
        hipDeviceSynchronize();
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("%-52s %8.3f ms  %9.2f Gop/s  (%d lanes/CU)\n", name, ms, ops_per_lane * (double)threads_per_cu * cus / ms / 1e6, threads_per_cu);
    };
    const int it = 1024;
    for (int wps : {2, 4, 8}) {
        const int blocks = cus * wps;
        run("int  8x32: fe_mul<Fp> (shipped)", [&] { hipLaunchKernelGGL(k_int_chain<FpParams>, dim3(blocks), dim3(256), 0, 0, di, it); }, it, 256 * wps);
        run("dp52 5x52: Montgomery product, FP64 FMA", [&] { hipLaunchKernelGGL(k_dp52_chain, dim3(blocks), dim3(256), 0, 0, d, K, it, 1); }, it, 256 * wps);
        run("dp52 5x52: 25 split products only (no reduction)", [&] { hipLaunchKernelGGL(k_dp52_chain, dim3(blocks), dim3(256), 0, 0, d, K, it, 0); }, it, 256 * wps);
    }
    return bad != 0;
}
