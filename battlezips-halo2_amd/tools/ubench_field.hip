// Throughput of the device field / curve primitives in isolation (no memory traffic):
// dependent chains of fe_mul, fe_add and XYZZ mixed additions per lane.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../csrc -o ubench_field ubench_field.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "curve29.cuh"
using namespace bzh;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <class P>
__global__ void __launch_bounds__(256) k_mul_chain(uint32_t* io, int iters) {
    size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    Fe<P> x = fe_load<P>(io + (g & 1023) * 8), y = fe_load<P>(io + ((g + 7) & 1023) * 8);
    for (int i = 0; i < iters; i++) x = fe_mul(x, y);
    if (x.l[0] == 0x12345u) fe_store(io + (g & 1023) * 8, x);
}
template <class P>
__global__ void __launch_bounds__(256) k_mul_chain2(uint32_t* io, int iters) {  // two independent chains per lane
    size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    Fe<P> x = fe_load<P>(io + (g & 1023) * 8), y = fe_load<P>(io + ((g + 7) & 1023) * 8), z = y;
    for (int i = 0; i < iters; i++) { x = fe_mul(x, y); z = fe_mul(z, y); }
    if ((x.l[0] ^ z.l[0]) == 0x12345u) fe_store(io + (g & 1023) * 8, x);
}
template <class P>
__global__ void __launch_bounds__(256) k_add_chain(uint32_t* io, int iters) {
    size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    Fe<P> x = fe_load<P>(io + (g & 1023) * 8), y = fe_load<P>(io + ((g + 7) & 1023) * 8);
    for (int i = 0; i < iters; i++) { x = fe_add(x, y); y = fe_sub(y, x); }
    if (x.l[0] == 0x12345u) fe_store(io + (g & 1023) * 8, x);
}
template <class P, int TPB>
__global__ void __launch_bounds__(TPB) k_madd_chain(uint32_t* io, int iters) {
    size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    Affine<P> q;
    q.x = fe_load<P>(io + (g & 511) * 16);
    q.y = fe_load<P>(io + (g & 511) * 16 + 8);
    Xyzz<P> acc = xyzz_identity<P>();
    for (int i = 0; i < iters; i++) {
        xyzz_madd(acc, q);
        q.x.l[0] ^= acc.x.l[1];  // keep the operand changing (not a curve point any more; timing only)
    }
    if (acc.x.l[0] == 0x12345u) fe_store(io + (g & 1023) * 8, acc.x);
}

// ---- the unsaturated 9 x 29-bit representation (csrc/fe29.cuh, csrc/curve29.cuh) ----
template <class P>
__global__ void __launch_bounds__(256) k_mul29_chain(uint32_t* io, int iters) {
    size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const Fe29Consts<P> k = fe29_consts<P>();
    Fe29<P> x = fe29_mul(fe29_from_sat_x32(fe_load<P>(io + (g & 1023) * 8)), k.one), y = fe29_mul(fe29_from_sat_x32(fe_load<P>(io + ((g + 7) & 1023) * 8)), k.one);
    for (int i = 0; i < iters; i++) x = fe29_mul(x, y);
    if (x.l[0] == 0x12345u) io[g & 1023] = x.l[1];
}
template <class P>
__global__ void __launch_bounds__(256) k_sqr29_chain(uint32_t* io, int iters) {
    size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const Fe29Consts<P> k = fe29_consts<P>();
    Fe29<P> x = fe29_mul(fe29_from_sat_x32(fe_load<P>(io + (g & 1023) * 8)), k.one);
    for (int i = 0; i < iters; i++) x = fe29_sqr(x);
    if (x.l[0] == 0x12345u) io[g & 1023] = x.l[1];
}
template <class P, int TPB>
__global__ void __launch_bounds__(TPB) k_madd29_chain(uint32_t* io, int iters) {
    size_t g = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const Fe29Consts<P> k = fe29_consts<P>();
    Affine<P> q;
    q.x = fe_load<P>(io + (g & 511) * 16);
    q.y = fe_load<P>(io + (g & 511) * 16 + 8);
    Xyzz29<P> acc = xyzz29_identity<P>();
    for (int i = 0; i < iters; i++) {
        xyzz29_madd(acc, q, k);
        q.x.l[0] ^= acc.x.l[1] & 0xffffu;  // keep the operand changing (timing only)
    }
    if (acc.x.l[0] == 0x12345u) io[g & 1023] = acc.x.l[1];
}

int main() {
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    uint32_t* d;
    CHK(hipMalloc(&d, 1024 * 64));
    std::vector<uint32_t> h(1024 * 16);
    for (size_t i = 0; i < h.size(); i++) h[i] = (uint32_t)(i * 2654435761u) & 0x3fffffffu;
    CHK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
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
        double total = ops_per_lane * (double)threads_per_cu * cus;
        printf("%-44s %8.3f ms  %9.2f Gop/s  (%d lanes/CU)\n", name, ms, total / ms / 1e6, threads_per_cu);
    };
    const int it = 2048;
    for (int wps : {1, 2, 4, 8}) {
        int blocks = cus * wps;  // 256-thread blocks: wps waves per SIMD
        run("fe_mul<Fp> dependent chain", [&] { hipLaunchKernelGGL(k_mul_chain<FpParams>, dim3(blocks), dim3(256), 0, 0, d, it); }, it, 256 * wps);
        run("fe_mul<Fp> 2 chains/lane", [&] { hipLaunchKernelGGL(k_mul_chain2<FpParams>, dim3(blocks), dim3(256), 0, 0, d, it); }, 2.0 * it, 256 * wps);
        run("fe_mul<BnFr> dependent chain", [&] { hipLaunchKernelGGL(k_mul_chain<BnFrParams>, dim3(blocks), dim3(256), 0, 0, d, it); }, it, 256 * wps);
        run("fe_add+fe_sub<Fp> chain (2 ops/iter)", [&] { hipLaunchKernelGGL(k_add_chain<FpParams>, dim3(blocks), dim3(256), 0, 0, d, it); }, 2.0 * it, 256 * wps);
        run("fe29_mul<Fp> dependent chain", [&] { hipLaunchKernelGGL(k_mul29_chain<FpParams>, dim3(blocks), dim3(256), 0, 0, d, it); }, it, 256 * wps);
        run("fe29_sqr<Fp> dependent chain", [&] { hipLaunchKernelGGL(k_sqr29_chain<FpParams>, dim3(blocks), dim3(256), 0, 0, d, it); }, it, 256 * wps);
        run("xyzz29_madd<Fq> chain", [&] { hipLaunchKernelGGL((k_madd29_chain<FqParams, 256>), dim3(blocks), dim3(256), 0, 0, d, it / 8); }, it / 8, 256 * wps);
        run("xyzz_madd<Fq> chain", [&] { hipLaunchKernelGGL((k_madd_chain<FqParams, 256>), dim3(blocks), dim3(256), 0, 0, d, it / 8); }, it / 8, 256 * wps);
    }
    run("xyzz_madd<Fq> chain, 512-thread blocks x2/CU", [&] { hipLaunchKernelGGL((k_madd_chain<FqParams, 512>), dim3(cus * 2), dim3(512), 0, 0, d, it / 8); }, it / 8, 1024);
    return 0;
}
