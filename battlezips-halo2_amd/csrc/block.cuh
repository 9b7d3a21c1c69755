// Workgroup-level helpers shared by polyops.hip and ipa.hip.
#pragma once
#include "field.cuh"

namespace bzh {

// ---------------------------------------------------------------------------
// block-wide sum of one field element per thread (LDS tree), result in thread 0
// ---------------------------------------------------------------------------
template <class P>
__device__ __forceinline__ Fe<P> block_sum(Fe<P> v, uint4* sh /* 2*T */, int T) {
    const int tid = threadIdx.x;
    for (int s = T >> 1; s >= 1; s >>= 1) {
        __syncthreads();
        if (tid >= s && tid < 2 * s) {
            sh[tid] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
            sh[T + tid] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
        }
        __syncthreads();
        if (tid < s) {
            uint4 a = sh[tid + s], b = sh[T + tid + s];
            Fe<P> o;
            o.l[0] = a.x; o.l[1] = a.y; o.l[2] = a.z; o.l[3] = a.w;
            o.l[4] = b.x; o.l[5] = b.y; o.l[6] = b.z; o.l[7] = b.w;
            v = fe_add(v, o);
        }
    }
    return v;
}

}  // namespace bzh
