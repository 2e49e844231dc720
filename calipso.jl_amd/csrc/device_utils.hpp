// device_utils.hpp — wave64 / workgroup reduction helpers and cone algebra shared by the kernels (gfx950).
#pragma once
#include <hip/hip_runtime.h>

namespace calipso {

// ---- instance addressing (internal.hpp: Batch): shift every per-instance pointer of a kernel to the slab of its instance -----
template <typename... P> __device__ __forceinline__ void inst_shift(const Batch& b, P&... p) {
    const long long o = b.delta[blockIdx.z];
    ((p += o), ...);
}
template <typename... P> __device__ __forceinline__ void inst_shift_i(const Batch& b, P&... p) {   // 4-byte element buffers
    const long long o = 2 * b.delta[blockIdx.z];
    ((p += o), ...);
}

// ---- wave64 reductions by DPP/shuffle (a CDNA wavefront is 64 lanes) ------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_down(v, off, 64));
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// workgroup reductions; `sm` must hold >= blockDim.x/64 doubles; result valid in thread 0 (deterministic order)
__device__ __forceinline__ double block_sum(double v, double* sm) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) sm[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) for (int i = 0; i < nw; ++i) r += sm[i];
    return r;
}
__device__ __forceinline__ double block_max(double v, double* sm) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) sm[w] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) for (int i = 0; i < nw; ++i) r = fmax(r, sm[i]);
    return r;
}

// exact inverse of arrow(u) applied to x (cones/second_order.jl:50-65), same operation order as the reference:
//   alpha = -1/u1^2 * |u2:|^2, beta = 1/(1+alpha), us = u2:/u1
//   x0_1 = x1 - us'x2: ; x1_2: = x2: - beta*us*x0_1 ; x2_1 = x1 - us'x1_2: ; out = x2/u1
// u, x, out are strided so the same routine serves registers-in-LDS and global memory.
__device__ __forceinline__ void arrow_inverse(int n, const double* u, const double* x, double* out) {
    double uu = 0.0;
    for (int i = 1; i < n; ++i) uu += u[i] * u[i];
    const double alpha = -1.0 / (u[0] * u[0]) * uu;
    const double beta = 1.0 / (1.0 + alpha);
    double d0 = 0.0;
    for (int i = 1; i < n; ++i) d0 += (u[i] / u[0]) * x[i];
    const double x0_1 = x[0] - d0;
    double d1 = 0.0;
    for (int i = 1; i < n; ++i) {
        const double v = x[i] - beta * ((u[i] / u[0]) * x0_1);
        out[i] = v;
        d1 += (u[i] / u[0]) * v;
    }
    const double x2_1 = x[0] - d1;
    out[0] = 1.0 / u[0] * x2_1;
    for (int i = 1; i < n; ++i) out[i] = 1.0 / u[0] * out[i];
}


// arrow_inverse for cones of dimension <= MAXD with every loop unrolled to constant indices (the arrays stay in registers): the operations and
// their order are those of arrow_inverse, so the result is the same to the bit
template <int MAXD>
__device__ __forceinline__ void arrow_inverse_small(int n, const double (&u)[MAXD], const double (&x)[MAXD], double (&out)[MAXD]) {
    double uu = 0.0;
#pragma unroll
    for (int i = 1; i < MAXD; ++i) if (i < n) uu += u[i] * u[i];
    const double alpha = -1.0 / (u[0] * u[0]) * uu;
    const double beta = 1.0 / (1.0 + alpha);
    double d0 = 0.0;
#pragma unroll
    for (int i = 1; i < MAXD; ++i) if (i < n) d0 += (u[i] / u[0]) * x[i];
    const double x0_1 = x[0] - d0;
    double d1 = 0.0;
#pragma unroll
    for (int i = 1; i < MAXD; ++i) if (i < n) {
        const double v = x[i] - beta * ((u[i] / u[0]) * x0_1);
        out[i] = v;
        d1 += (u[i] / u[0]) * v;
    }
    const double x2_1 = x[0] - d1;
    out[0] = 1.0 / u[0] * x2_1;
#pragma unroll
    for (int i = 1; i < MAXD; ++i) if (i < n) out[i] = 1.0 / u[0] * out[i];
}

}  // namespace calipso
