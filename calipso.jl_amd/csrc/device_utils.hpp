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

// ---- wave64 reductions (a CDNA wavefront is 64 lanes) ------------------------------------------------
// The same reductions with the result in LANE 63 only, by data-parallel moves on the vector unit: four shifts inside the rows of 16 lanes, then two row broadcasts.
// __shfl_down above is two ds_bpermute through the LDS pipeline per step and each step waits for the one before (~0.4 us per sum): where a kernel is a chain of short
// dependent phases (the multifrontal sweeps, the small-problem solve! kernel) these are what to call.  (Another summation order: other bits than wave_sum.)
template <int CTRL, int ROW_MASK> __device__ __forceinline__ double dpp_moved(double v) {     // the value from the lane the control names; 0 where there is none or the row is masked
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, true), hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_sum_l63(double v) {
    v += dpp_moved<0x111, 0xf>(v);                                            // row_shr:1
    v += dpp_moved<0x112, 0xf>(v);                                            // row_shr:2
    v += dpp_moved<0x114, 0xf>(v);                                            // row_shr:4
    v += dpp_moved<0x118, 0xf>(v);                                            // row_shr:8   -> lane 15 of every row holds the row's sum
    v += dpp_moved<0x142, 0xa>(v);                                            // row_bcast:15 into rows 1 and 3
    v += dpp_moved<0x143, 0xc>(v);                                            // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    return v;
}
template <int CTRL, int ROW_MASK> __device__ __forceinline__ double dpp_moved_or_own(double v) {     // ... the lane's own value where there is none
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), CTRL, ROW_MASK, 0xf, false), hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_max_l63(double v) {
    v = fmax(v, dpp_moved_or_own<0x111, 0xf>(v));
    v = fmax(v, dpp_moved_or_own<0x112, 0xf>(v));
    v = fmax(v, dpp_moved_or_own<0x114, 0xf>(v));
    v = fmax(v, dpp_moved_or_own<0x118, 0xf>(v));
    v = fmax(v, dpp_moved_or_own<0x142, 0xa>(v));
    v = fmax(v, dpp_moved_or_own<0x143, 0xc>(v));
    return v;
}
// ... and in EVERY lane (what the callers that read lane 0, or broadcast it, expect): the total travels from lane 63 through a scalar register
__device__ __forceinline__ double from_lane63(double v) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
__device__ __forceinline__ double wave_sum(double v) { return from_lane63(wave_sum_l63(v)); }
__device__ __forceinline__ double wave_max(double v) { return from_lane63(wave_max_l63(v)); }
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
