// fallback.hip — search_direction_nonsymmetric! (src/solver/search_direction.jl:106-119): step = H \ residual on the UNREDUCED
// N x N matrix.  The reference takes this path when iterative refinement fails (search_direction.jl:22) and uses SparseArrays'
// `\` (UMFPACK, a partially pivoted sparse LU).  It is an exception path (no test problem of the reference reaches it), so the
// device counterpart favours robustness over speed: H is materialised densely from its block closed forms
// (residual_jacobian_variables.jl:1-108, the same blocks k_Hmul_vec applies matrix-free) and factored by a plain partially
// pivoted LU (one column at a time).
#include "internal.hpp"
#include "device_utils.hpp"

namespace calipso {

// value of arrow(u)[k][c] (cone-local indices) for the cone layout: diagonal for nonnegative entries, arrow for second-order cones
__device__ __forceinline__ double arrow_entry(const ConeDev& cd, const double* __restrict__ u, int k, int c) {
    const int jk = cd.entry_soc[k], jc = cd.entry_soc[c];
    if (jk < 0 || jc < 0) return (k == c) ? u[k] : 0.0;
    if (jk != jc) return 0.0;
    const int st = cd.soc_start[jk];
    if (k == st) return u[c];
    if (c == st) return u[k];
    return (c == k) ? u[st] : 0.0;
}

// H (N x N, column-major) as residual_jacobian_variables! writes it, regularisation included
__global__ void k_assemble_H(Dims d, Scalars sc, ConeDev cd, const double* __restrict__ Lxx, const double* __restrict__ Z,
                             const double* __restrict__ w, double* __restrict__ H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;   // row (fast)
    const int j = blockIdx.y;                              // column
    if (i >= d.N) return;
    const int nx = d.nx, ne = d.ne, m = d.m;
    const int orr = d.orr(), os = d.os(), oy = d.oy(), oz = d.oz(), ot = d.ot();
    double v = 0.0;
    if (i < orr) {                                        // x rows
        if (j < orr) { v = Lxx[i + (size_t)j * nx]; if (i == j) v += sc.ep; }
        else if (j >= oy && j < oz) v = Z[(j - oy) + (size_t)i * m];             // gx'
        else if (j >= oz && j < ot) v = Z[ne + (j - oz) + (size_t)i * m];        // hx'
    } else if (i < os) {                                  // r rows
        const int k = i - orr;
        if (j == i) v = sc.rho + sc.ep;
        else if (j == oy + k) v = -1.0;
    } else if (i < oy) {                                  // s rows
        const int k = i - os;
        if (j == i) v = 0.0 + sc.ep;
        else if (j == oz + k) v = -1.0;
        else if (j == ot + k) v = -1.0;
    } else if (i < oz) {                                  // y rows
        const int k = i - oy;
        if (j < orr) v = Z[k + (size_t)j * m];
        else if (j == orr + k) v = -1.0;
        else if (j == i) v = 0.0 - sc.ed;
    } else if (i < ot) {                                  // z rows
        const int k = i - oz;
        if (j < orr) v = Z[ne + k + (size_t)j * m];
        else if (j == os + k) v = -1.0;
        else if (j == i) v = 0.0 - sc.ed;
    } else {                                              // t rows: d(s o t)/ds = arrow(t), d(s o t)/dt = arrow(s) - ed I
        const int k = i - ot;
        if (j >= os && j < oy) v = arrow_entry(cd, w + ot, k, j - os);
        else if (j >= ot) { v = arrow_entry(cd, w + os, k, j - ot); if (j == i) v -= sc.ed; }
    }
    H[i + (size_t)j * d.N] = v;
}

// ---- dense LU with partial pivoting (right-looking, one column at a time) -------------------------------------------------------
// An exception path: clarity over speed.  Per column k: pivot search (one workgroup), row swap across all columns, scaling of
// the column, rank-1 update of the trailing block (HBM-bound: sum_k 16 (N-k)^2 bytes ~ 3 TB at N = 8500, ~1 s).
__global__ __launch_bounds__(1024) void k_lu_pivot(int N, int k, const double* __restrict__ A, int* __restrict__ piv, int* __restrict__ info) {
    __shared__ double sv[16];
    __shared__ int si[16];
    const double* col = A + (size_t)k * N;
    double best = -1.0; int bi = k;
    for (int i = k + threadIdx.x; i < N; i += 1024) { const double a = fabs(col[i]); if (a > best) { best = a; bi = i; } }   // first max per lane
    for (int off = 32; off > 0; off >>= 1) {
        const double ob = __shfl_down(best, off, 64); const int oi = __shfl_down(bi, off, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
        piv[k] = bi;
        if (!(best > 0.0) && *info == 0) *info = k + 1;      // exactly singular (LAPACK convention: first zero pivot, 1-based)
    }
}
__global__ void k_lu_swap(int N, int k, const int* __restrict__ piv, double* __restrict__ A) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int p = piv[k];
    if (j >= N || p == k) return;
    double* c = A + (size_t)j * N;
    const double t = c[k]; c[k] = c[p]; c[p] = t;
}
__global__ void k_lu_scale(int N, int k, double* __restrict__ A) {
    const int i = k + 1 + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    double* col = A + (size_t)k * N;
    const double d = col[k];
    if (d != 0.0) col[i] = col[i] / d;
}
__global__ void k_lu_update(int N, int k, double* __restrict__ A) {
    const int i = k + 1 + blockIdx.x * blockDim.x + threadIdx.x;   // row (contiguous)
    const int j = k + 1 + blockIdx.y;                              // column
    if (i >= N) return;
    A[i + (size_t)j * N] -= A[i + (size_t)k * N] * A[k + (size_t)j * N];
}
// b <- U^-1 L^-1 P b with one workgroup (column-oriented substitutions; b stays in global memory, visible across the
// workgroup after each barrier)
__global__ __launch_bounds__(1024) void k_lu_solve(int N, const double* __restrict__ A, const int* __restrict__ piv, double* __restrict__ b) {
    if (threadIdx.x == 0)
        for (int k = 0; k < N; ++k) { const int p = piv[k]; if (p != k) { const double t = b[k]; b[k] = b[p]; b[p] = t; } }
    __syncthreads();
    for (int k = 0; k < N; ++k) {                 // L y = P b (unit lower)
        const double bk = b[k];
        const double* col = A + (size_t)k * N;
        for (int i = k + 1 + threadIdx.x; i < N; i += 1024) b[i] -= col[i] * bk;
        __syncthreads();
    }
    for (int k = N - 1; k >= 0; --k) {            // U x = y
        const double* col = A + (size_t)k * N;
        if (threadIdx.x == 0) b[k] = b[k] / col[k];
        __syncthreads();
        const double bk = b[k];
        for (int i = threadIdx.x; i < k; i += 1024) b[i] -= col[i] * bk;
        __syncthreads();
    }
}

// step = H \ res.  Returns CALIPSO_OK, CALIPSO_ERR_HIP (allocation) or CALIPSO_WARN_ZERO_PIVOT (singular H).
int nonsymmetric_solve(calipso_hip_solver* s, const double* res, double* step) {
    const Dims& d = s->d;
    const int N = d.N;
    if (!s->Hdense) {
        CK(hipMalloc((void**)&s->Hdense, sizeof(double) * (size_t)N * N));
        CK(hipMalloc((void**)&s->lu_ipiv, sizeof(int) * ((size_t)N + 1)));
    }
    hipLaunchKernelGGL(k_assemble_H, dim3((unsigned)((N + 255) / 256), (unsigned)N), dim3(256), 0, s->stream, d, s->sc, s->cone, s->Lxx, s->Z,
                       s->solution, s->Hdense);
    if (step != res) CK(hipMemcpyAsync(step, res, sizeof(double) * N, hipMemcpyDeviceToDevice, s->stream));
    int* info = s->lu_ipiv + N;
    CK(hipMemsetAsync(info, 0, sizeof(int), s->stream));
    for (int k = 0; k < N; ++k) {
        hipLaunchKernelGGL(k_lu_pivot, dim3(1), dim3(1024), 0, s->stream, N, k, s->Hdense, s->lu_ipiv, info);
        hipLaunchKernelGGL(k_lu_swap, dim3((N + 255) / 256), dim3(256), 0, s->stream, N, k, s->lu_ipiv, s->Hdense);
        const int rest = N - k - 1;
        if (rest > 0) {
            hipLaunchKernelGGL(k_lu_scale, dim3((rest + 255) / 256), dim3(256), 0, s->stream, N, k, s->Hdense);
            hipLaunchKernelGGL(k_lu_update, dim3((rest + 255) / 256, rest), dim3(256), 0, s->stream, N, k, s->Hdense);
        }
    }
    hipLaunchKernelGGL(k_lu_solve, dim3(1), dim3(1024), 0, s->stream, N, s->Hdense, s->lu_ipiv, step);
    int hinfo = 0;
    CK(hipMemcpyAsync(&hinfo, info, sizeof(int), hipMemcpyDeviceToHost, s->stream));
    CK(hipStreamSynchronize(s->stream));
    s->stats.fallbacks += 1;
    return hinfo == 0 ? CALIPSO_OK : CALIPSO_WARN_ZERO_PIVOT;
}

void nonsymmetric_release(calipso_hip_solver* s) {
    if (s->Hdense) { (void)hipFree(s->Hdense); s->Hdense = nullptr; }
    if (s->lu_ipiv) { (void)hipFree(s->lu_ipiv); s->lu_ipiv = nullptr; }
}

}  // namespace calipso
