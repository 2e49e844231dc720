// fallback.hip — search_direction_nonsymmetric! (src/solver/search_direction.jl:106-119): step = H \ residual on the UNREDUCED
// N x N matrix.  The reference takes this path when iterative refinement fails (search_direction.jl:22) and uses SparseArrays'
// `\` (UMFPACK, a partially pivoted sparse LU).  It is an exception path (no test problem of the reference reaches it), so the
// device counterpart favours robustness over speed: H is materialised densely from its block closed forms
// (residual_jacobian_variables.jl:1-108, the same blocks k_Hmul_vec applies matrix-free) and factored by a blocked partially
// pivoted LU.
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

// ---- dense LU with partial pivoting: blocked right-looking, panels of LUB columns -------------------------------------------------
// Per panel: (1) k_lu_panel — ONE workgroup factors the (N-k0) x LUB panel column by column (pivot search by a workgroup
// reduction, row swap inside the panel, scaling, rank-1 update of the remaining panel columns; the panel stays in L2);
// (2) k_lu_swaps — the panel's row interchanges applied to all other columns; (3) k_lu_trsm — U12 = L11^-1 A12, one lane per
// column; (4) A22 -= L21 U12 with the fp64 MFMA GEMM of gemm.hip.  The solve applies the interchanges to the right-hand side and
// runs block forward / backward substitutions (two launches per panel each way).
constexpr int LUB = 32;

__global__ __launch_bounds__(1024) void k_lu_panel(int N, int k0, int nb, double* __restrict__ A, int* __restrict__ piv, int* __restrict__ info) {
    __shared__ double sv[16];
    __shared__ int si[16];
    __shared__ int sp;
    __shared__ double rowc[LUB];
    const int tid = threadIdx.x, m = N - k0;
    double* P = A + k0 + (size_t)k0 * N;               // panel origin: P[i + j*N], i < m, j < nb
    for (int c = 0; c < nb; ++c) {
        double best = -1.0; int bi = c;
        for (int i = c + tid; i < m; i += 1024) { const double a = fabs(P[i + (size_t)c * N]); if (a > best) { best = a; bi = i; } }
        for (int off = 32; off > 0; off >>= 1) {
            const double ob = __shfl_down(best, off, 64); const int oi = __shfl_down(bi, off, 64);
            if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
        }
        if ((tid & 63) == 0) { sv[tid >> 6] = best; si[tid >> 6] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 16; ++w) if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
            sp = bi;
            piv[k0 + c] = k0 + bi;
            if (!(best > 0.0) && *info == 0) *info = k0 + c + 1;   // exactly singular (LAPACK convention: first zero pivot, 1-based)
        }
        __syncthreads();
        const int p = sp;
        if (p != c && tid < nb) { const double t = P[c + (size_t)tid * N]; P[c + (size_t)tid * N] = P[p + (size_t)tid * N]; P[p + (size_t)tid * N] = t; }
        __syncthreads();
        // scaling and the rank-1 update of the remaining panel columns in one pass: the lane that owns row i forms l_i and applies it
        const double d = P[c + (size_t)c * N];
        if (tid < nb) rowc[tid] = P[c + (size_t)tid * N];
        __syncthreads();
        if (d != 0.0)
            for (int i = c + 1 + tid; i < m; i += 1024) {
                const double l = P[i + (size_t)c * N] / d;
                P[i + (size_t)c * N] = l;
                for (int j = c + 1; j < nb; ++j) P[i + (size_t)j * N] -= l * rowc[j];
            }
        __syncthreads();
    }
}
// row interchanges of panel k0 applied to the columns outside the panel
__global__ void k_lu_swaps(int N, int k0, int nb, const int* __restrict__ piv, double* __restrict__ A) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N - nb) return;
    if (j >= k0) j += nb;                               // skip the panel's own columns
    double* col = A + (size_t)j * N;
    for (int c = 0; c < nb; ++c) {
        const int r = k0 + c, p = piv[r];
        if (p != r) { const double t = col[r]; col[r] = col[p]; col[p] = t; }
    }
}
// U12 = L11^-1 A12: one lane per column right of the panel
__global__ void k_lu_trsm(int N, int k0, int nb, double* __restrict__ A) {
    __shared__ double L[LUB * LUB];
    for (int e = threadIdx.x; e < nb * nb; e += blockDim.x) L[e] = A[(k0 + e % nb) + (size_t)(k0 + e / nb) * N];   // L[r + c*nb]
    __syncthreads();
    const int j = k0 + nb + blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= N) return;
    double u[LUB];
    double* col = A + k0 + (size_t)j * N;
#pragma unroll
    for (int r = 0; r < LUB; ++r) u[r] = r < nb ? col[r] : 0.0;
#pragma unroll
    for (int c = 0; c < LUB; ++c)
#pragma unroll
        for (int r = c + 1; r < LUB; ++r) if (r < nb) u[r] -= L[r + c * nb] * u[c];
#pragma unroll
    for (int r = 0; r < LUB; ++r) if (r < nb) col[r] = u[r];
}
__global__ void k_lu_permute(int N, const int* __restrict__ piv, double* __restrict__ b) {
    if (threadIdx.x == 0 && blockIdx.x == 0)
        for (int k = 0; k < N; ++k) { const int p = piv[k]; if (p != k) { const double t = b[k]; b[k] = b[p]; b[p] = t; } }
}
// block substitutions: diagonal block solved by one lane, the update by one lane per row
// (one wavefront: lane r holds b[r]; the pivot entry travels by shuffle, the block by LDS)
__global__ __launch_bounds__(64) void k_lu_fwd_block(int N, int k0, int nb, const double* __restrict__ A, double* __restrict__ b) {
    __shared__ double L[LUB * (LUB + 1)];
    const int r = threadIdx.x;
    for (int e = r; e < nb * nb; e += 64) L[(e % nb) * (LUB + 1) + e / nb] = A[(k0 + e % nb) + (size_t)(k0 + e / nb) * N];
    double v = r < nb ? b[k0 + r] : 0.0;
    __syncthreads();
    for (int c = 0; c < nb; ++c) {
        const double bc = __shfl(v, c, 64);
        if (r > c && r < nb) v -= L[r * (LUB + 1) + c] * bc;
    }
    if (r < nb) b[k0 + r] = v;
}
__global__ void k_lu_fwd_update(int N, int k0, int nb, const double* __restrict__ A, double* __restrict__ b) {
    const int i = k0 + nb + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    double acc = 0.0;
    for (int c = 0; c < nb; ++c) acc += A[i + (size_t)(k0 + c) * N] * b[k0 + c];
    b[i] -= acc;
}
__global__ __launch_bounds__(64) void k_lu_bwd_block(int N, int k0, int nb, const double* __restrict__ A, double* __restrict__ b) {
    __shared__ double U[LUB * (LUB + 1)];
    const int r = threadIdx.x;
    for (int e = r; e < nb * nb; e += 64) U[(e % nb) * (LUB + 1) + e / nb] = A[(k0 + e % nb) + (size_t)(k0 + e / nb) * N];
    double v = r < nb ? b[k0 + r] : 0.0;
    __syncthreads();
    for (int c = nb - 1; c >= 0; --c) {
        const double bc = __shfl(v, c, 64) / U[c * (LUB + 1) + c];
        if (r == c) v = bc;
        if (r < c) v -= U[r * (LUB + 1) + c] * bc;
    }
    if (r < nb) b[k0 + r] = v;
}
__global__ void k_lu_bwd_update(int N, int k0, int nb, const double* __restrict__ A, double* __restrict__ b) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k0) return;
    double acc = 0.0;
    for (int c = 0; c < nb; ++c) acc += A[i + (size_t)(k0 + c) * N] * b[k0 + c];
    b[i] -= acc;
}

// step = H \ res.  Returns CALIPSO_OK, CALIPSO_ERR_HIP (allocation) or CALIPSO_WARN_ZERO_PIVOT (singular H).
int nonsymmetric_solve(calipso_hip_solver* s, const double* res, double* step) {
    const Dims& d = s->d;
    const int N = d.N;
    if (!s->Hdense) {
        CK(hipMalloc((void**)&s->Hdense, sizeof(double) * (size_t)N * N));
        CK(hipMalloc((void**)&s->lu_ipiv, sizeof(int) * ((size_t)N + 1)));
    }
    const calipso::BatchSc* group = s->cur;
    s->cur = nullptr;                                   // this path works on the handle alone (gemm() consults batch_of)
    double* A = s->Hdense;
    double *Ltmp = nullptr, *Ztmp = nullptr;           // structured handle: dense temporaries of the blocks for the (rare) pivoted fallback
    if (s->compact) { const int urc = blocks_unpack_dense(s, &Ltmp, &Ztmp); if (urc < 0) { s->cur = group; if (Ltmp) (void)hipFree(Ltmp); if (Ztmp) (void)hipFree(Ztmp); return urc; } }
    hipLaunchKernelGGL(k_assemble_H, dim3((unsigned)((N + 255) / 256), (unsigned)N), dim3(256), 0, s->stream, d, s->sc, s->cone, s->compact ? Ltmp : s->Lxx, s->compact ? Ztmp : s->Z,
                       s->solution, A);
    if (s->compact) { (void)hipStreamSynchronize(s->stream); (void)hipFree(Ltmp); (void)hipFree(Ztmp); }
    if (step != res) (void)hipMemcpyAsync(step, res, sizeof(double) * N, hipMemcpyDeviceToDevice, s->stream);
    int* info = s->lu_ipiv + N;
    (void)hipMemsetAsync(info, 0, sizeof(int), s->stream);
    for (int k0 = 0; k0 < N; k0 += LUB) {
        const int nb = N - k0 < LUB ? N - k0 : LUB;
        hipLaunchKernelGGL(k_lu_panel, dim3(1), dim3(1024), 0, s->stream, N, k0, nb, A, s->lu_ipiv, info);
        if (N - nb > 0) hipLaunchKernelGGL(k_lu_swaps, dim3((N - nb + 255) / 256), dim3(256), 0, s->stream, N, k0, nb, s->lu_ipiv, A);
        const int rest = N - k0 - nb;
        if (rest > 0) {
            hipLaunchKernelGGL(k_lu_trsm, dim3((rest + 255) / 256), dim3(256), 0, s->stream, N, k0, nb, A);
            gemm(s, rest, rest, nb, -1.0, A + (k0 + nb) + (size_t)k0 * N, N, false, A + k0 + (size_t)(k0 + nb) * N, N, 1.0,
                 A + (k0 + nb) + (size_t)(k0 + nb) * N, N);
        }
    }
    hipLaunchKernelGGL(k_lu_permute, dim3(1), dim3(64), 0, s->stream, N, s->lu_ipiv, step);
    for (int k0 = 0; k0 < N; k0 += LUB) {              // L y = P b
        const int nb = N - k0 < LUB ? N - k0 : LUB, rest = N - k0 - nb;
        hipLaunchKernelGGL(k_lu_fwd_block, dim3(1), dim3(64), 0, s->stream, N, k0, nb, A, step);
        if (rest > 0) hipLaunchKernelGGL(k_lu_fwd_update, dim3((rest + 255) / 256), dim3(256), 0, s->stream, N, k0, nb, A, step);
    }
    for (int k0 = ((N - 1) / LUB) * LUB; k0 >= 0; k0 -= LUB) {   // U x = y
        const int nb = N - k0 < LUB ? N - k0 : LUB;
        hipLaunchKernelGGL(k_lu_bwd_block, dim3(1), dim3(64), 0, s->stream, N, k0, nb, A, step);
        if (k0 > 0) hipLaunchKernelGGL(k_lu_bwd_update, dim3((k0 + 255) / 256), dim3(256), 0, s->stream, N, k0, nb, A, step);
    }
    s->cur = group;
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
