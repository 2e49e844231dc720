// ldl.hip — blocked right-looking LDL^T (no pivoting) of the nx x nx Schur complement S produced by schur.hip, and the
// triangular solves with its factors.  Together with the closed-form constraint pivots of schur.hip this is the
// factorisation  P K P' = L D L'  of factorize!/QDLDL_factor! (linear_solver.jl:19-31, qdldl.jl:400-589) in the order
// [z | y | x]; S is quasi-definite's positive block, so every pivot must be > 0 for the inertia test (inertia.jl:7-11).
//
// Per panel of NB = 64 columns:
//   k_ldl_diag      one workgroup factors the 64 x 64 diagonal block in LDS (right-looking, one barrier pair per column),
//                   counts pivot signs (compute_inertia!, linear_solver.jl:33-44) and flags exact zeros (qdldl.jl:579)
//   k_ldl_panel     one lane per row below: y L11' = a  by substitution with L11 broadcast from LDS;  L21 = y / d,
//                   Y21 = y (= L21 * D) kept for the trailing update
//   k_ldl_trailing  A22 -= L21 * Y21'   on the fp64 matrix cores (v_mfma_f64_16x16x4_f64), one wavefront per 64 x 64
//                   tile of the lower triangle, operands straight from L2 (the two panels are 1.3 MB each)
// Triangular solves use the explicit inverses of the unit-lower diagonal blocks (k_invert_blocks), so each block step is
// two small mat-vecs instead of a 64-long dependent chain.
#include "internal.hpp"
#include "device_utils.hpp"

namespace calipso {

typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int LDP = NB + 1;

__global__ __launch_bounds__(256) void k_ldl_diag(int NP, int nx, int k0, double* __restrict__ S, double* __restrict__ Dx, int* __restrict__ icount) {
    __shared__ double A[NB * LDP];
    __shared__ double col_y[NB], col_l[NB];
    const int tid = threadIdx.x;
    const int i = tid & 63, kq = tid >> 6;
    for (int c = kq; c < NB; c += 4) A[i * LDP + c] = (i >= c) ? S[(k0 + i) + (size_t)(k0 + c) * NP] : 0.0;
    __syncthreads();
    for (int j = 0; j < NB; ++j) {
        const double dj = A[j * LDP + j];
        if (tid < NB && tid > j) {
            const double yv = A[tid * LDP + j];
            col_y[tid] = yv;
            col_l[tid] = yv / dj;
        }
        __syncthreads();
        for (int k = j + 1 + kq; k < NB; k += 4)
            if (i >= k) A[i * LDP + k] -= col_l[i] * col_y[k];
        if (tid < NB && tid > j) A[tid * LDP + j] = col_l[tid];
        __syncthreads();
    }
    for (int c = kq; c < NB; c += 4)
        if (i > c) S[(k0 + i) + (size_t)(k0 + c) * NP] = A[i * LDP + c];
    if (tid < NB) {
        const double dd = A[tid * LDP + tid];
        Dx[k0 + tid] = dd;
        int pos = 0, nonpos = 0, zero = 0;
        if (k0 + tid < nx) { pos = dd > 0.0; nonpos = dd <= 0.0; zero = dd == 0.0; }
        pos = wave_sum_i(pos); nonpos = wave_sum_i(nonpos); zero = wave_sum_i(zero);
        if (tid == 0) { atomicAdd(&icount[3], pos); atomicAdd(&icount[4], nonpos); atomicAdd(&icount[5], zero); }
    }
}

__global__ __launch_bounds__(256) void k_ldl_panel(int NP, int k0, double* __restrict__ S, const double* __restrict__ Dx, double* __restrict__ Y) {
    __shared__ double L11[NB * NB];   // L11[c*NB + k] = L[c][k], k < c
    __shared__ double dinv[NB];
    const int tid = threadIdx.x;
    for (int idx = tid; idx < NB * NB; idx += 256) {
        const int c = idx >> 6, k = idx & 63;
        L11[idx] = (k < c) ? S[(k0 + c) + (size_t)(k0 + k) * NP] : 0.0;
    }
    if (tid < NB) dinv[tid] = 1.0 / Dx[k0 + tid];
    __syncthreads();
    const int r = k0 + NB + blockIdx.x * 256 + tid;
    if (r >= NP) return;
    double a[NB];
#pragma unroll
    for (int c = 0; c < NB; ++c) a[c] = S[r + (size_t)(k0 + c) * NP];
#pragma unroll
    for (int c = 1; c < NB; ++c) {
        double y = a[c];
#pragma unroll
        for (int k = 0; k < c; ++k) y -= a[k] * L11[c * NB + k];
        a[c] = y;
    }
#pragma unroll
    for (int c = 0; c < NB; ++c) {
        Y[r + (size_t)c * NP] = a[c];
        S[r + (size_t)(k0 + c) * NP] = a[c] * dinv[c];
    }
}

// A22 -= L21 * Y21'  (lower triangle).  The tile is computed transposed (D[j][i] = sum_k Y[j,k] L[i,k]) so that the
// 16-lane fast index of the MFMA result maps to the contiguous (row) dimension of the column-major S.
__global__ __launch_bounds__(256) void k_ldl_trailing(int NP, int k0, double* __restrict__ S, const double* __restrict__ Y, int ntiles) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= ntiles) return;
    int ti = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    while (ti * (ti + 1) / 2 > t) --ti;
    const int tj = t - ti * (ti + 1) / 2;
    const int r0 = k0 + NB;
    const int i0 = r0 + ti * 64, j0 = r0 + tj * 64;
    const int fr = lane & 15, fk = lane >> 4;
    v4d acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
    const double* Lp = S + (size_t)k0 * NP;   // L21 lives in columns k0..k0+63 of S
#pragma unroll 4
    for (int kk = 0; kk < NB / 4; ++kk) {
        const int k = kk * 4 + fk;
        double ya[4], lb[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) ya[m] = Y[(j0 + m * 16 + fr) + (size_t)k * NP];
#pragma unroll
        for (int n = 0; n < 4; ++n) lb[n] = Lp[(i0 + n * 16 + fr) + (size_t)k * NP];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(ya[m], lb[n], acc[m][n], 0, 0, 0);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gj = j0 + m * 16 + fk + 4 * r;   // MFMA row  -> S column
                const int gi = i0 + n * 16 + fr;           // MFMA col  -> S row (contiguous)
                S[gi + (size_t)gj * NP] -= acc[m][n][r];
            }
}

// X = L_kk^-1 for every 64 x 64 unit-lower diagonal block (one wavefront per block, lane c builds column c)
__global__ __launch_bounds__(64) void k_invert_blocks(int NP, const double* __restrict__ S, double* __restrict__ Linv) {
    __shared__ double Ls[NB * LDP];
    __shared__ double Xs[NB * LDP];
    const int k0 = blockIdx.x * NB, c = threadIdx.x;
    for (int i = 0; i < NB; ++i) Ls[i * LDP + c] = (i > c) ? S[(k0 + i) + (size_t)(k0 + c) * NP] : 0.0;
    __syncthreads();
    for (int i = 0; i < NB; ++i) {
        double acc = (i == c) ? 1.0 : 0.0;
        for (int k = c; k < i; ++k) acc -= Ls[i * LDP + k] * Xs[k * LDP + c];
        Xs[i * LDP + c] = (i >= c) ? acc : 0.0;
    }
    double* out = Linv + (size_t)blockIdx.x * NB * NB;
    for (int i = 0; i < NB; ++i) out[i + c * NB] = Xs[i * LDP + c];
}

void launch_ldl(calipso_hip_solver* s) {
    const int NP = s->d.NP, nblk = NP / NB;
    for (int kb = 0; kb < nblk; ++kb) {
        const int k0 = kb * NB;
        hipLaunchKernelGGL(k_ldl_diag, dim3(1), dim3(256), 0, s->stream, NP, s->d.nx, k0, s->S, s->Dx, s->icount);
        const int rows = NP - k0 - NB;
        if (rows > 0) {
            hipLaunchKernelGGL(k_ldl_panel, dim3((rows + 255) / 256), dim3(256), 0, s->stream, NP, k0, s->S, s->Dx, s->Ypanel);
            const int ntr = rows / 64, ntiles = ntr * (ntr + 1) / 2;
            hipLaunchKernelGGL(k_ldl_trailing, dim3((ntiles + 3) / 4), dim3(256), 0, s->stream, NP, k0, s->S, s->Ypanel, ntiles);
        }
    }
    hipLaunchKernelGGL(k_invert_blocks, dim3(nblk), dim3(64), 0, s->stream, NP, s->S, s->Linv);
}

// ---- triangular solves -------------------------------------------------------------------------------------------------------
// forward step kb: every workgroup forms x_k = Linv_kk * b_k in LDS; workgroup 0 publishes it, workgroup w > 0 updates
// its own 64 rows  b_{kb+w} -= L[(kb+w), kb] x_k.  The diagonal scaling by 1/D is fused into the publish.
__global__ __launch_bounds__(256) void k_trsv_fwd(int NP, int kb, const double* __restrict__ S, const double* __restrict__ Linv,
                                                   const double* __restrict__ Dx, double* __restrict__ b, double* __restrict__ xf) {
    __shared__ double bs[NB], part[4][NB], xs[NB];
    const int tid = threadIdx.x, i = tid & 63, q = tid >> 6;
    const int k0 = kb * NB;
    if (tid < NB) bs[tid] = b[k0 + tid];
    __syncthreads();
    const double* Li = Linv + (size_t)kb * NB * NB;
    double acc = 0.0;
#pragma unroll 4
    for (int c = q * 16; c < q * 16 + 16; ++c) acc += Li[i + c * NB] * bs[c];
    part[q][i] = acc;
    __syncthreads();
    if (tid < NB) xs[tid] = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
    __syncthreads();
    if (blockIdx.x == 0) {
        if (tid < NB) xf[k0 + tid] = xs[tid] / Dx[k0 + tid];   // z = D^-1 (L^-1 b)
        return;
    }
    const int r0 = k0 + blockIdx.x * NB;
    acc = 0.0;
#pragma unroll 4
    for (int c = q * 16; c < q * 16 + 16; ++c) acc += S[(r0 + i) + (size_t)(k0 + c) * NP] * xs[c];
    part[q][i] = acc;
    __syncthreads();
    if (tid < NB) b[r0 + tid] -= (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
}

// backward step kb: v_k = Linv_kk' * z_k ; workgroup w > 0 updates block j = w-1:  z_j -= L[kb, j]' v_k
__global__ __launch_bounds__(256) void k_trsv_bwd(int NP, int kb, const double* __restrict__ S, const double* __restrict__ Linv,
                                                   double* __restrict__ z, double* __restrict__ v) {
    __shared__ double zs[NB], part[4][NB], vs[NB];
    const int tid = threadIdx.x, i = tid & 63, q = tid >> 6;
    const int k0 = kb * NB;
    if (tid < NB) zs[tid] = z[k0 + tid];
    __syncthreads();
    const double* Li = Linv + (size_t)kb * NB * NB;
    double acc = 0.0;
#pragma unroll 4
    for (int c = q * 16; c < q * 16 + 16; ++c) acc += Li[c + i * NB] * zs[c];   // (Linv')[i][c] = Linv[c][i]
    part[q][i] = acc;
    __syncthreads();
    if (tid < NB) vs[tid] = (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
    __syncthreads();
    if (blockIdx.x == 0) {
        if (tid < NB) v[k0 + tid] = vs[tid];
        return;
    }
    const int j0 = (blockIdx.x - 1) * NB;
    acc = 0.0;
#pragma unroll 4
    for (int c = q * 16; c < q * 16 + 16; ++c) acc += S[(k0 + c) + (size_t)(j0 + i) * NP] * vs[c];   // L[kb rows c, column j0+i]
    part[q][i] = acc;
    __syncthreads();
    if (tid < NB) z[j0 + tid] -= (part[0][tid] + part[1][tid]) + (part[2][tid] + part[3][tid]);
}

// x (length NP, padded entries zero) <- S^-1 x ; uses vtmp[0..2NP) as scratch
void launch_trsv(calipso_hip_solver* s, double* x) {
    const int NP = s->d.NP, nblk = NP / NB;
    double* zf = s->vtmp;   // forward result (already scaled by 1/D)
    for (int kb = 0; kb < nblk; ++kb)
        hipLaunchKernelGGL(k_trsv_fwd, dim3(nblk - kb), dim3(256), 0, s->stream, NP, kb, s->S, s->Linv, s->Dx, x, zf);
    for (int kb = nblk - 1; kb >= 0; --kb)
        hipLaunchKernelGGL(k_trsv_bwd, dim3(kb + 1), dim3(256), 0, s->stream, NP, kb, s->S, s->Linv, zf, x);
}

}  // namespace calipso
