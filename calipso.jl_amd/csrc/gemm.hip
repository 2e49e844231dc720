// gemm.hip — general fp64 GEMM on the matrix cores, used where the hot path has many right-hand sides at once:
// differentiate! (src/solver/differentiate.jl:29-58) solves one condensed system per parameter column; here all p columns go
// through the same factors together, so the mat-vecs and the block triangular solves become GEMMs / TRSMs.
//     C(M x N) = alpha * op(A)(M x K) * B(K x N) + beta * C          op(A) = A or A', everything column-major
// 64 x 64 tile per workgroup of 1024 threads (16 wavefronts, one 16 x 16 MFMA tile each), K staged through LDS in chunks of 32.
#include "internal.hpp"

#include <algorithm>
#include "device_utils.hpp"

namespace calipso {

typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int GK = 32, GLD = GK + 2;

__global__ __launch_bounds__(1024) void k_gemm(int M, int N, int K, double alpha, const double* __restrict__ A, int lda, int transA,
                                                const double* __restrict__ B, int ldb, double beta, double* __restrict__ C, int ldc) {
    __shared__ double As[64 * GLD];   // As[i][k]
    __shared__ double Bs[64 * GLD];   // Bs[j][k]
    const int i0 = blockIdx.x * 64, j0 = blockIdx.y * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 2, wj = wave & 3;
    const int fr = lane & 15, fk = lane >> 4;
    v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
    for (int k0 = 0; k0 < K; k0 += GK) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int idx = tid + it * 1024;                 // 64 x 32 elements per operand
            if (transA) {                                    // op(A)[i][k] = A[k + i*lda]: lanes along k
                const int kk = idx & 31, i = idx >> 5;
                As[i * GLD + kk] = (i0 + i < M && k0 + kk < K) ? A[(k0 + kk) + (size_t)(i0 + i) * lda] : 0.0;
            } else {                                         // A[i + k*lda]: lanes along i
                const int i = idx & 63, kk = idx >> 6;
                As[i * GLD + kk] = (i0 + i < M && k0 + kk < K) ? A[(i0 + i) + (size_t)(k0 + kk) * lda] : 0.0;
            }
            const int kb = idx & 31, j = idx >> 5;           // B[k + j*ldb]: lanes along k
            Bs[j * GLD + kb] = (j0 + j < N && k0 + kb < K) ? B[(k0 + kb) + (size_t)(j0 + j) * ldb] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < GK / 4; ++kk) {
            const double a = As[(wi * 16 + fr) * GLD + kk * 4 + fk];
            const double b = Bs[(wj * 16 + fr) * GLD + kk * 4 + fk];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, acc, 0, 0, 0);   // transposed: MFMA row <-> j, column <-> i
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int j = j0 + wj * 16 + fk + 4 * r, i = i0 + wi * 16 + fr;
        if (i < M && j < N) {
            double* c = C + i + (size_t)j * ldc;
            *c = (beta == 0.0) ? alpha * acc[r] : alpha * acc[r] + beta * *c;
        }
    }
}

void gemm(calipso_hip_solver* s, int M, int N, int K, double alpha, const double* A, int lda, bool transA, const double* B, int ldb, double beta,
          double* C, int ldc) {
    if (M <= 0 || N <= 0) return;
    hipLaunchKernelGGL(k_gemm, dim3((M + 63) / 64, (N + 63) / 64), dim3(1024), 0, s->stream, M, N, K, alpha, A, lda, transA ? 1 : 0, B, ldb, beta, C, ldc);
}

// rows of X (NP x p) scaled by 1/D: Z = D^-1 U
__global__ void k_scale_rows_by_dinv(int NP, int p, const double* __restrict__ Dx, const double* __restrict__ U, double* __restrict__ Z) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
    if (i < NP && j < p) Z[i + (size_t)j * NP] = U[i + (size_t)j * NP] / Dx[i];
}

// X (NP x p, ld NP) <- S^-1 X with the block factors of ldl.hip: forward  U_k = Tinv_k B_k ; B_rest -= L[rest,k] U_k ;
// Z = D^-1 U ; backward  V_k = Tinv_k' Z_k ; Z_above -= L[k,above]' V_k.  U and Z are NP x p scratch.
void trsm_multi(calipso_hip_solver* s, double* X, int p, double* U, double* Zm) {
    const int NP = s->d.NP, tb = trsv_block(NP, (int)s->solve_block), nb = (NP + tb - 1) / tb;      // the last block may be narrower (NP = 2560: 1024 + 1024 + 512)
    if (s->stage_parallel && s->spS) {        // the factor lives in the fronts of sparse.hip (calipso_hip_set_stage_parallel): all columns through the tree together
        const BatchSc bsc = batch_of(s);
        if (bsc.b.n == 1 && sparse_solve_inplace_multi(s->spS, s->stream, bsc.b.slot[0], X + bsc.b.delta[0], NP, p) == CALIPSO_OK) return;
        for (int j = 0; j < p; ++j) launch_trsv(s, X + (size_t)j * NP);
        return;
    }
    for (int kb = 0; kb < nb; ++kb) {
        const int k0 = kb * tb, w = std::min(tb, NP - k0);
        gemm(s, w, p, w, 1.0, s->Tinv + (size_t)kb * tb * tb, tb, false, X + k0, NP, 0.0, U + k0, NP);
        const int rest = NP - k0 - w;
        if (rest > 0) gemm(s, rest, p, w, -1.0, s->Lf + (k0 + w) + (size_t)k0 * NP, NP, false, U + k0, NP, 1.0, X + k0 + w, NP);
    }
    hipLaunchKernelGGL(k_scale_rows_by_dinv, dim3((NP + 255) / 256, p), dim3(256), 0, s->stream, NP, p, s->Dx, U, Zm);
    for (int kb = nb - 1; kb >= 0; --kb) {
        const int k0 = kb * tb, w = std::min(tb, NP - k0);
        gemm(s, w, p, w, 1.0, s->Tinv + (size_t)kb * tb * tb, tb, true, Zm + k0, NP, 0.0, X + k0, NP);
        if (k0 > 0) gemm(s, k0, p, w, -1.0, s->Lf + k0, NP, true, X + k0, NP, 1.0, Zm, NP);
    }
}

}  // namespace calipso
