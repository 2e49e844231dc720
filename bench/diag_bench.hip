// diag_bench.hip — stand-alone harness used to tune the 64 x 64 diagonal-block LDL^T kernel (the sequential pivot chain that
// bounds the factorisation of the Schur complement).  Times variants on one block and checks L D L' = A, X L = I.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include <type_traits>
constexpr int NB = 64;

__device__ __forceinline__ double fast_rcp(double v) {   // v_rcp_f64 + 2 Newton steps (no div_scale/fixup: pivots are normal numbers)
    double r = __builtin_amdgcn_rcp(v);
    r = fma(fma(-v, r, 1.0), r, r);
    r = fma(fma(-v, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ void lds_barrier() {   // barrier that only waits for LDS traffic, not for global memory
    __builtin_amdgcn_s_waitcnt(0xc07f);           // lgkmcnt(0) only
    __builtin_amdgcn_s_barrier();
}

// WAVES wavefronts; lane i = row; wavefront cg owns columns k = cg + WAVES*c, c < NB/WAVES
template <int WAVES, bool WITH_X, bool FAST_RCP, bool RAW_BARRIER>
__global__ __launch_bounds__(WAVES * 64) void k_diag(int ld, double* __restrict__ S, double* __restrict__ Dx, double* __restrict__ Xout) {
    constexpr int CPW = NB / WAVES;
    __shared__ double colbuf[2][NB];
    __shared__ double xrow[2][NB];
    __shared__ double rinvbuf[2];
    __shared__ double dd[NB];
    const int tid = threadIdx.x, i = tid & 63, cg = tid >> 6;
    double a[CPW], x[CPW];
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int k = cg + WAVES * c;
        a[c] = (i >= k) ? S[i + (size_t)k * ld] : 0.0;
        x[c] = (i == k) ? 1.0 : 0.0;
    }
    if (cg == 0) {
        colbuf[0][i] = a[0];
        if (i == 0) { rinvbuf[0] = 1.0 / a[0]; dd[0] = a[0]; }
    }
    if (WITH_X && i == 0) {
#pragma unroll
        for (int c = 0; c < CPW; ++c) xrow[0][cg + WAVES * c] = x[c];
    }
#pragma unroll 1
    for (int jb = 0; jb < NB; jb += WAVES)
#pragma unroll
    for (int jj = 0; jj < WAVES; ++jj) {
        const int j = jb + jj;
        const int cur = jj & 1, nxt = cur ^ 1;
        if (RAW_BARRIER) lds_barrier(); else __syncthreads();
        double yk[CPW], xk[CPW];
        const double yi = colbuf[cur][i];
        const double rinv = rinvbuf[cur];
#pragma unroll
        for (int c = 0; c < CPW; ++c) { yk[c] = colbuf[cur][cg + WAVES * c]; if (WITH_X) xk[c] = xrow[cur][cg + WAVES * c]; }
        const double li = yi * rinv;
        const double lrow = (i > j) ? li : 0.0;
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
            const int k = cg + WAVES * c;
            const double la = (k > j && i >= k) ? li : 0.0;
            a[c] -= la * yk[c];
            if (WITH_X) { const double lx = (k <= j) ? lrow : 0.0; x[c] -= lx * xk[c]; }
        }
        if (j + 1 < NB) {
            if (cg == (jj + 1) % WAVES) {
                const int cs = (j + 1) / WAVES;
                double v = a[0];
#pragma unroll
                for (int c = 1; c < CPW; ++c) v = (cs == c) ? a[c] : v;
                colbuf[nxt][i] = v;
                if (i == j + 1) { rinvbuf[nxt] = FAST_RCP ? fast_rcp(v) : 1.0 / v; dd[j + 1] = v; }
            }
            if (WITH_X && i == j + 1) {
#pragma unroll
                for (int c = 0; c < CPW; ++c) xrow[nxt][cg + WAVES * c] = x[c];
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int k = cg + WAVES * c;
        if (i > k) S[i + (size_t)k * ld] = a[c] * (1.0 / dd[k]);
        if (WITH_X) Xout[i + k * NB] = (i >= k) ? x[c] : 0.0;
    }
    if (tid < NB) Dx[tid] = dd[tid];
}

// merged-register variant: column k lives in ONE register per lane: it holds A[:,k] until its pivot step k, then X[:,k].
// One LDS vector m[] per step: m[i] (i > j) = unscaled pivot column, m[k] (k <= j) = row j of X.
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_diag_merged(int ld, double* __restrict__ S, double* __restrict__ Dx, double* __restrict__ Xout) {
    constexpr int CPW = NB / WAVES;
    __shared__ double m[2][NB];
    __shared__ double rinvbuf[2];
    __shared__ double dd[NB];
    __shared__ double Lsave[NB * NB];   // Lsave[k*NB + i]: unscaled column k at its pivot step
    const int tid = threadIdx.x, i = tid & 63, cg = tid >> 6;
    double r[CPW];
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int k = cg + WAVES * c;
        r[c] = (i >= k) ? S[i + (size_t)k * ld] : 0.0;
    }
    if (cg == 0) {
        m[0][i] = r[0];
        Lsave[i] = r[0];
        if (i == 0) { rinvbuf[0] = 1.0 / r[0]; dd[0] = r[0]; }
    }
#pragma unroll 1
    for (int jb = 0; jb < NB; jb += WAVES)
#pragma unroll
    for (int jj = 0; jj < WAVES; ++jj) {
        const int j = jb + jj;
        const int cur = jj & 1, nxt = cur ^ 1;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        const double mi = m[cur][i];
        const double rinv = rinvbuf[cur];
        double mk[CPW];
#pragma unroll
        for (int c = 0; c < CPW; ++c) mk[c] = m[cur][cg + WAVES * c];
        const double lrow = (i > j) ? mi * rinv : 0.0;     // l_i for rows below the pivot, 0 for finished rows
        const int cj = j / WAVES;                          // register slot of the pivot column in its owner wavefront
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
            if (cg == jj && c == cj) r[c] = (i == j) ? 1.0 : -lrow;     // pivot column -> starts its life as X[:,j]
            else r[c] -= lrow * mk[c];
        }
        if (j + 1 < NB) {
            const int cs = (j + 1) / WAVES;
            if (cg == (jj + 1) % WAVES) {                  // owner of column j+1 publishes it (rows > j+1) and its pivot
                double v = r[0];
#pragma unroll
                for (int c = 1; c < CPW; ++c) v = (cs == c) ? r[c] : v;
                if (i > j + 1) m[nxt][i] = v;
                Lsave[(j + 1) * NB + i] = v;
                if (i == j + 1) { rinvbuf[nxt] = fast_rcp(v); dd[j + 1] = v; }
            }
            if (i == j + 1) {                              // row j+1 of X for the columns k <= j+1 this wavefront owns
#pragma unroll
                for (int c = 0; c < CPW; ++c) {
                    const int k = cg + WAVES * c;
                    if (k <= j) m[nxt][k] = r[c];
                    else if (k == j + 1) m[nxt][k] = 1.0;
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int k = cg + WAVES * c;
        if (i > k) S[i + (size_t)k * ld] = Lsave[k * NB + i] * (1.0 / dd[k]);
        Xout[i + k * NB] = (i >= k) ? r[c] : 0.0;
    }
    if (tid < NB) Dx[tid] = dd[tid];
}

// skeleton: only the synchronisation pattern of a column step (barrier + LDS write -> read), to measure its floor
template <int WAVES, int MODE>
__global__ __launch_bounds__(WAVES * 64) void k_skeleton(int ld, double* __restrict__ S, double* __restrict__ Dx, double* __restrict__ Xout) {
    __shared__ double m[2][NB];
    const int tid = threadIdx.x, i = tid & 63, cg = tid >> 6;
    double r = S[i + (size_t)(cg % NB) * ld];
    if (cg == 0) m[0][i] = r;
#pragma unroll 1
    for (int jb = 0; jb < NB; jb += WAVES)
#pragma unroll
    for (int jj = 0; jj < WAVES; ++jj) {
        const int cur = jj & 1, nxt = cur ^ 1;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        const double mi = m[cur][i];
        r = fma(mi, 0.5, r);
        if (MODE >= 1) r = r * fast_rcp(mi + 3.0);          // a reciprocal on the critical path (all lanes)
        if (cg == (jj + 1) % WAVES) m[nxt][i] = r;
    }
    Xout[tid % (NB * NB)] = r;
    if (tid < NB) { Dx[tid] = 1.0; }
}

// v3: LDL^T loop without the inverse; X = L^-1 afterwards by 16 x 16 wave-synchronous inversions + two merge levels in LDS
constexpr int LDD3 = NB + 1;
__global__ __launch_bounds__(1024) void k_diag_v3(int ld, double* __restrict__ S, double* __restrict__ Dx, double* __restrict__ Xout) {
    constexpr int WAVES = 16, CPW = 4;
    __shared__ double colbuf[2][NB];
    __shared__ double rinvbuf[2];
    __shared__ double dd[NB];
    __shared__ double Ls[NB * LDD3];
    __shared__ double Xs[NB * LDD3];
    __shared__ double Ts[32 * 33];
    const int tid = threadIdx.x, i = tid & 63, cg = tid >> 6;
    double a[CPW];
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int k = cg + WAVES * c;
        a[c] = (i >= k) ? S[i + (size_t)k * ld] : 0.0;
    }
    if (cg == 0) {
        colbuf[0][i] = a[0];
        if (i == 0) { rinvbuf[0] = fast_rcp(a[0]); dd[0] = a[0]; }
    }
#pragma unroll 1
    for (int jb = 0; jb < NB; jb += WAVES)
#pragma unroll
    for (int jj = 0; jj < WAVES; ++jj) {
        const int j = jb + jj;
        const int cur = jj & 1, nxt = cur ^ 1;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        const double yi = colbuf[cur][i];
        const double rinv = rinvbuf[cur];
        double yk[CPW];
#pragma unroll
        for (int c = 0; c < CPW; ++c) yk[c] = colbuf[cur][cg + WAVES * c];
        const double li = (i > j) ? yi * rinv : 0.0;      // rows at/above the pivot: no update (upper part is never read)
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
            const int k = cg + WAVES * c;
            a[c] -= ((k > j) ? li : 0.0) * yk[c];
        }
        if (j + 1 < NB && cg == (jj + 1) % WAVES) {
            const int cs = (j + 1) / WAVES;
            double v = a[0];
#pragma unroll
            for (int c = 1; c < CPW; ++c) v = (cs == c) ? a[c] : v;
            colbuf[nxt][i] = v;
            if (i == j + 1) { rinvbuf[nxt] = fast_rcp(v); dd[j + 1] = v; }
        }
    }
    __syncthreads();
    // scaled L into LDS (for the inverse) and to global
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int k = cg + WAVES * c;
        const double l = (i > k) ? a[c] * (1.0 / dd[k]) : 0.0;
        Ls[i * LDD3 + k] = l;
        Xs[i * LDD3 + k] = 0.0;
        if (i > k) S[i + (size_t)k * ld] = l;
    }
    __syncthreads();
    // (a) the four 16 x 16 diagonal blocks of X: wave w, lane c < 16 builds column c by forward substitution (registers, no barrier)
    if (cg < 4 && i < 16) {
        const int o = 16 * cg;
        double x[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            double acc = (r == i) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < r; ++k) acc -= Ls[(o + r) * LDD3 + o + k] * x[k];
            x[r] = (r >= i) ? acc : 0.0;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) Xs[(o + r) * LDD3 + o + i] = x[r];
    }
    __syncthreads();
    // (b) level 1: X21 = -X22 (L21 X11) for the two 32-blocks; 512 threads, one output each
    {
        const int p = tid >> 8, ii = (tid >> 4) & 15, jj2 = tid & 15, o = 32 * p;
        double t = 0.0;
        if (tid < 512) {
#pragma unroll
            for (int k = 0; k < 16; ++k) t += Ls[(o + 16 + ii) * LDD3 + o + k] * Xs[(o + k) * LDD3 + o + jj2];
            Ts[(p * 16 + ii) * 33 + jj2] = t;
        }
        __syncthreads();
        if (tid < 512) {
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < 16; ++k) v -= Xs[(o + 16 + ii) * LDD3 + o + 16 + k] * Ts[(p * 16 + k) * 33 + jj2];
            Xs[(o + 16 + ii) * LDD3 + o + jj2] = v;
        }
        __syncthreads();
    }
    // (c) level 2: X21 (32 x 32) = -X22 (L21 X11); 1024 threads, one output each
    {
        const int ii = tid >> 5, jj2 = tid & 31;
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 32; ++k) t += Ls[(32 + ii) * LDD3 + k] * Xs[k * LDD3 + jj2];
        Ts[ii * 33 + jj2] = t;
        __syncthreads();
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < 32; ++k) v -= Xs[(32 + ii) * LDD3 + 32 + k] * Ts[k * 33 + jj2];
        __syncthreads();
        Xs[(32 + ii) * LDD3 + jj2] = v;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int k = cg + WAVES * c;
        Xout[i + k * NB] = Xs[i * LDD3 + k];
    }
    if (tid < NB) Dx[tid] = dd[tid];
}

// v4 (v3 + unmasked column updates, finished columns stashed in LDS, per-lane reciprocal vector): LDL^T loop without the inverse; X = L^-1 afterwards by 16 x 16 wave-synchronous inversions + two merge levels in LDS

__global__ __launch_bounds__(1024) void k_diag_v4(int ld, double* __restrict__ S, double* __restrict__ Dx, double* __restrict__ Xout) {
    constexpr int WAVES = 16, CPW = 4;
    __shared__ double colbuf[2][NB];
    __shared__ double rinvvec[2][NB];
    __shared__ double Ls[NB * LDD3];
    __shared__ double Xs[NB * LDD3];
    __shared__ double Ts[32 * 33];
    const int tid = threadIdx.x, i = tid & 63, cg = tid >> 6;
    double a[CPW];
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int k = cg + WAVES * c;
        a[c] = (i >= k) ? S[i + (size_t)k * ld] : 0.0;
    }
    if (cg == 0) {
        colbuf[0][i] = a[0];
        rinvvec[0][i] = fast_rcp(a[0]);
        Ls[i * LDD3 + 0] = a[0];
    }
#pragma unroll 1
    for (int jb = 0; jb < NB; jb += WAVES)
#pragma unroll
    for (int jj = 0; jj < WAVES; ++jj) {
        const int j = jb + jj;
        const int cur = jj & 1, nxt = cur ^ 1;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        const double yi = colbuf[cur][i];
        const double rinv = rinvvec[cur][j];
        double yk[CPW];
#pragma unroll
        for (int c = 0; c < CPW; ++c) yk[c] = colbuf[cur][cg + WAVES * c];
        const double li = (i > j) ? yi * rinv : 0.0;      // rows at/above the pivot: no update (upper part is never read)
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
            const int k = cg + WAVES * c;
            a[c] -= li * yk[c];          // finished columns (k <= j) were stashed in Ls when published; garbage here is never read
            (void)k;
        }
        if (j + 1 < NB && cg == (jj + 1) % WAVES) {
            const int cs = (j + 1) / WAVES;
            double v = a[0];
#pragma unroll
            for (int c = 1; c < CPW; ++c) v = (cs == c) ? a[c] : v;
            colbuf[nxt][i] = v;
            rinvvec[nxt][i] = fast_rcp(v);           // every lane: readers pick entry j+1 (no divergent single-lane path)
            Ls[i * LDD3 + j + 1] = v;                // unscaled column j+1 (rows >= j+1 meaningful), pivot on the diagonal
        }
    }
    __syncthreads();
    // scale the stashed columns: L into LDS (for the inverse) and to global; pivots out
    double dk[CPW];
#pragma unroll
    for (int c = 0; c < CPW; ++c) dk[c] = Ls[(cg + WAVES * c) * LDD3 + cg + WAVES * c];
    if (tid < NB) Dx[tid] = Ls[tid * LDD3 + tid];
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int k = cg + WAVES * c;
        const double l = (i > k) ? Ls[i * LDD3 + k] * (1.0 / dk[c]) : 0.0;
        Ls[i * LDD3 + k] = l;
        Xs[i * LDD3 + k] = 0.0;
        if (i > k) S[i + (size_t)k * ld] = l;
    }
    __syncthreads();
    // (a) the four 16 x 16 diagonal blocks of X: wave w, lane c < 16 builds column c by forward substitution (registers, no barrier)
    if (cg < 4 && i < 16) {
        const int o = 16 * cg;
        double x[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            double acc = (r == i) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < r; ++k) acc -= Ls[(o + r) * LDD3 + o + k] * x[k];
            x[r] = (r >= i) ? acc : 0.0;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) Xs[(o + r) * LDD3 + o + i] = x[r];
    }
    __syncthreads();
    // (b) level 1: X21 = -X22 (L21 X11) for the two 32-blocks; 512 threads, one output each
    {
        const int p = tid >> 8, ii = (tid >> 4) & 15, jj2 = tid & 15, o = 32 * p;
        double t = 0.0;
        if (tid < 512) {
#pragma unroll
            for (int k = 0; k < 16; ++k) t += Ls[(o + 16 + ii) * LDD3 + o + k] * Xs[(o + k) * LDD3 + o + jj2];
            Ts[(p * 16 + ii) * 33 + jj2] = t;
        }
        __syncthreads();
        if (tid < 512) {
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < 16; ++k) v -= Xs[(o + 16 + ii) * LDD3 + o + 16 + k] * Ts[(p * 16 + k) * 33 + jj2];
            Xs[(o + 16 + ii) * LDD3 + o + jj2] = v;
        }
        __syncthreads();
    }
    // (c) level 2: X21 (32 x 32) = -X22 (L21 X11); 1024 threads, one output each
    {
        const int ii = tid >> 5, jj2 = tid & 31;
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 32; ++k) t += Ls[(32 + ii) * LDD3 + k] * Xs[k * LDD3 + jj2];
        Ts[ii * 33 + jj2] = t;
        __syncthreads();
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < 32; ++k) v -= Xs[(32 + ii) * LDD3 + 32 + k] * Ts[k * 33 + jj2];
        __syncthreads();
        Xs[(32 + ii) * LDD3 + jj2] = v;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int k = cg + WAVES * c;
        Xout[i + k * NB] = Xs[i * LDD3 + k];
    }
}


// v6: two-level blocking.  The 64 columns are four 16-column sub-panels.  ONE wavefront factors a sub-panel without any barrier: lane = row
// (64 rows x 16 columns, 16 registers per lane), the pivot row travels by v_readlane; the rank-16 update of the columns to the right is
// done by all 16 wavefronts out of LDS.  8 workgroup barriers per 64 columns instead of 64.  X = L^-1 as in v4 (16 x 16 inversions + merges).
__device__ __forceinline__ double readlane_d(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}
__device__ long long g_ts[32];
#define TS(n) do { if (tid == 0) g_ts[n] = wall_clock64(); } while (0)
__global__ __launch_bounds__(1024) void k_diag_v6(int ld, double* __restrict__ S, double* __restrict__ Dx, double* __restrict__ Xout) {
    constexpr int WAVES = 16, CPW = 4, SP = 16;
    __shared__ double As[NB * LDD3];    // working matrix, row-major As[i][k]; ends as L (strictly lower) with the pivots in dd
    __shared__ double Xs[NB * LDD3];
    __shared__ double Ts[32 * 33];
    __shared__ double dd[NB];
    const int tid = threadIdx.x, i = tid & 63, cg = tid >> 6;
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int k = cg + WAVES * c;
        As[i * LDD3 + k] = (i >= k) ? S[i + (size_t)k * ld] : 0.0;
        Xs[i * LDD3 + k] = 0.0;
    }
    TS(0);
    __syncthreads();
    TS(1);
#pragma unroll 1
    for (int p = 0; p < NB / SP; ++p) {
        const int c0 = SP * p;
        TS(2 + 3 * p);
        if (cg == 0) {
            // lane i = row i (rows >= c0 take part); a[k] = A[i][c0 + k]; the 16 x 16 diagonal sub-block is held with BOTH triangles
            double a[SP];
#pragma unroll
            for (int k = 0; k < SP; ++k) {
                const int col = c0 + k;
                a[k] = (i >= col) ? As[i * LDD3 + col] : ((i >= c0) ? As[col * LDD3 + i] : 0.0);
            }
#pragma unroll
            for (int j = 0; j < SP; ++j) {
                const int r = c0 + j;
                const double d = readlane_d(a[j], r);
                const double rinv = fast_rcp(d);
                const double li = a[j] * rinv;
#pragma unroll
                for (int k = j + 1; k < SP; ++k) {
                    const double yk = readlane_d(a[k], r);       // A[r][c0 + k] (symmetric sub-block) = unscaled pivot-column entry of row c0 + k
                    a[k] -= li * yk;                             // rows <= r collect garbage that is never read
                }
                a[j] = (i > r) ? li : a[j];                      // L below the pivot, the pivot itself stays in lane r
            }
#pragma unroll
            for (int k = 0; k < SP; ++k) {
                const int col = c0 + k;
                if (i > col) As[i * LDD3 + col] = a[k];
                if (i == col) dd[col] = a[k];
            }
        }
        TS(3 + 3 * p);
        __syncthreads();
        TS(4 + 3 * p);
        // rank-16 update of the square to the right / below: A[i][k] -= sum_q L[i][q] d_q L[k][q]   (i >= k >= c0 + 16)
        const int m = NB - c0 - SP;
        for (int e = tid; e < m * m; e += 1024) {
            const int ii = c0 + SP + e / m, kk = c0 + SP + e % m;
            if (ii >= kk) {
                double acc = 0.0;
#pragma unroll
                for (int q = 0; q < SP; ++q) acc += As[ii * LDD3 + c0 + q] * (As[kk * LDD3 + c0 + q] * dd[c0 + q]);
                As[ii * LDD3 + kk] -= acc;
            }
        }
        __syncthreads();
    }
    TS(14);
    if (tid < NB) Dx[tid] = dd[tid];
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int k = cg + WAVES * c;
        if (i > k) S[i + (size_t)k * ld] = As[i * LDD3 + k];
    }
    TS(15);
    // (a) the four 16 x 16 diagonal blocks of X
    if (cg < 4 && i < 16) {
        const int o = 16 * cg;
        double x[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            double acc = (r == i) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < r; ++k) acc -= As[(o + r) * LDD3 + o + k] * x[k];
            x[r] = (r >= i) ? acc : 0.0;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) Xs[(o + r) * LDD3 + o + i] = x[r];
    }
    __syncthreads();
    {
        const int p = tid >> 8, ii = (tid >> 4) & 15, jj2 = tid & 15, o = 32 * p;
        double t = 0.0;
        if (tid < 512) {
#pragma unroll
            for (int k = 0; k < 16; ++k) t += As[(o + 16 + ii) * LDD3 + o + k] * Xs[(o + k) * LDD3 + o + jj2];
            Ts[(p * 16 + ii) * 33 + jj2] = t;
        }
        __syncthreads();
        if (tid < 512) {
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < 16; ++k) v -= Xs[(o + 16 + ii) * LDD3 + o + 16 + k] * Ts[(p * 16 + k) * 33 + jj2];
            Xs[(o + 16 + ii) * LDD3 + o + jj2] = v;
        }
        __syncthreads();
    }
    {
        const int ii = tid >> 5, jj2 = tid & 31;
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 32; ++k) t += As[(32 + ii) * LDD3 + k] * Xs[k * LDD3 + jj2];
        Ts[ii * 33 + jj2] = t;
        __syncthreads();
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < 32; ++k) v -= Xs[(32 + ii) * LDD3 + 32 + k] * Ts[k * 33 + jj2];
        __syncthreads();
        Xs[(32 + ii) * LDD3 + jj2] = v;
    }
    __syncthreads();
    TS(16);
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int k = cg + WAVES * c;
        Xout[i + k * NB] = Xs[i * LDD3 + k];
    }
    TS(17);
}


// v7: recursion at 16 columns with the pivot row broadcast by DPP.  Per 16-column stage:
//   (A) wavefront 0 factors the 16 x 16 diagonal sub-block: lane & 15 = row, both triangles in 16 registers, the pivot row of every
//       step reaches the other lanes of the 16-lane DPP row by row_newbcast (plain VALU, no SGPR round trip, no barrier);
//   (B) the same wavefront then solves the rows below (lane = row): Y21 = A21 L11^-T, L21 = Y21 D^-1, with the L11 entries read from LDS
//       as broadcasts;
//   (C) all wavefronts apply the rank-16 update to the rest of the block on the matrix cores (one 16 x 16 tile per wavefront).
// Two workgroup barriers per 16 columns.  The 16 x 16 inverses of X = L^-1 are formed by wavefront 1 while wavefront 0 works on the next stage.
typedef double v4d_ __attribute__((ext_vector_type(4)));
template <int N> __device__ __forceinline__ double bcast16(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + N, 0xf, 0xf, false);    // row_newbcast:N
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + N, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int J> struct Step16 {
    static __device__ __forceinline__ void run(double (&a)[16], int r16) {
        const double d = bcast16<J>(a[J]);
        const double rinv = fast_rcp(d);
        const double li = a[J] * rinv;
        step_cols<J + 1>(a, li);
        a[J] = (r16 > J) ? li : a[J];
        Step16<J + 1>::run(a, r16);
    }
    template <int K> static __device__ __forceinline__ void step_cols(double (&a)[16], double li) {
        if constexpr (K < 16) { a[K] -= li * bcast16<J>(a[K]); step_cols<K + 1>(a, li); }
    }
};
template <> struct Step16<16> { static __device__ __forceinline__ void run(double (&)[16], int) {} };

// inverse of the unit-lower 16 x 16 block at (o, o) of L (row-major, ld LDD3): lane c < 16 builds column c by forward substitution
__device__ __forceinline__ void inv16(const double* __restrict__ Lm, double* __restrict__ Xm, int o, int c) {
    double x[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        double acc = (r == c) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < r; ++k) acc -= Lm[(o + r) * LDD3 + o + k] * x[k];
        x[r] = (r >= c) ? acc : 0.0;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) Xm[(o + r) * LDD3 + o + c] = x[r];
}

__global__ __launch_bounds__(1024) void k_diag_v7(int ld, double* __restrict__ S, double* __restrict__ Dx, double* __restrict__ Xout) {
    constexpr int WAVES = 16, CPW = 4, SP = 16;
    __shared__ double As[NB * LDD3];    // working matrix, row-major As[i][k]; ends as L (strictly lower)
    __shared__ double Xs[NB * LDD3];
    __shared__ double Ts[32 * 33];      // Y panel (48 x 16, ld 17) during the stages, merge scratch afterwards
    __shared__ double dd[NB], rd[NB];
    const int tid = threadIdx.x, i = tid & 63, cg = tid >> 6;
    const int fr = i & 15, fk = i >> 4;
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int k = cg + WAVES * c;
        As[i * LDD3 + k] = (i >= k) ? S[i + (size_t)k * ld] : 0.0;
        Xs[i * LDD3 + k] = 0.0;
    }
    TS(0);
    __syncthreads();
#pragma unroll 1
    for (int p = 0; p < NB / SP; ++p) {
        const int c0 = SP * p;
        const int m = NB - c0 - SP;                  // rows / columns to the right of this stage
        TS(1 + 3 * p);
        if (cg == 0) {
            // (A) diagonal sub-block
            const int row = c0 + fr;
            double a[SP];
#pragma unroll
            for (int k = 0; k < SP; ++k) a[k] = (fr >= k) ? As[row * LDD3 + c0 + k] : As[(c0 + k) * LDD3 + row];
            Step16<0>::run(a, fr);
            if (i < SP) {
#pragma unroll
                for (int k = 0; k < SP; ++k) {
                    if (fr > k) As[row * LDD3 + c0 + k] = a[k];
                    if (fr == k) { dd[row] = a[k]; rd[row] = fast_rcp(a[k]); }
                }
            }
            // (B) rows below: lane = row c0 + 16 + i
            const int rb = c0 + SP + i;
            if (i < m) {
                double y[SP];
#pragma unroll
                for (int k = 0; k < SP; ++k) y[k] = As[rb * LDD3 + c0 + k];
#pragma unroll
                for (int k = 1; k < SP; ++k) {
#pragma unroll
                    for (int q = 0; q < k; ++q) y[k] -= y[q] * As[(c0 + k) * LDD3 + c0 + q];     // uniform address: LDS broadcast
                }
#pragma unroll
                for (int k = 0; k < SP; ++k) { Ts[i * 17 + k] = y[k]; As[rb * LDD3 + c0 + k] = y[k] * rd[c0 + k]; }
            }
        } else if (cg == 1 && p > 0 && i < 16) {
            inv16(As, Xs, c0 - SP, i);               // inverse of the previous diagonal sub-block (final since the last barrier)
        }
        TS(2 + 3 * p);
        __syncthreads();
        // (C) A22 -= L21 Y21' on the matrix cores: lower tiles (ti >= tj) of the (m/16)^2 tiling, one per wavefront
        {
            const int nt = m / 16;
            int ti = 0, tj = 0, t = cg;
            bool have = false;
            for (int a_ = 0; a_ < nt && !have; ++a_) { if (t <= a_) { ti = a_; tj = t; have = true; } else t -= a_ + 1; }
            if (have) {
                const int r0 = c0 + SP + 16 * ti, j0 = c0 + SP + 16 * tj;
                v4d_ acc;
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = As[(r0 + fk + 4 * r) * LDD3 + j0 + fr];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const double av = -As[(r0 + fr) * LDD3 + c0 + 4 * kk + fk];                  // A[i = fr][k = fk] = -L
                    const double bv = Ts[(16 * tj + fr) * 17 + 4 * kk + fk];                     // B[k = fk][j = fr] = Y[j][k]
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) As[(r0 + fk + 4 * r) * LDD3 + j0 + fr] = acc[r];
            }
        }
        TS(3 + 3 * p);
        __syncthreads();
    }
    TS(13);
    if (tid < NB) Dx[tid] = dd[tid];
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int k = cg + WAVES * c;
        if (i > k) S[i + (size_t)k * ld] = As[i * LDD3 + k];
    }
    if (cg == 1 && i < 16) inv16(As, Xs, 48, i);
    TS(14);
    __syncthreads();
    {
        const int p = tid >> 8, ii = (tid >> 4) & 15, jj2 = tid & 15, o = 32 * p;
        double t = 0.0;
        if (tid < 512) {
#pragma unroll
            for (int k = 0; k < 16; ++k) t += As[(o + 16 + ii) * LDD3 + o + k] * Xs[(o + k) * LDD3 + o + jj2];
            Ts[(p * 16 + ii) * 33 + jj2] = t;
        }
        __syncthreads();
        if (tid < 512) {
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < 16; ++k) v -= Xs[(o + 16 + ii) * LDD3 + o + 16 + k] * Ts[(p * 16 + k) * 33 + jj2];
            Xs[(o + 16 + ii) * LDD3 + o + jj2] = v;
        }
        __syncthreads();
    }
    TS(15);
    {
        const int ii = tid >> 5, jj2 = tid & 31;
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 32; ++k) t += As[(32 + ii) * LDD3 + k] * Xs[k * LDD3 + jj2];
        Ts[ii * 33 + jj2] = t;
        __syncthreads();
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < 32; ++k) v -= Xs[(32 + ii) * LDD3 + 32 + k] * Ts[k * 33 + jj2];
        __syncthreads();
        Xs[(32 + ii) * LDD3 + jj2] = v;
    }
    __syncthreads();
    TS(16);
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int k = cg + WAVES * c;
        Xout[i + k * NB] = Xs[i * LDD3 + k];
    }
    TS(17);
}


// v8: 4-column mini-panels.  Wavefront cg owns the four CONSECUTIVE columns 4 cg .. 4 cg + 3 (lane = row).  Per mini-panel the owner wave
// factors its four columns alone (pivot entries by v_readlane inside the wave, no barrier), publishes the unscaled columns + reciprocals to a
// double-buffered LDS panel, ONE workgroup barrier, and every later wave applies the rank-4 update to its own four columns.  16 barriers per
// 64 columns instead of 64; the critical wave executes ~50 + ~40 instructions per four columns instead of 4 x 30.
__device__ __forceinline__ double fast_rcp1(double v) {   // v_rcp_f64 + 1 Newton step
    double r = __builtin_amdgcn_rcp(v);
    r = fma(fma(-v, r, 1.0), r, r);
    r = fma(fma(-v, r, 1.0), r, r);
    return r;
}
template <int P> struct MiniPanel {
    // a[0..3]: this lane's row of the owner's four columns (global columns 4P..4P+3); returns y (unscaled), l (scaled below the pivot), rinv
    static __device__ __forceinline__ void run(double (&a)[4], int i, double (&y)[4], double (&rinv)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = 4 * P + j;
            const double d = readlane_d(a[j], r);
            rinv[j] = fast_rcp1(d);
            y[j] = a[j];
            const double li = a[j] * rinv[j];
#pragma unroll
            for (int k = j + 1; k < 4; ++k) a[k] -= li * readlane_d(a[j], 4 * P + k);     // A[4P+k][r]: the symmetric partner of row r's entry
        }
    }
};
__global__ __launch_bounds__(1024) void k_diag_v8(int ld, double* __restrict__ S, double* __restrict__ Dx, double* __restrict__ Xout) {
    constexpr int WAVES = 16;
    __shared__ double Ls[NB * LDD3];
    __shared__ double Xs[NB * LDD3];
    __shared__ double Ts[32 * 33];
    __shared__ double ypan[2][4][NB];       // unscaled pivot columns of the current mini-panel
    __shared__ double rpan[2][4];           // their reciprocals
    const int tid = threadIdx.x, i = tid & 63, cg = tid >> 6;
    double a[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int k = 4 * cg + c;
        a[c] = (i >= k) ? S[i + (size_t)k * ld] : 0.0;
    }
    TS(0);
    auto panel = [&](auto Ptag) {
        constexpr int P = decltype(Ptag)::value;
        const int buf = P & 1;
        if (cg == P) {
            double y[4], rinv[4];
            MiniPanel<P>::run(a, i, y, rinv);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ypan[buf][j][i] = y[j];
                Ls[i * LDD3 + 4 * P + j] = (i > 4 * P + j) ? y[j] * rinv[j] : ((i == 4 * P + j) ? y[j] : 0.0);    // L below, the pivot on the diagonal
            }
            if (i < 4) rpan[buf][i] = rinv[i];
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        if (cg > P) {
            double l[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) l[j] = (i > 4 * P + j) ? ypan[buf][j][i] * rpan[buf][j] : 0.0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int k = 4 * cg + c;
#pragma unroll
                for (int j = 0; j < 4; ++j) a[c] -= l[j] * ypan[buf][j][k];
            }
        }
    };
    panel(std::integral_constant<int, 0>{}); panel(std::integral_constant<int, 1>{}); panel(std::integral_constant<int, 2>{}); panel(std::integral_constant<int, 3>{});
    panel(std::integral_constant<int, 4>{}); panel(std::integral_constant<int, 5>{}); panel(std::integral_constant<int, 6>{}); panel(std::integral_constant<int, 7>{});
    panel(std::integral_constant<int, 8>{}); panel(std::integral_constant<int, 9>{}); panel(std::integral_constant<int, 10>{}); panel(std::integral_constant<int, 11>{});
    panel(std::integral_constant<int, 12>{}); panel(std::integral_constant<int, 13>{}); panel(std::integral_constant<int, 14>{}); panel(std::integral_constant<int, 15>{});
    __syncthreads();
    TS(1);
    if (tid < NB) Dx[tid] = Ls[tid * LDD3 + tid];
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int k = 4 * cg + c;
        if (i > k) S[i + (size_t)k * ld] = Ls[i * LDD3 + k];
        if (i <= k) Ls[i * LDD3 + k] = 0.0;           // the inverse below reads L only: clear the diagonal / upper part
        Xs[i * LDD3 + k] = 0.0;
    }
    __syncthreads();
    TS(2);
    if (cg < 4 && i < 16) inv16(Ls, Xs, 16 * cg, i);
    __syncthreads();
    {
        const int p = tid >> 8, ii = (tid >> 4) & 15, jj2 = tid & 15, o = 32 * p;
        double t = 0.0;
        if (tid < 512) {
#pragma unroll
            for (int k = 0; k < 16; ++k) t += Ls[(o + 16 + ii) * LDD3 + o + k] * Xs[(o + k) * LDD3 + o + jj2];
            Ts[(p * 16 + ii) * 33 + jj2] = t;
        }
        __syncthreads();
        if (tid < 512) {
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < 16; ++k) v -= Xs[(o + 16 + ii) * LDD3 + o + 16 + k] * Ts[(p * 16 + k) * 33 + jj2];
            Xs[(o + 16 + ii) * LDD3 + o + jj2] = v;
        }
        __syncthreads();
    }
    {
        const int ii = tid >> 5, jj2 = tid & 31;
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 32; ++k) t += Ls[(32 + ii) * LDD3 + k] * Xs[k * LDD3 + jj2];
        Ts[ii * 33 + jj2] = t;
        __syncthreads();
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < 32; ++k) v -= Xs[(32 + ii) * LDD3 + 32 + k] * Ts[k * 33 + jj2];
        __syncthreads();
        Xs[(32 + ii) * LDD3 + jj2] = v;
    }
    __syncthreads();
    TS(3);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int k = 4 * cg + c;
        Xout[i + k * NB] = Xs[i * LDD3 + k];
    }
    TS(4);
}

// v5 = v4 templated on the number of wavefronts (v3 + unmasked column updates, finished columns stashed in LDS, per-lane reciprocal vector): LDL^T loop without the inverse; X = L^-1 afterwards by 16 x 16 wave-synchronous inversions + two merge levels in LDS

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void k_diag_v5(int ld, double* __restrict__ S, double* __restrict__ Dx, double* __restrict__ Xout) {
    constexpr int CPW = NB / WAVES;
    __shared__ double colbuf[2][NB];
    __shared__ double rinvvec[2][NB];
    __shared__ double Ls[NB * LDD3];
    __shared__ double Xs[NB * LDD3];
    __shared__ double Ts[32 * 33];
    const int tid = threadIdx.x, i = tid & 63, cg = tid >> 6;
    double a[CPW];
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int k = cg + WAVES * c;
        a[c] = (i >= k) ? S[i + (size_t)k * ld] : 0.0;
    }
    if (cg == 0) {
        colbuf[0][i] = a[0];
        rinvvec[0][i] = fast_rcp(a[0]);
        Ls[i * LDD3 + 0] = a[0];
    }
#pragma unroll 1
    for (int jb = 0; jb < NB; jb += WAVES)
#pragma unroll
    for (int jj = 0; jj < WAVES; ++jj) {
        const int j = jb + jj;
        const int cur = jj & 1, nxt = cur ^ 1;
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_s_barrier();
        const double yi = colbuf[cur][i];
        const double rinv = rinvvec[cur][j];
        double yk[CPW];
#pragma unroll
        for (int c = 0; c < CPW; ++c) yk[c] = colbuf[cur][cg + WAVES * c];
        const double li = (i > j) ? yi * rinv : 0.0;      // rows at/above the pivot: no update (upper part is never read)
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
            const int k = cg + WAVES * c;
            a[c] -= li * yk[c];          // finished columns (k <= j) were stashed in Ls when published; garbage here is never read
            (void)k;
        }
        if (j + 1 < NB && cg == (jj + 1) % WAVES) {
            const int cs = (j + 1) / WAVES;
            double v = a[0];
#pragma unroll
            for (int c = 1; c < CPW; ++c) v = (cs == c) ? a[c] : v;
            colbuf[nxt][i] = v;
            rinvvec[nxt][i] = fast_rcp(v);           // every lane: readers pick entry j+1 (no divergent single-lane path)
            Ls[i * LDD3 + j + 1] = v;                // unscaled column j+1 (rows >= j+1 meaningful), pivot on the diagonal
        }
    }
    __syncthreads();
    // scale the stashed columns: L into LDS (for the inverse) and to global; pivots out
    double dk[CPW];
#pragma unroll
    for (int c = 0; c < CPW; ++c) dk[c] = Ls[(cg + WAVES * c) * LDD3 + cg + WAVES * c];
    if (tid < NB) Dx[tid] = Ls[tid * LDD3 + tid];
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int k = cg + WAVES * c;
        const double l = (i > k) ? Ls[i * LDD3 + k] * (1.0 / dk[c]) : 0.0;
        Ls[i * LDD3 + k] = l;
        Xs[i * LDD3 + k] = 0.0;
        if (i > k) S[i + (size_t)k * ld] = l;
    }
    __syncthreads();
    // (a) the four 16 x 16 diagonal blocks of X: wave w, lane c < 16 builds column c by forward substitution (registers, no barrier)
    for (int blk = cg; blk < 4; blk += WAVES) if (i < 16) {
        const int o = 16 * blk;
        double x[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            double acc = (r == i) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < r; ++k) acc -= Ls[(o + r) * LDD3 + o + k] * x[k];
            x[r] = (r >= i) ? acc : 0.0;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) Xs[(o + r) * LDD3 + o + i] = x[r];
    }
    __syncthreads();
    // (b) level 1: X21 = -X22 (L21 X11) for the two 32-blocks; 512 threads, one output each
    {
        for (int e = tid; e < 512; e += WAVES * 64) {
            const int p = e >> 8, ii = (e >> 4) & 15, jj2 = e & 15, o = 32 * p;
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < 16; ++k) t += Ls[(o + 16 + ii) * LDD3 + o + k] * Xs[(o + k) * LDD3 + o + jj2];
            Ts[(p * 16 + ii) * 33 + jj2] = t;
        }
        __syncthreads();
        for (int e = tid; e < 512; e += WAVES * 64) {
            const int p = e >> 8, ii = (e >> 4) & 15, jj2 = e & 15, o = 32 * p;
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < 16; ++k) v -= Xs[(o + 16 + ii) * LDD3 + o + 16 + k] * Ts[(p * 16 + k) * 33 + jj2];
            Xs[(o + 16 + ii) * LDD3 + o + jj2] = v;
        }
        __syncthreads();
    }
    // (c) level 2: X21 (32 x 32) = -X22 (L21 X11); 1024 threads, one output each
    {
        for (int e = tid; e < 1024; e += WAVES * 64) {
            const int ii = e >> 5, jj2 = e & 31;
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < 32; ++k) t += Ls[(32 + ii) * LDD3 + k] * Xs[k * LDD3 + jj2];
            Ts[ii * 33 + jj2] = t;
        }
        __syncthreads();
        double vv[1024 / (WAVES * 64)];
        for (int e = tid, n = 0; e < 1024; e += WAVES * 64, ++n) {
            const int ii = e >> 5, jj2 = e & 31;
            double v = 0.0;
#pragma unroll
            for (int k = 0; k < 32; ++k) v -= Xs[(32 + ii) * LDD3 + 32 + k] * Ts[k * 33 + jj2];
            vv[n] = v;
        }
        __syncthreads();
        for (int e = tid, n = 0; e < 1024; e += WAVES * 64, ++n) Xs[(32 + (e >> 5)) * LDD3 + (e & 31)] = vv[n];
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int k = cg + WAVES * c;
        Xout[i + k * NB] = Xs[i * LDD3 + k];
    }
}

template <typename K>
void run(const char* name, K kern, int threads, const std::vector<double>& A0, bool with_x) {
    const int ld = NB, reps = 200;
    double *S, *D, *X;
    hipMalloc(&S, sizeof(double) * NB * NB); hipMalloc(&D, sizeof(double) * NB); hipMalloc(&X, sizeof(double) * NB * NB);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipMemcpy(S, A0.data(), sizeof(double) * NB * NB, hipMemcpyHostToDevice);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, ld, S, D, X);   // (repeats refactor garbage: timing only)
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    hipMemcpy(S, A0.data(), sizeof(double) * NB * NB, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(kern, dim3(1), dim3(threads), 0, 0, ld, S, D, X);
    std::vector<double> L(NB * NB), d(NB), Xh(NB * NB);
    hipMemcpy(L.data(), S, sizeof(double) * NB * NB, hipMemcpyDeviceToHost);
    hipMemcpy(d.data(), D, sizeof(double) * NB, hipMemcpyDeviceToHost);
    hipMemcpy(Xh.data(), X, sizeof(double) * NB * NB, hipMemcpyDeviceToHost);
    double err = 0, errx = 0;
    for (int i = 0; i < NB; ++i) for (int j = 0; j <= i; ++j) {
        double s = 0;
        for (int k = 0; k <= j; ++k) s += (i == k ? 1.0 : L[i + k * NB]) * d[k] * (j == k ? 1.0 : L[j + k * NB]);
        err = fmax(err, fabs(s - A0[i + j * NB]));
        if (with_x) { double t = 0; for (int k = j; k <= i; ++k) t += Xh[i + k * NB] * (k == j ? 1.0 : L[k + j * NB]); errx = fmax(errx, fabs(t - (i == j ? 1.0 : 0.0))); }
    }
    printf("%-44s %7.2f us/launch (incl. launch gap)   |LDL'-A| %.1e  |XL-I| %.1e\n", name, best * 1e3 / reps, err, errx);
    hipFree(S); hipFree(D); hipFree(X);
}

int main() {
    std::vector<double> A(NB * NB);
    unsigned s = 12345;
    for (int i = 0; i < NB; ++i) for (int j = 0; j <= i; ++j) {
        s = s * 1664525u + 1013904223u;
        double v = ((s >> 8) & 0xffff) / 65536.0 - 0.5;
        A[i + j * NB] = A[j + i * NB] = (i == j) ? 8.0 + v : v * 0.2;
    }
    run("16 waves, X, div, syncthreads", k_diag<16, true, false, false>, 1024, A, true);
    run("16 waves, X, rcp, syncthreads", k_diag<16, true, true, false>, 1024, A, true);
    run("16 waves, X, rcp, raw barrier", k_diag<16, true, true, true>, 1024, A, true);
    run("16 waves, noX, rcp, raw barrier", k_diag<16, false, true, true>, 1024, A, false);
    run(" 8 waves, X, rcp, raw barrier", k_diag<8, true, true, true>, 512, A, true);
    run(" 4 waves, X, rcp, raw barrier", k_diag<4, true, true, true>, 256, A, true);
    run(" 4 waves, noX, rcp, raw barrier", k_diag<4, false, true, true>, 256, A, false);
    run("skeleton 16 waves (barrier+LDS)", k_skeleton<16,0>, 1024, A, false);
    run("skeleton 16 waves + rcp", k_skeleton<16,1>, 1024, A, false);
    run("skeleton  4 waves (barrier+LDS)", k_skeleton<4,0>, 256, A, false);
    run("skeleton  4 waves + rcp", k_skeleton<4,1>, 256, A, false);
    run("skeleton  1 wave + rcp", k_skeleton<1,1>, 64, A, false);
    run("v3: LDL loop + blocked inverse", k_diag_v3, 1024, A, true);
    run("v4: v3 + unmasked/stash/rcp-vector", k_diag_v4, 1024, A, true);
    run("v6: 16-col sub-panels in one wave (readlane)", k_diag_v6, 1024, A, true);
    { long long h[32]; hipMemcpyFromSymbol(h, HIP_SYMBOL(g_ts), sizeof(h)); printf("v6 timeline (10 ns ticks from start):"); for (int q = 0; q < 18; ++q) printf(" [%d]%lld", q, h[q] - h[0]); printf("\n"); }
    run("v7: 16-col stages, DPP pivot broadcast, MFMA update", k_diag_v7, 1024, A, true);
    { long long h[32]; hipMemcpyFromSymbol(h, HIP_SYMBOL(g_ts), sizeof(h)); printf("v7 timeline (10 ns ticks from start):"); for (int q = 0; q < 18; ++q) printf(" [%d]%lld", q, h[q] - h[0]); printf("\n"); }
    run("v8: 4-column mini-panels (readlane inside the owner wave)", k_diag_v8, 1024, A, true);
    { long long h[32]; hipMemcpyFromSymbol(h, HIP_SYMBOL(g_ts), sizeof(h)); printf("v8 timeline (10 ns ticks from start):"); for (int q = 0; q < 5; ++q) printf(" [%d]%lld", q, h[q] - h[0]); printf("\n"); }
    run("v5 16 waves", k_diag_v5<16>, 1024, A, true);
    run("v5  8 waves", k_diag_v5<8>, 512, A, true);
    run("v5  4 waves", k_diag_v5<4>, 256, A, true);
    run("merged 16 waves", k_diag_merged<16>, 1024, A, true);
    run("merged  8 waves", k_diag_merged<8>, 512, A, true);
    run("merged  4 waves", k_diag_merged<4>, 256, A, true);
    return 0;
}
