// ldl.hip — blocked right-looking LDL^T (no pivoting) of the nx x nx Schur complement S produced by schur.hip, and the
// triangular solves with its factors.  Together with the closed-form constraint pivots of schur.hip this is the
// factorisation  P K P' = L D L'  of factorize!/QDLDL_factor! (linear_solver.jl:19-31, qdldl.jl:400-589) in the order
// [z | y | x]; every pivot of S must be > 0 for the inertia test (inertia.jl:7-11).
//
// Round 3: ONE launch per panel of NB = 64 columns (round 2: two — a panel GEMM and the trailing update).  The panel below a
// factored diagonal block is never scaled on the critical path: with M_k = (L_kk D_k L_kk')^-1 = X_k' D_k^-1 X_k, X_k = L_kk^-1,
// the trailing update is  A(i,j) -= [A(i,k) M_k] A(j,k)'  straight from the RAW panel columns that sit in S, so the launch of
// panel k needs nothing but M_k (a 64 x 64 block its predecessor wrote) and the pivot chain is
//     launch k:  tile (k+1,k+1) -= A(k+1,k) M_k A(k+1,k)'  ->  LDL^T of that 64 x 64 block  ->  X_{k+1}, M_{k+1}
// with the bulk of the update overlapped in the same launch.  The factor L(i,k) = A(i,k) X_k' D_k^-1 is formed for ALL panels
// afterwards in one fully parallel launch (k_ldl_scale); kernel boundaries on one stream are the only synchronisation.
//   k_ldl_diag      one workgroup: the first diagonal block (diag_block below: 64 x 64 LDL^T in registers, four-column mini-panels,
//                   then X = L11^-1 by blocked inversion, M = X' D^-1 X on the matrix cores, pivot signs for compute_inertia!
//                   (linear_solver.jl:33-44), exact zeros flagged as qdldl.jl:579 does).
//   k_ldl_step      A22 -= (A21 M) A21'  on the matrix cores: 64 x 64 tiles of the lower triangle, 1024 threads (16 wavefronts, one
//                   16 x 16 MFMA tile each), persistent workgroups walking CONTIGUOUS runs of tiles (Z = A(i,k) M is formed once per
//                   tile row of a run); tile 0 goes on to factor the next diagonal block.  <1> / <2>: the pair schedule of groups
//                   (two panels applied in one pass).
//   k_ldl_scale     L21 = A21 X' D^-1 for every panel at once (what round 2's k_ldl_panel did per panel, on the critical path).
// A handle whose S is stage-structured can bypass all of this: calipso_hip_set_stage_parallel routes launch_ldl / launch_trsv to the multifrontal
// sparse LDL^T of sparse.hip over a nested dissection of S.
// Triangular solves work on blocks of up to 1024 columns: the inverses of the unit-lower diagonal blocks of L are assembled
// from the 64 x 64 inverses by four levels of small matrix-core GEMMs (k_tinv_*), so a solve is 2 launches per block
// (10 launches for NP = 2560) instead of a 2560-long dependent chain.
#include "internal.hpp"
#include "device_utils.hpp"

#include <algorithm>
#include <mutex>

namespace calipso {

typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int TB = 1024;           // largest triangular-solve block; the block actually used is tb = min(TB, NP); the last block of a solve may be narrower
constexpr int LDT = NB + 2;        // LDS leading dimension of a k-fastest 64-deep operand panel

// ---- diagonal block ---------------------------------------------------------------------------------------------------------
// The 2500 sequential pivots of S are the critical path of the factorisation; this kernel is tuned with the stand-alone
// harness bench/diag_bench.hip (variant v8; profiles/r02_diag_bench.txt has the timelines of the alternatives).  1024 threads: lane
// i = tid & 63 is a ROW of the block, wavefront cg = tid >> 6 owns the four CONSECUTIVE columns 4 cg .. 4 cg + 3 in registers.
//   phase 1  LDL^T in 16 mini-panels of four columns: the owner wavefront factors its four columns alone — the pivot entries it needs
//            from other rows are in its own lanes and travel by v_readlane, no barrier — publishes the four unscaled columns and their
//            reciprocal pivots to a double-buffered LDS panel, ONE workgroup barrier, and every later wavefront applies the rank-4 update
//            to its own columns.  16 barriers per block instead of 64 (one per column in round 1); a single wavefront issues one dependent
//            instruction every ~13 cycles, so what counts on the critical path is the instruction count of the wave that holds the
//            next pivot: ~50 (mini-panel) + ~40 (update) per four columns instead of 4 x 30 + 4 barriers.  10.6 us instead of 15.
//            Round 3: the wavefronts that are done with the LDL^T (cg <= P) apply G_P^-1 to their columns of X = L11^-1 in the same step
//            (v_readlane inside the wave), so the inverse is complete when the last pivot is (round 2: a separate phase 2, 7 us per block).
//   phase 2  M = X' D^-1 X on the matrix cores (what the next panel step multiplies the raw panel with), then ALL global stores: D, L, X, M.
// Nothing is written to global memory before the last barrier (a pending store would make a barrier wait on memory).
constexpr int DIAG_THREADS = 1024;
constexpr int LDD = NB + 1;        // stride of the 64 x 64 tile handed to the diagonal block through LDS

// Optional timeline of the pivot chain (build with -DCALIPSO_LDL_TRACE; bench/ldl_trace.py reads it through calipso_hip_debug_ldl_trace):
// 100 MHz wall-clock stamps of the workgroup that carries tile 0 + the diagonal block, instance 0, per panel step.
#ifdef CALIPSO_LDL_TRACE
__device__ long long g_ldl_trace[64 * 16];
#define LDL_STAMP(step, slot) do { if (threadIdx.x == 0 && (step) < 64) g_ldl_trace[(step) * 16 + (slot)] = wall_clock64(); } while (0)
#define LDL_STAMP_IF(cond, step, slot) do { if ((cond) && (step) < 64) g_ldl_trace[(step) * 16 + (slot)] = wall_clock64(); } while (0)
#else
#define LDL_STAMP(step, slot) do { } while (0)
#define LDL_STAMP_IF(cond, step, slot) do { } while (0)
#endif

__device__ __forceinline__ double fast_rcp(double v) {   // v_rcp_f64 + 2 Newton steps (pivots are normal numbers; 0 -> inf as 1/0)
    double r = __builtin_amdgcn_rcp(v);
    r = fma(fma(-v, r, 1.0), r, r);
    r = fma(fma(-v, r, 1.0), r, r);
    return r;
}

// LDS carve (doubles): XT | XTs | ypan[2][4][NB] | rpan[2][4] | dpiv[NB] | dinv[NB].  When the block arrives through LDS (fused with the trailing update)
// it sits at the start (row-major, stride LDD) and is consumed into registers before anything is written.
constexpr int DIAG_LDS_DOUBLES = 2 * NB * LDT + 2 * 4 * NB + 8 + 2 * NB;

__device__ __forceinline__ void lds_barrier_all() {                  // workgroup barrier that orders LDS traffic only (global stores stay in flight)
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
}
__device__ __forceinline__ double readlane_d(double v, int lane) {     // lane: wave-uniform
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}
typedef double v2d __attribute__((ext_vector_type(2)));
template <bool FROM_LDS>
__device__ __forceinline__ void diag_block(double* __restrict__ smem, int NP, int nx, int k0, int tb, double* __restrict__ S, double* __restrict__ Dx,
                                           double* __restrict__ Tinv, double* __restrict__ Minv, int* __restrict__ icount) {
    constexpr int WAVES = 16, CPW = 4;
    double* XT = smem;                       // XT[a][r] = X[r][a], stride LDT: the matrix-core fragments of M read it without bank conflicts
    double* XTs = XT + NB * LDT;             // XT scaled by the reciprocal pivot of row r
    double (*ypan)[4][NB] = reinterpret_cast<double (*)[4][NB]>(XTs + NB * LDT);   // [2][4][NB] unscaled pivot columns of a mini-panel
    double (*rpan)[4] = reinterpret_cast<double (*)[4]>(XTs + NB * LDT + 8 * NB);  // [2][4] their reciprocal pivots
    double* dpiv = XTs + NB * LDT + 8 * NB + 8;                                     // the 64 pivots
    double* dinv = dpiv + NB;                                                       // and their reciprocals
    const int tid = threadIdx.x, i = tid & 63, cg = tid >> 6;
    double a[CPW];
#pragma unroll
    for (int c = 0; c < CPW; ++c) {
        const int k = 4 * cg + c;
        if (FROM_LDS) a[c] = (i >= k) ? smem[i * LDD + k] : 0.0;
        else a[c] = (i >= k) ? S[(k0 + i) + (size_t)(k0 + k) * NP] : 0.0;
    }
    if (FROM_LDS) __syncthreads();   // every lane has its entries before the region is reused
    LDL_STAMP(k0 / NB, 2);
    // X = L11^-1 grows alongside: L = G_0 G_1 ... G_15 (G_P = identity + the four columns of mini-panel P), so X = G_15^-1 ... G_0^-1 and
    // G_P^-1 is applied from the left as soon as mini-panel P is published:  X[i][:] -= L[i][4P+j] X[4P+j][:]  for j = 0..3 in turn, i > 4P+j.
    // Wavefront cg holds columns 4 cg .. 4 cg + 3 of X (lane = row, like A); row 4P+j of its columns sits in its own lane 4P+j (v_readlane).
    // Only columns <= 4P+3 are touched by G_P^-1, i.e. wavefronts cg <= P — exactly the ones with nothing left to do in the LDL^T — so the
    // inverse costs the pivot chain nothing (round 2 formed it afterwards: 16 x 16 forward substitutions + two merge levels, 7 us per block).
    double x[CPW], lfin[CPW];                 // lfin: the finished columns of L of this wavefront (they go to global memory at the very end)
#pragma unroll
    for (int c = 0; c < CPW; ++c) { x[c] = (i == 4 * cg + c) ? 1.0 : 0.0; lfin[c] = 0.0; }
#pragma unroll 1
    for (int P = 0; P < WAVES; ++P) {
        const int buf = P & 1;
        if (cg == P) {
            // the owner's four columns, alone: rows <= the pivot only collect garbage that is never read
            double y[4], rinv[4], dv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dv[j] = readlane_d(a[j], 4 * P + j);
                rinv[j] = fast_rcp(dv[j]);
                y[j] = a[j];
                const double li = a[j] * rinv[j];
                lfin[j] = li;
#pragma unroll
                for (int k = j + 1; k < 4; ++k) a[k] -= li * readlane_d(a[j], 4 * P + k);    // A[4P+k][4P+j]: the symmetric partner of the pivot row's entry
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) ypan[buf][j][i] = y[j];
            if (i < 4) {
                const double dd = i == 0 ? dv[0] : i == 1 ? dv[1] : i == 2 ? dv[2] : dv[3], rr = i == 0 ? rinv[0] : i == 1 ? rinv[1] : i == 2 ? rinv[2] : rinv[3];
                dpiv[4 * P + i] = dd; dinv[4 * P + i] = rr; rpan[buf][i] = rr;
            }
            LDL_STAMP_IF(i == 0 && cg == 9, k0 / NB, 10);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): only LDS traffic is outstanding here
        __builtin_amdgcn_s_barrier();
        LDL_STAMP_IF(i == 0 && cg == 9 && P == 8, k0 / NB, 8);
        LDL_STAMP_IF(i == 0 && cg == 9 && P == 9, k0 / NB, 11);
        // (every load is issued before any is used: a load inside the `i > pivot` conditional turns into exec-masked blocks with an LDS round
        // trip each).  Measured alternatives (bench/diag_bench2.hip, profiles/r03_diag_bench2.txt): 16-byte (y, l) pairs, pivot-row entries by
        // v_readlane instead of broadcast reads, s_sleep for the non-critical wavefronts, s_setprio — all slower.
        double yl[4], rp[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { yl[j] = ypan[buf][j][i]; rp[j] = rpan[buf][j]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) { const double v = yl[j] * rp[j]; l[j] = (i > 4 * P + j) ? v : 0.0; }
        if (cg > P) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                double yr[CPW];
#pragma unroll
                for (int c = 0; c < CPW; ++c) yr[c] = ypan[buf][j][4 * cg + c];                   // Y[4cg+c][4P+j] (broadcast reads)
#pragma unroll
                for (int c = 0; c < CPW; ++c) a[c] -= l[j] * yr[c];
            }
            LDL_STAMP_IF(i == 0 && cg == 9 && P == 8 && a[0] != 1.2345e300, k0 / NB, 9);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {                 // (j outermost: the four columns are independent chains)
                double xs[CPW];
#pragma unroll
                for (int c = 0; c < CPW; ++c) xs[c] = readlane_d(x[c], 4 * P + j);
#pragma unroll
                for (int c = 0; c < CPW; ++c) x[c] -= l[j] * xs[c];                               // (l[j] = 0 for the rows i <= 4P+j)
            }
        }
    }
    LDL_STAMP(k0 / NB, 3);
    // X' to LDS for the matrix cores, plain and scaled by the reciprocal pivot of its row (the last mini-panel's pivots are visible: barrier 15)
    {
        const double di = dinv[i];
#pragma unroll
        for (int c = 0; c < CPW; ++c) { XT[(4 * cg + c) * LDT + i] = x[c]; XTs[(4 * cg + c) * LDT + i] = x[c] * di; }
    }
    lds_barrier_all();
    LDL_STAMP(k0 / NB, 4);
    // M = X' D^-1 X = (L11 D L11')^-1: what the NEXT launch multiplies the raw panel with.  M[a][b] = sum_r X[r][a] X[r][b] / d[r] on the matrix
    // cores: wavefront (wa, wb) forms the 16 x 16 tile (rows a, columns b); both operand fragments are "row a (b), k index r" reads of X'.
    {
        const int wa = cg >> 2, wb = cg & 3, fr = i & 15, fk = i >> 4;
        v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
        // X[r][a] = 0 for r < a: the k blocks above the later of the two tile origins contribute exact zeros and are skipped (the workgroup's 256
        // MFMAs shrink to 120; the matrix cores of one CU are what bounds this product)
        for (int kk = 4 * (wa > wb ? wa : wb); kk < NB / 4; ++kk) {
            const double xa = XTs[(wa * 16 + fr) * LDT + 4 * kk + fk];
            const double xb = XT[(wb * 16 + fr) * LDT + 4 * kk + fk];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, xb, acc, 0, 0, 0);
        }
        double* Mo = Minv + (size_t)(k0 / NB) * NB * NB;
        LDL_STAMP(k0 / NB, 6);
#pragma unroll
        for (int q = 0; q < 4; ++q) Mo[(wb * 16 + fr) + (size_t)(wa * 16 + fk + 4 * q) * NB] = acc[q];   // lane holds M(a = fk + 4 q, b = fr) = M(b, a): 128-byte runs along fr
    }
    // everything that goes to global memory leaves here, after the last barrier: D and the inertia counts (compute_inertia!), the strictly
    // lower L of the block (from the registers of the wavefront that factored the column), X = L11^-1 on the diagonal of the triangular-solve
    // inverse block (zeros above)
    if (tid < NB) {
        const double d = dpiv[tid];
        Dx[k0 + tid] = d;
        const bool real = k0 + tid < nx;                                  // (padding rows carry unit pivots that are not counted)
        const int pos = __popcll(__ballot(real && d > 0.0)), nonpos = __popcll(__ballot(real && d <= 0.0)), zero = __popcll(__ballot(real && d == 0.0));
        if (tid == 0) { atomicAdd(&icount[3], pos); atomicAdd(&icount[4], nonpos); atomicAdd(&icount[5], zero); }
    }
    {
        const int q = k0 / tb, o = k0 % tb;
        double* T = Tinv + (size_t)q * tb * tb;
#pragma unroll
        for (int c = 0; c < CPW; ++c) {
            const int k = 4 * cg + c;
            T[(o + i) + (size_t)(o + k) * tb] = x[c];
            if (i > k) S[(k0 + i) + (size_t)(k0 + k) * NP] = lfin[c];
        }
    }
    LDL_STAMP(k0 / NB, 5);
}

__global__ __launch_bounds__(DIAG_THREADS) void k_ldl_diag(Batch bt, int NP, int nx, int k0, int tb, double* __restrict__ S, double* __restrict__ Dx,
                                                            double* __restrict__ Tinv, double* __restrict__ Minv, int* __restrict__ icount) {
    __shared__ double smem[DIAG_LDS_DOUBLES];
    inst_shift(bt, S, Dx, Tinv, Minv);
    inst_shift_i(bt, icount);
    diag_block<false>(smem, NP, nx, k0, tb, S, Dx, Tinv, Minv, icount);
}

// ---- factor columns: L21 = A21 X' D^-1 for every panel in one launch -------------------------------------------------------------
// After the last panel step the sub-diagonal tiles of S still hold the RAW panel columns A(i,k) (final Schur-complement values); the
// factor is L(i,k) = A(i,k) X_k' D_k^-1.  D[c][r] = sum_k X[c][k] A21[r][k]: MFMA A operand = X (rows c), B operand = A21' so that the
// 16-lane fast index of the result is the contiguous row index r of the column-major panel.  One workgroup (16 wavefronts) per
// 64 x 64 tile (blockIdx.x = tile row below the panel, blockIdx.y = panel); wavefront (wr, wc) computes the 16 x 16 tile rows 16 wr..,
// columns 16 wc..; X is staged in LDS.  Off the critical path: 780 independent tiles at C3.
__global__ __launch_bounds__(1024) void k_ldl_scale(Batch bt, int NP, int tb, int band_rows, double* __restrict__ S, const double* __restrict__ Dx,
                                                     const double* __restrict__ Tinv) {
    __shared__ double Xs[NB * LDT];   // Xs[c][k]
    __shared__ double dinv[NB];
    const int k0 = (int)blockIdx.y * NB;
    const int rows = min(NP - k0 - NB, band_rows);           // banded S: the panel stops at the band
    if ((int)blockIdx.x * 64 >= rows) return;
    inst_shift(bt, S, Dx, Tinv);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, fk = lane >> 4;
    const int r0 = k0 + NB + blockIdx.x * 64 + wr * 16;
    double b[NB / 4];
#pragma unroll
    for (int kk = 0; kk < NB / 4; ++kk) b[kk] = S[(r0 + fr) + (size_t)(k0 + kk * 4 + fk) * NP];
    {
        const int q = k0 / tb, o = k0 % tb;
        const double* T = Tinv + (size_t)q * tb * tb;
        const int c = tid & 63;
#pragma unroll
        for (int kk = tid >> 6; kk < NB; kk += 16) Xs[c * LDT + kk] = T[(o + c) + (size_t)(o + kk) * tb];
        if (tid < NB) dinv[tid] = 1.0 / Dx[k0 + tid];
    }
    __syncthreads();                  // (also: every lane holds its raw entries before any of them is overwritten below)
    v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < NB / 4; ++kk) {
        const double xa = Xs[(wc * 16 + fr) * LDT + kk * 4 + fk];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, b[kk], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int c = wc * 16 + fk + 4 * r;   // MFMA row  -> panel column
        const int row = r0 + fr;              // MFMA col  -> panel row (contiguous)
        S[row + (size_t)(k0 + c) * NP] = acc[r] * dinv[c];
    }
}

// ---- panel step: A22 -= (A21 M) A21' ---------------------------------------------------------------------------------------------
// 64 x 64 tile of the lower triangle per workgroup of 1024 threads (16 wavefronts, one 16 x 16 MFMA tile each).  ONE workgroup is resident
// per CU (registers).  Small tiles keep all 256 CUs busy on the shrinking trailing matrix.  The tile is computed transposed (MFMA row <->
// column j of S) so result stores are 128-byte runs.
constexpr int TR_THREADS = 1024;
constexpr int TT = 64;
constexpr int step_lds_doubles(int nh) { return (nh + 2) * TT * LDT > DIAG_LDS_DOUBLES ? (nh + 2) * TT * LDT : DIAG_LDS_DOUBLES; }   // Zs[nh] | Ys | Ms
// Tile 0 of the trailing update IS the next diagonal block: its workgroup keeps going and factors that block (diag_block),
// so the 64-column pivot chain of panel k+1 runs inside this launch, overlapped with the other tiles, and a panel step is ONE launch.
// Workgroups are persistent: workgroup 0 takes tile 0 (and then the diagonal block), workgroup w >= 1 walks a CONTIGUOUS run of the
// row-major tile list (so that consecutive tiles mostly share their tile row i, whose Z = A(i,k) M is formed once and kept in LDS) and
// fetches the operands of its next tile into registers while the matrix cores work on the current one.
__device__ __forceinline__ void trailing_tile_index(int t, int& ti, int& tj) {
    ti = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((ti + 1) * (ti + 2) / 2 <= t) ++ti;
    while (ti * (ti + 1) / 2 > t) --ti;
    tj = t - ti * (ti + 1) / 2;
}
// MODE 0: one panel (columns k0 .. k0 + 63), every tile of the trailing matrix.
// MODE 1: one panel, only the FIRST tile column (tiles (i, 0)): what the next panel needs.
// MODE 2: the two panels k0 and k0 + 64 applied in ONE pass over the tiles from block k0 + 128 on: every entry of the trailing matrix is
//         read and written once per 128 pivots instead of once per 64 — the early, HBM-bound updates of a group move half the bytes.  The
//         arithmetic is that of two MODE 0 passes, operation for operation (a separate accumulator per panel, subtracted in panel order),
//         so the pair schedule (MODE 1 + MODE 2) and the plain one (MODE 0 twice) give the same bits.
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every outstanding GLOBAL access of the wave (vmcnt(0)): in the
// tile loop below that would put the write latency of the tile just stored, and the arrival of the operands prefetched for the next one, on
// the critical path of every tile.  Tiles are disjoint in global memory; only the LDS panels are shared between the waves.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0), vmcnt / expcnt untouched
    __builtin_amdgcn_s_barrier();
}
// acc[r] = sum_k Y[16 wc + fk + 4 r][k] L[16 wr + fr][k] for one wavefront: 16 v_mfma_f64_16x16x4_f64 on fragments of two k-fastest LDS panels (row stride
// LDT), lb / yb = LDS byte addresses of this lane's row of the B / A operand panel at k = fk.  MFMA fragments by explicit ds_read_b64 (lane
// (fr, fk) reads row fr, k = 4 kk + fk: dword address 132 fr + 2 fk + 8 kk — the 32 lanes of a half-wave hit 32 distinct bank pairs modulo 64).
// Plain loads would be paired by the compiler into ds_read2_b64 / ds_read_b128, whose lane groups conflict 2-way on this layout.  A ring of two
// register groups of four k-steps: the reads of group g + 2 are issued as soon as the MFMAs of group g have taken their operands, so 16 doubles
// hold the fragments instead of 32; the waits release the loads to the matrix cores in order (LDS returns in order).
__device__ __forceinline__ v4d frag_product(const unsigned lb, const unsigned yb) {
    v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
    double fl[8], fy[8];
#define TR_READ(G, KK0)                                                                                                                    \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                                                      \
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(fl[(G) * 4 + q]) : "v"(lb), "n"(((KK0) + q) * 32) : "memory");                 \
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(fy[(G) * 4 + q]) : "v"(yb), "n"(((KK0) + q) * 32) : "memory");                 \
    }
#define TR_WAIT(N, G)                                                                                                                      \
    asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(fl[(G) * 4]), "+v"(fy[(G) * 4]), "+v"(fl[(G) * 4 + 1]), "+v"(fy[(G) * 4 + 1]),               \
                 "+v"(fl[(G) * 4 + 2]), "+v"(fy[(G) * 4 + 2]), "+v"(fl[(G) * 4 + 3]), "+v"(fy[(G) * 4 + 3]) :: "memory")
#define TR_MFMA(G)                                                                                                                         \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fy[(G) * 4 + q], fl[(G) * 4 + q], acc, 0, 0, 0);
    TR_READ(0, 0) TR_READ(1, 4)
    TR_WAIT(8, 0); TR_MFMA(0)
    TR_READ(0, 8)
    TR_WAIT(8, 1); TR_MFMA(1)
    TR_READ(1, 12)
    TR_WAIT(8, 0); TR_MFMA(0)
    TR_WAIT(0, 1); TR_MFMA(1)
#undef TR_READ
#undef TR_WAIT
#undef TR_MFMA
    return acc;
}

// Z = A(i, panel) M into Zs (LDS, [row i][c fastest], ld LDT): on a change of tile row.  The raw rows travel through `stage` (the buffer the
// column operand uses afterwards) and M (symmetric, 32 KB, in L2 for every workgroup of the launch) through `Ms`, both fetched in ONE batch of
// global loads (the workgroup that carries the pivot chain pays one memory round trip here, not two).
__device__ __forceinline__ void form_Z(const double* __restrict__ Ap, int NP, const double* __restrict__ Mk, double* __restrict__ stage, double* __restrict__ Ms,
                                       double* __restrict__ Zs, int row, int cb, int wr, int wc, int fr, int fk) {
    double av[4], mv[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        av[it] = Ap[row + (size_t)(cb + it * 16) * NP];
        mv[it] = Mk[row + (size_t)(cb + it * 16) * NB];
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        stage[row * LDT + cb + it * 16] = av[it];
        Ms[row * LDT + cb + it * 16] = mv[it];           // Ms[c][k] = M[c][k]
    }
    lds_barrier();
    // Z[i][c] = sum_k A[i][k] M[c][k]: the fragment sequence of the tile product with (Ms, stage) in the places of (Ys, Zs); this lane receives
    // Z(i = 16 wr + fr, c = 16 wc + fk + 4 r)
    const v4d z = frag_product((unsigned)(uintptr_t)(stage + (wr * 16 + fr) * LDT + fk), (unsigned)(uintptr_t)(Ms + (wc * 16 + fr) * LDT + fk));
#pragma unroll
    for (int r = 0; r < 4; ++r) Zs[(wr * 16 + fr) * LDT + wc * 16 + fk + 4 * r] = z[r];
    lds_barrier();                        // Z visible; every read of `stage` / Ms is done (they are refilled next)
}
// ONE grid dimension over all instances of the launch.  Workgroup w runs on XCD w % 8 (dispatch order; used for speed only); the first bt.n
// workgroups take tile 0 of one instance each (and then its diagonal block), the others share the remaining (instance, tile) pairs so that every XCD
// owns a CONTIGUOUS eighth of the instance-major, tile-row-major list and every workgroup of that XCD a contiguous run of it: what runs
// concurrently on an XCD then touches the panels of one or two instances and neighbouring tile rows — they stay in that XCD's 4 MB L2.
template <int MODE>
__global__ __launch_bounds__(TR_THREADS) void k_ldl_step(Batch bt, int NP, int nx, int k0, int ntiles, int tb, double* __restrict__ S, double* __restrict__ Minv,
                                                         double* __restrict__ Dx, double* __restrict__ Tinv, int* __restrict__ icount) {
    constexpr int NH = MODE == 2 ? 2 : 1;   // panels per pass
    __shared__ double smem[step_lds_doubles(NH)];
    double* Zs = smem;                    // Zs[h][i][c]: Z = A(i, panel h) M_h of the current tile row
    double* Ys = smem + NH * TT * LDT;    // Ys[j][k]: rows of the j block of the raw panel (and the staging buffer of form_Z)
    double* Ms = Ys + TT * LDT;           // M of the panel whose Z is being formed
    const int r0 = k0 + NB * NH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const int fr = lane & 15, fk = lane >> 4;
    // staging map: every contiguous 16-lane group (the unit ds_write_b64 is banked over, modulo 32 dwords) covers 8 rows x 2 ADJACENT
    // panel columns: with the row stride LDT = 66 doubles (132 dwords = 4 mod 32) its 16 stores fall on 16 distinct bank pairs; a wave's
    // global load is still 4 columns x 16 consecutive rows = four full 128-byte runs
    const int row = (lane & 7) + 8 * ((lane >> 4) & 1) + 16 * (wave & 3);
    const int cb = ((lane >> 3) & 1) + 2 * ((lane >> 5) & 1) + 4 * (wave >> 2);   // 0..15
    // this workgroup's run [item, item_end) of the (instance, tile) list; tile 0 of every instance has a workgroup of its own
    int t = 0;
    long long item = 0, item_end = 0, off = 0;
    {
        const int nz = bt.n, W = (int)gridDim.x, lin = (int)blockIdx.x;
        if (lin < nz) { off = bt.delta[lin]; }
        else {
            if (ntiles < 2) return;
            const int k = lin & 7;
            const int first = nz + ((k - (nz & 7) + 8) & 7);              // first worker of this XCD (the host grid holds one for every XCD)
            const int u = (lin - first) >> 3, Uk = (W - 1 - first) / 8 + 1;
            const long long G = (long long)nz * (ntiles - 1);
            const long long lo = (long long)k * G / 8, hi = (long long)(k + 1) * G / 8;
            const long long chunk = (hi - lo + Uk - 1) / Uk;
            item = lo + (long long)u * chunk; item_end = item + chunk < hi ? item + chunk : hi;
            if (item >= item_end) return;
            t = 1 + (int)(item % (ntiles - 1)); off = bt.delta[item / (ntiles - 1)];
        }
        S += off; Minv += off;
    }
    const double* Mk = Minv + (size_t)(k0 / NB) * NB * NB;     // M of panel k0 (and of k0 + 64 right behind it)
    int ti, tj;
    if (MODE == 1) { ti = t; tj = 0; } else trailing_tile_index(t, ti, tj);
    int i0 = r0 + ti * TT, j0 = r0 + tj * TT;
    bool newrow = true;
    if (t == 0) LDL_STAMP(r0 / NB, 0);
    double cS[4], yv[NH][4];              // operands of the current tile: the entries of S this lane updates, its share of the raw column panels
#pragma unroll
    for (int r = 0; r < 4; ++r) cS[r] = S[(i0 + wr * 16 + fr) + (size_t)(j0 + wc * 16 + fk + 4 * r) * NP];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
#pragma unroll
        for (int it = 0; it < 4; ++it) yv[h][it] = S[(j0 + row) + (size_t)(k0 + h * NB + cb + it * 16) * NP];
    }
    for (;;) {
        // next tile of this workgroup: its operands travel while the matrix cores work
        bool more = false, nextrow = false;
        long long doff = 0;                                   // slab offset of the next tile's instance relative to the current one
        int tn = t + 1;
        if (t != 0 && item + 1 < item_end) {
            more = true;
            const long long nxt = item + 1;
            tn = 1 + (int)(nxt % (ntiles - 1));
            doff = bt.delta[nxt / (ntiles - 1)] - off;
        }
        int in0 = 0, jn0 = 0;
        double cN[4];
        if (more) {
            int a, b;
            if (MODE == 1) { a = tn; b = 0; } else trailing_tile_index(tn, a, b);
            in0 = r0 + a * TT; jn0 = r0 + b * TT;
            nextrow = in0 != i0 || doff != 0;
        }
        const double* Sn = S + doff;
        if (newrow) {
#pragma unroll
            for (int h = 0; h < NH; ++h)
                form_Z(S + i0 + (size_t)(k0 + h * NB) * NP, NP, Mk + (size_t)h * NB * NB, Ys, Ms, Zs + h * TT * LDT, row, cb, wr, wc, fr, fk);
            if (t == 0) LDL_STAMP(r0 / NB, 1);
        }
#pragma unroll
        for (int h = 0; h < NH; ++h) {
#pragma unroll
            for (int it = 0; it < 4; ++it) Ys[row * LDT + cb + it * 16] = yv[h][it];
            lds_barrier();
            // the registers just staged are free: fetch the SAME panel's share of the next tile into them (one tile = NH units ahead)
            if (more) {
                if (h + 1 == NH) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) cN[r] = Sn[(in0 + wr * 16 + fr) + (size_t)(jn0 + wc * 16 + fk + 4 * r) * NP];
                }
#pragma unroll
                for (int it = 0; it < 4; ++it) yv[h][it] = Sn[(jn0 + row) + (size_t)(k0 + h * NB + cb + it * 16) * NP];
            }
            const v4d acc = frag_product((unsigned)(uintptr_t)(Zs + h * TT * LDT + (wr * 16 + fr) * LDT + fk), (unsigned)(uintptr_t)(Ys + (wc * 16 + fr) * LDT + fk));
#pragma unroll
            for (int r = 0; r < 4; ++r) cS[r] -= acc[r];
            if (h + 1 < NH) lds_barrier();            // the operand reads of the first panel are done before Ys is refilled
        }
        if (t == 0) {
            // tile 0 = the diagonal block of the next panel: hand it over through LDS (row-major, stride LDD) and factor it
            __syncthreads();                      // all MFMA operand reads of Zs/Ys are done
#pragma unroll
            for (int r = 0; r < 4; ++r) smem[(wr * 16 + fr) * LDD + (wc * 16 + fk + 4 * r)] = cS[r];
            __syncthreads();
            Dx += off; Tinv += off; icount += 2 * off;
            diag_block<true>(smem, NP, nx, r0, tb, S, Dx, Tinv, Minv, icount);
            return;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) S[(i0 + wr * 16 + fr) + (size_t)(j0 + wc * 16 + fk + 4 * r) * NP] = cS[r];
        if (!more) return;
        t = tn; i0 = in0; j0 = jn0; newrow = nextrow;
        item += 1; off += doff; S += doff; Minv += doff; Mk += doff;
#pragma unroll
        for (int r = 0; r < 4; ++r) cS[r] = cN[r];
        lds_barrier();                            // the operand reads of this tile are done before LDS is refilled
    }
}

// ---- inverses of the (up to) 1024 x 1024 diagonal blocks from the 64 x 64 ones -----------------------------------------------------------
// inv([A 0; B C]) = [A^-1 0; -C^-1 B A^-1  C^-1].  Level 1 joins 64-blocks into 128-blocks, level 2 joins 128-blocks into
// 256-blocks, ... level 4 joins 512-blocks into 1024-blocks (a trailing 512-block of NP stays as it is).  Each level is two launches of the same 64 x 64-output-tile GEMM (T = B * A^-1, then X21 = -C^-1 * T).
// C_tile(64 x 64) = alpha * A(64 x K) * B(K x 64), K <= 128, operands staged whole in LDS, 1024 threads (16 wavefronts, one
// 16 x 16 MFMA tile each).
struct GemmDesc { const double* A; int lda; const double* B; int ldb; double* C; int ldc; };

// k runs over [kbeg, kend) only (multiples of 64): the triangular operand of a merge is zero outside that range
__device__ __forceinline__ void gemm_tile64(const GemmDesc g, int kbeg, int kend, double alpha, double* smem) {
    constexpr int KC = 128;                 // K chunk held in LDS
    double* As = smem;                      // As[i][k], ld KC+2
    double* Bs = smem + 64 * (KC + 2);      // Bs[j][k]
    const int ldk = KC + 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 2, wj = wave & 3;
    const int fr = lane & 15, fk = lane >> 4;
    v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
    for (int kc = kbeg; kc < kend; kc += KC) {
        const int kn = (kend - kc) < KC ? (kend - kc) : KC;
        __syncthreads();
        for (int idx = tid; idx < 64 * kn; idx += 1024) {
            const int i = idx & 63, kk = idx >> 6;            // A column-major: lanes along i (contiguous)
            As[i * ldk + kk] = g.A[i + (size_t)(kc + kk) * g.lda];
        }
        for (int idx = tid; idx < 64 * kn; idx += 1024) {
            const int kk = idx % kn, j = idx / kn;            // B column-major: lanes along k (contiguous)
            Bs[j * ldk + kk] = g.B[(kc + kk) + (size_t)j * g.ldb];
        }
        __syncthreads();
        for (int kk = 0; kk < kn / 4; ++kk) {
            const double a = As[(wi * 16 + fr) * ldk + kk * 4 + fk];
            const double b = Bs[(wj * 16 + fr) * ldk + kk * 4 + fk];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, acc, 0, 0, 0);   // transposed: row <-> j, col <-> i
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int j = wj * 16 + fk + 4 * r, i = wi * 16 + fr;
        g.C[i + (size_t)j * g.ldc] = alpha * acc[r];
    }
}

// level 1, 2, 3, 4: half = 64, 128, 256, 512; phase 0: T = L21 * X11 ; phase 1: X21 = -X22 * T
__global__ __launch_bounds__(1024) void k_tinv_merge(Batch bt, int NP, int tb, int half, int phase, const double* __restrict__ S, double* __restrict__ Tinv,
                                                      double* __restrict__ Ttmp) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    inst_shift(bt, S, Tinv, Ttmp);
    const int tiles = half / 64;                 // tiles per side of the half x half result
    const int pair = blockIdx.x / (tiles * tiles);
    const int tt = blockIdx.x % (tiles * tiles);
    const int tiy = tt / tiles, tjx = tt % tiles;
    const int g0 = pair * 2 * half;              // first row/col of the pair in the global numbering
    const int q = g0 / tb, o = g0 % tb;
    double* T = Tinv + (size_t)q * tb * tb;
    double* tmp = Ttmp + (size_t)pair * half * half;
    GemmDesc g;
    if (phase == 0) {        // tmp(half x half) = L21 * X11
        g.A = S + (g0 + half + tiy * 64) + (size_t)g0 * NP; g.lda = NP;
        g.B = T + o + (size_t)(o + tjx * 64) * tb; g.ldb = tb;
        g.C = tmp + tiy * 64 + (size_t)(tjx * 64) * half; g.ldc = half;
        gemm_tile64(g, tjx * 64, half, 1.0, smem);          // X11 is lower triangular: rows k < 64 tjx of its column tile are zero
    } else {                 // X21 = -X22 * tmp
        g.A = T + (o + half + tiy * 64) + (size_t)(o + half) * tb; g.lda = tb;
        g.B = tmp + (size_t)(tjx * 64) * half; g.ldb = half;
        g.C = T + (o + half + tiy * 64) + (size_t)(o + tjx * 64) * tb; g.ldc = tb;
        gemm_tile64(g, 0, (tiy + 1) * 64, -1.0, smem);      // X22 is lower triangular: columns k >= 64 (tiy + 1) of its row tile are zero
    }
}

// the panel steps (the pivot chain): NP / 64 launches
static void enqueue_ldl_steps(calipso_hip_solver* s) {
    const int NP = s->d.NP, nblk = NP / NB, tb = trsv_block(NP, (int)s->solve_block);
    const Batch bt = batch_of(s).b;
    const unsigned nz = bt.n;
    double* Minv = s->Ypanel;           // NP x 64: M_k of every panel (the buffer held round 2's unscaled panels)
    hipLaunchKernelGGL(k_ldl_diag, dim3(1, 1, nz), dim3(DIAG_THREADS), 0, s->stream, bt, NP, s->d.nx, 0, tb, s->S, s->Dx, s->Tinv, Minv, s->icount);
    const int band = s->band64 > 0 ? s->band64 : nblk;     // 64-row blocks below a diagonal block that can be non-zero (structure.hip)
    // persistent workgroups: one is resident per CU (registers); 256, 512 and 768 launched workgroups time the same (profiles/README.md), 512 is kept
    static const int resident_total = [] { const char* e = getenv("CALIPSO_HIP_LDL_RESIDENT"); return e && atoi(e) > 0 ? atoi(e) : 512; }();
    const int resident = std::max(2, resident_total / (int)nz);
    // Pair schedule (dense S, several instances per launch): the first tile column of panel k's update (whose tile 0 factors diagonal block
    // k + 1), then BOTH panels in one pass over the rest.  Same arithmetic as the plain schedule (k_ldl_step, MODE 2), half
    // the read-modify-write traffic on the trailing matrix; one instance alone is bound by the pivot chain, not by traffic, and keeps the
    // plain schedule (one launch per panel).
    static const int pairs_env = [] { const char* e = getenv("CALIPSO_HIP_LDL_PAIRS"); return e ? atoi(e) : -1; }();
    const bool pairs = s->band64 == 0 && (pairs_env >= 0 ? pairs_env != 0 : nz >= 4);
    // flattened XCD-aware grid: one workgroup per instance for tile 0 (+ the diagonal block), then workers in multiples of 8 plus 7, so that
    // every XCD (workgroup index mod 8) has at least one worker for its share of the tile list; surplus workgroups leave at once
    auto grid = [&](int tiles) { const int workers = std::min(std::max(tiles - 1, 0), resident) * (int)nz; return dim3(nz + (workers ? (workers + 7) / 8 * 8 + 7 : 0)); };
    for (int kb = 0; kb + 1 < nblk;) {
        const int k0 = kb * NB;
        const int rows = std::min(NP - k0 - NB, band * NB);  // banded S: the panel and its trailing update stop at the band
        const int ntr = rows / TT;
        if (pairs && kb + 2 < nblk) {
            hipLaunchKernelGGL((k_ldl_step<1>), grid(ntr), dim3(TR_THREADS), 0, s->stream, bt, NP, s->d.nx, k0, ntr, tb, s->S, Minv, s->Dx,
                               s->Tinv, s->icount);
            const int ntr2 = ntr - 1, ntiles2 = ntr2 * (ntr2 + 1) / 2;
            hipLaunchKernelGGL((k_ldl_step<2>), grid(ntiles2), dim3(TR_THREADS), 0, s->stream, bt, NP, s->d.nx, k0, ntiles2, tb, s->S, Minv,
                               s->Dx, s->Tinv, s->icount);
            kb += 2;
        } else {
            const int ntiles = ntr * (ntr + 1) / 2;
            // trailing update; its tile 0 also factors the next diagonal block (k0 + 64)
            hipLaunchKernelGGL((k_ldl_step<0>), grid(ntiles), dim3(TR_THREADS), 0, s->stream, bt, NP, s->d.nx, k0, ntiles, tb, s->S, Minv, s->Dx, s->Tinv, s->icount);
            kb += 1;
        }
    }
}

// what follows the chain, fully parallel: the factor columns L = A X' D^-1 of every panel, then the inverses of the triangular-solve blocks
static void enqueue_ldl_finish(calipso_hip_solver* s) {
    const int NP = s->d.NP, nblk = NP / NB, tb = trsv_block(NP, (int)s->solve_block);
    const Batch bt = batch_of(s).b;
    const unsigned nz = bt.n;
    const int band = s->band64 > 0 ? s->band64 : nblk;
    if (nblk > 1) {
        const int maxrows = std::min(NP - NB, band * NB);
        hipLaunchKernelGGL(k_ldl_scale, dim3(maxrows / 64, nblk - 1, nz), dim3(1024), 0, s->stream, bt, NP, tb, band * NB, s->S, s->Dx, s->Tinv);
    }
    const size_t mg_lds = 2 * 64 * (128 + 2) * sizeof(double);
    for (int level = 1; level <= 4; ++level) {
        const int half = 32 << level, tiles = half / 64, pairs = NP / (2 * half);
        if (2 * half > tb) break;
        for (int phase = 0; phase < 2; ++phase)
            hipLaunchKernelGGL(k_tinv_merge, dim3(pairs * tiles * tiles, 1, nz), dim3(1024), mg_lds, s->stream, bt, NP, tb, half, phase, s->S, s->Tinv, s->Ttmp);
    }
}

// ---- triangular solves with blocks of up to 1024 columns ---------------------------------------------------------------------------
// forward:  L u = b.   kernel B_k: u_k = Tinv_k b_k ;  kernel A_k: b_rest -= L[rest, k] u_k
// backward: L' v = z.  kernel B'_k: v_k = Tinv_k' z_k ; kernel A'_k: z_above -= L[k, above]' v_k
// Each output entry is a dot product of a matrix row/column with a block-wide vector; the vector sits in LDS.  Block kb covers columns
// k0 = kb tb .. k0 + w - 1 with w = min(tb, NP - k0): NP = 2560 is 1024 + 1024 + 512.  PARTS = column parts per row (8 for blocks of up to
// 512 columns, 16 for 1024): 32 PARTS threads per workgroup, 64 columns per thread.

// u_k = Tinv_k b_k (lower-triangular mat-vec, lanes along rows); also z_k = u_k / D.  32 rows per workgroup; each lane issues ALL its
// loads before using any (these kernels are latency-bound: one round trip, not four).
template <int PARTS>
__global__ __launch_bounds__(32 * PARTS) void k_trsv_block_n(Batch bt, int kb, int tb, int w, const double* __restrict__ Tinv, const double* __restrict__ b,
                                                              const double* __restrict__ Dx, double* __restrict__ u, double* __restrict__ z) {
    __shared__ double bs[64 * PARTS];
    __shared__ double part[PARTS][32];
    inst_shift(bt, Tinv, b, Dx, u, z);
    const int tid = threadIdx.x, k0 = kb * tb;
    const int r = tid & 31, p = tid >> 5;
    const int row = blockIdx.x * 32 + r;
    const double* T = Tinv + (size_t)kb * tb * tb + row;
    const int cend = blockIdx.x * 32 + 32;         // lower triangular: columns beyond the workgroup's last row are zero
    double v[64];
#pragma unroll
    for (int q = 0; q < 64; ++q) { const int c = p + PARTS * q; v[q] = (c < cend) ? T[(size_t)c * tb] : 0.0; }
    for (int i = tid; i < 64 * PARTS; i += 32 * PARTS) bs[i] = i < w ? b[k0 + i] : 0.0;
    __syncthreads();
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < 64; ++q) acc += v[q] * bs[p + PARTS * q];
    part[p][r] = acc;
    __syncthreads();
    if (tid < 32) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < PARTS; ++q) s += part[q][tid];
        const int gi = k0 + blockIdx.x * 32 + tid;
        u[gi] = s;
        z[gi] = s / Dx[gi];
    }
}

// b[rows below block kb] -= L[rows, block kb] * u_k      (32 rows per workgroup, 64 PARTS = tb columns, one load batch)
template <int PARTS>
__global__ __launch_bounds__(32 * PARTS) void k_trsv_update_n(Batch bt, int NP, int k0, const double* __restrict__ S, const double* __restrict__ u, double* __restrict__ b) {
    __shared__ double us[64 * PARTS];
    __shared__ double part[PARTS][32];
    inst_shift(bt, S, u, b);
    constexpr int W = 64 * PARTS;
    const int tid = threadIdx.x;
    const int r = tid & 31, p = tid >> 5;
    const int row = k0 + W + blockIdx.x * 32 + r;
    const double* Sp = S + row + (size_t)k0 * NP;
    double v[64];
#pragma unroll
    for (int q = 0; q < 64; ++q) v[q] = Sp[(size_t)(p + PARTS * q) * NP];
    for (int i = tid; i < W; i += 32 * PARTS) us[i] = u[k0 + i];
    __syncthreads();
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < 64; ++q) acc += v[q] * us[p + PARTS * q];
    part[p][r] = acc;
    __syncthreads();
    if (tid < 32) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < PARTS; ++q) s += part[q][tid];
        b[k0 + W + blockIdx.x * 32 + tid] -= s;
    }
}

// v_k = Tinv_k' z_k : one wavefront per column (4 columns per workgroup), lanes stride down the column
__global__ __launch_bounds__(256) void k_trsv_block_t(Batch bt, int kb, int tb, int w, const double* __restrict__ Tinv, const double* __restrict__ z, double* __restrict__ v) {
    __shared__ double zs[TB];
    inst_shift(bt, Tinv, z, v);
    const int tid = threadIdx.x, lane = tid & 63, k0 = kb * tb;
    for (int i = tid; i < TB; i += 256) zs[i] = i < w ? z[k0 + i] : 0.0;
    __syncthreads();
    const int c = blockIdx.x * 4 + (tid >> 6);
    const double* T = Tinv + (size_t)kb * tb * tb + (size_t)c * tb;
    double tv[TB / 64];
#pragma unroll
    for (int q = 0; q < TB / 64; ++q) { const int r = lane + 64 * q; tv[q] = (r >= (c & ~63) && r < w) ? T[r] : 0.0; }   // column c is zero above row c
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < TB / 64; ++q) acc += tv[q] * zs[lane + 64 * q];
    acc = wave_sum(acc);
    if (lane == 0) v[k0 + c] = acc;
}

// z[columns left of block kb] -= L[block kb, columns]' v_k : one wavefront per column
__global__ __launch_bounds__(256) void k_trsv_update_t(Batch bt, int NP, int k0, int w, int cfirst, const double* __restrict__ S, const double* __restrict__ v, double* __restrict__ z) {
    __shared__ double vs[TB];
    inst_shift(bt, S, v, z);
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < TB; i += 256) vs[i] = i < w ? v[k0 + i] : 0.0;
    __syncthreads();
    const int c = cfirst + blockIdx.x * 4 + (tid >> 6);     // cfirst <= c < k0 (columns further left are outside the band)
    const double* Lc = S + (size_t)c * NP + k0;
    double lv[TB / 64];
#pragma unroll
    for (int q = 0; q < TB / 64; ++q) lv[q] = (lane + 64 * q < w) ? Lc[lane + 64 * q] : 0.0;
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < TB / 64; ++q) acc += lv[q] * vs[lane + 64 * q];
    acc = wave_sum(acc);
    if (lane == 0) z[c] -= acc;
}

// x (length NP, padded entries zero) <- S^-1 x
static void enqueue_trsv(calipso_hip_solver* s, double* x) {
    const int NP = s->d.NP, tb = trsv_block(NP, (int)s->solve_block), nb = (NP + tb - 1) / tb;
    double* u = s->zf;         // forward result (unscaled), consumed by the updates
    double* z = s->zf2;        // D^-1 u, then overwritten block by block with v
    const Batch bt = batch_of(s).b;
    const unsigned nz = bt.n;
    for (int kb = 0; kb < nb; ++kb) {
        const int k0 = kb * tb, w = std::min(tb, NP - k0);
        if (w > 512) hipLaunchKernelGGL(k_trsv_block_n<16>, dim3(w / 32, 1, nz), dim3(512), 0, s->stream, bt, kb, tb, w, s->Tinv, x, s->Dx, u, z);
        else hipLaunchKernelGGL(k_trsv_block_n<8>, dim3(w / 32, 1, nz), dim3(256), 0, s->stream, bt, kb, tb, w, s->Tinv, x, s->Dx, u, z);
        int rest = NP - (k0 + w);
        if (s->band64 > 0) rest = std::min(rest, ((s->half_bandwidth + 31) / 32) * 32);      // rows below the block that its columns reach
        if (rest > 0) {                                                                       // (a block with rows below it is tb = 512 or 1024 wide)
            if (w > 512) hipLaunchKernelGGL(k_trsv_update_n<16>, dim3(rest / 32, 1, nz), dim3(512), 0, s->stream, bt, NP, k0, s->S, u, x);
            else hipLaunchKernelGGL(k_trsv_update_n<8>, dim3(rest / 32, 1, nz), dim3(256), 0, s->stream, bt, NP, k0, s->S, u, x);
        }
    }
    for (int kb = nb - 1; kb >= 0; --kb) {
        const int k0 = kb * tb, w = std::min(tb, NP - k0);
        hipLaunchKernelGGL(k_trsv_block_t, dim3(w / 4, 1, nz), dim3(256), 0, s->stream, bt, kb, tb, w, s->Tinv, z, x);
        if (kb > 0) {
            const int cfirst = s->band64 > 0 ? std::max(0, ((k0 - s->half_bandwidth) / 4) * 4) : 0;   // columns left of the block that reach into it
            hipLaunchKernelGGL(k_trsv_update_t, dim3((k0 - cfirst) / 4, 1, nz), dim3(256), 0, s->stream, bt, NP, k0, w, cfirst, s->S, x, z);
        }
    }
}

// The factorisation of S and the triangular solves are fixed kernel sequences with fixed arguments (118 and 18 launches):
// they are captured once per handle into hipGraphs and replayed, so the host issues one graph launch instead of queueing every
// kernel (the GPU otherwise waits on the host between the many few-microsecond kernels).
void ldl_set_attributes() {
    static std::once_flag done;      // > 64 KiB of dynamic LDS must be requested explicitly; host lanes may arrive here concurrently
    std::call_once(done, [] { (void)hipFuncSetAttribute((const void*)k_tinv_merge, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * 64 * (128 + 2) * sizeof(double))); });
}

template <typename F>
static bool replay_or_capture(calipso_hip_solver* s, hipGraphExec_t& exec, bool& tried, F enqueue) {
    if (exec) return hipGraphLaunch(exec, s->stream) == hipSuccess;
    if (tried) return false;
    tried = true;
    hipGraph_t graph = nullptr;
    if (hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) return false;
    enqueue();
    if (hipStreamEndCapture(s->stream, &graph) != hipSuccess || !graph) return false;
    const bool ok = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
    (void)hipGraphDestroy(graph);
    if (!ok) { exec = nullptr; return false; }
    return hipGraphLaunch(exec, s->stream) == hipSuccess;
}

#ifdef CALIPSO_LDL_TRACE
}  // namespace calipso
extern "C" int32_t calipso_hip_debug_ldl_trace(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(calipso::g_ldl_trace), sizeof(long long) * 64 * 16) == hipSuccess ? 0 : -1; }
namespace calipso {
#endif

void ldl_drop_graphs(calipso_hip_solver* s) {       // the captured launch sequences depend on the band / the block sizes
    if (s->graph_ldl) { (void)hipGraphExecDestroy(s->graph_ldl); s->graph_ldl = nullptr; }
    if (s->graph_ldl_fin) { (void)hipGraphExecDestroy(s->graph_ldl_fin); s->graph_ldl_fin = nullptr; }
    if (s->graph_trsv) { (void)hipGraphExecDestroy(s->graph_trsv); s->graph_trsv = nullptr; }
    s->graph_ldl_tried = false; s->graph_ldl_fin_tried = false; s->graph_trsv_tried = false;
}

// ev[14] marks the end of the panel steps (the pivot chain), so that their duration can be reported apart from the parallel finish
// (calipso_hip_kernel_times)
void launch_ldl(calipso_hip_solver* s) {
    if (s->stage_parallel && s->spS) {        // stage-parallel: multifrontal LDL^T of S over its nested-dissection tree (sparse.hip)
        const Batch bt = batch_of(s).b;
        if (sparse_factor_from_dense(s->spS, s->stream, bt, s->S, s->spS_src, s->icount) == CALIPSO_OK) { (void)hipEventRecord(s->ev[14], s->stream); return; }
        if (s->compact) { s->err = "structured handle: the multifrontal factorisation was refused and there is no blocked one to fall back to"; (void)hipEventRecord(s->ev[14], s->stream); return; }
        s->stage_parallel = false;            // (a group larger than the reserved batch: back to the blocked factorisation)
        launch_pad_identity(s);               // launch_schur skipped the padding of S for the multifrontal path: the blocked one needs its unit pivots
    }
    ldl_set_attributes();
    // (a group launch covers a changing set of instances: its kernel arguments differ from call to call, so no graph there)
    const bool graphs = !s->cur && s->use_graphs;
    if (!graphs || !replay_or_capture(s, s->graph_ldl, s->graph_ldl_tried, [&] { enqueue_ldl_steps(s); })) enqueue_ldl_steps(s);
    (void)hipEventRecord(s->ev[14], s->stream);
    if (!graphs || !replay_or_capture(s, s->graph_ldl_fin, s->graph_ldl_fin_tried, [&] { enqueue_ldl_finish(s); })) enqueue_ldl_finish(s);
}

void launch_trsv(calipso_hip_solver* s, double* x) {
    if (s->stage_parallel && s->spS) { (void)sparse_solve_inplace(s->spS, s->stream, batch_of(s).b, x); return; }
    if (s->cur || x != s->xbuf || !s->use_graphs || !replay_or_capture(s, s->graph_trsv, s->graph_trsv_tried, [&] { enqueue_trsv(s, s->xbuf); })) enqueue_trsv(s, x);
}

}  // namespace calipso
