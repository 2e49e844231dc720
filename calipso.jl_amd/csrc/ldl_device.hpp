// ldl_device.hpp — device code shared by the two schedules of the blocked LDL^T of the Schur complement: the 64 x 64 diagonal block (the pivot chain) and the
// 64 x 64 x 64 products of the trailing updates.  ldl.hip: right-looking, one launch per panel (groups, banded S); lfac.hip: left-looking, the Schur complement's own
// products folded into the panel launches (one dense system alone).  Design notes: ldl.hip.
#pragma once
#include "internal.hpp"
#include "device_utils.hpp"
#include "pivot16.hpp"

namespace calipso {

typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int TB = 2048;           // largest triangular-solve block; the block actually used is tb = min(opt.solve_block, NP); the last block of a solve may be narrower
constexpr int LDT = NB + 2;        // LDS leading dimension of a k-fastest 64-deep operand panel

// ---- diagonal block ---------------------------------------------------------------------------------------------------------
// The 2500 sequential pivots of S are the critical path of the factorisation; this block is tuned with the stand-alone harnesses
// bench/diag_bench3.hip (this design; profiles/r03_diag_bench3.txt) and bench/diag_bench2.hip / diag_bench.hip (its predecessors: four-column
// mini-panels exchanged through LDS, one barrier each; 13.5 us in the pivot loop).  1024 threads = 16 wavefronts; wavefront (R, C) = (w >> 2, w & 3)
// holds the 16 x 16 tile (R, C) of the block in the accumulator layout of v_mfma_f64_16x16x4 for the whole factorisation (exactly what the
// trailing update of k_ldl_step leaves in its registers: no hand-over).  Four ROUNDS of 16 columns:
//   [A] the tiles of column block r go to LDS (cp), barrier;
//   [owner] wavefront (r, r) takes the 16 columns with lane = row and factors them alone, in registers: no LDS traffic and no barrier between
//       pivots.  The pivot-row entry a rank-1 update needs is fused into the multiply-add by DPP (v_fmac_f64_dpp row_newbcast: lane K of
//       every 16-lane row to all lanes of that row) — for that, the pivot column is read back from LDS, where it goes anyway, as "its rows of
//       the diagonal 16 x 16 block, replicated in every 16-lane row".  The reciprocal chain of the next pivot (v_rcp_f64 + two Newton steps)
//       is threaded by hand through the updates of the current one (a wavefront issues in order).  Unscaled columns (Yk), L (Lk) and the
//       pivots go to LDS, barrier;
//   [C] the tiles right of the block take the rank-16 update on the matrix cores (4 MFMAs per tile).
// X = L11^-1 is assembled meanwhile by the wavefronts that have nothing to do: the 16 x 16 diagonal inverses in-wave by DPP (four helper
// wavefronts, kept off the SIMD of the owner), the blocks below by products on the matrix cores, X_RC = -X_RR (sum_K L_RK X_KC), with the
// inner sum handed from one MFMA chain to the next in registers (the k order of an MFMA is free).  Two short phases remain after the last
// pivot; then M = X' D^-1 X on the matrix cores (what the next panel step multiplies the raw panel with) and ALL global stores: D, L, X, M.
// Nothing is written to global memory before the last barrier (a pending store would make a barrier wait on memory).
// Measured (bench/diag_bench3.hip, one block alone): 14.3 us per launch against 18.2 for the four-column design.
constexpr int DIAG_THREADS = 1024;
constexpr int YS = 18;             // row stride of the 16-column panel of unscaled pivot columns (k fastest; 36 dwords: conflict-free fragment reads)
constexpr int CPS = NB;            // column stride of the column block handed to the next owner

// Optional timeline of the pivot chain (build with -DCALIPSO_LDL_TRACE; bench/ldl_trace.py reads it through calipso_hip_debug_ldl_trace):
// 100 MHz wall-clock stamps of the workgroup that carries tile 0 + the diagonal block, instance 0, per panel step.
#if defined(CALIPSO_LDL_TRACE) && defined(LDL_TRACE_OWNER)
__device__ long long g_ldl_trace[64 * 16];
#define LDL_STAMP(step, slot) do { if (threadIdx.x == 0 && (step) < 64) g_ldl_trace[(step) * 16 + (slot)] = wall_clock64(); } while (0)
// one worker workgroup of the FIRST two-panel pass (k0 == 0): core-clock stamps of its first 32 tiles, 8 slots each
__device__ long long g_ldl_bulk[32 * 8 + 32 * 16 * 2];    // + per wavefront: start / end of the MFMA phase
#define BULK_WAVE_STAMP(slot) do { if (MODE == 2 && bulk_traced && bulk_tile < 32 && (threadIdx.x & 63) == 0) g_ldl_bulk[32 * 8 + (bulk_tile * 16 + (threadIdx.x >> 6)) * 2 + (slot)] = __builtin_readcyclecounter(); } while (0)
#define BULK_STAMP(slot) do { if (MODE == 2 && bulk_traced && bulk_tile < 32 && threadIdx.x == 0) g_ldl_bulk[bulk_tile * 8 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define LDL_STAMP(step, slot) do { } while (0)
#define BULK_STAMP(slot) do { } while (0)
#define BULK_WAVE_STAMP(slot) do { } while (0)
#endif


// LDS carve (doubles): Lk | Yk | cp | XT | XTs | dpiv | dinv
constexpr int DIAG_LDS_DOUBLES = 3 * NB * LDT + NB * YS + 16 * CPS + 2 * NB;

__device__ __forceinline__ void lds_barrier_all() {                  // workgroup barrier that orders LDS traffic only (global stores stay in flight)
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_s_barrier();
}
// -- diagonal 16 x 16 inverse, in-wave by DPP: a helper wavefront grows columns 4 hq .. 4 hq + 3 (lane & 15 = row; the four 16-lane rows compute the
// same).  X = G_14^-1 ... G_0^-1 applied to the identity: x[i] -= L[i][j] x[j] for i > j, j = 0 .. 14 in turn (x[j] by the row broadcast)
template <int J> __device__ __forceinline__ void xrr_steps(double (&x)[4], const double (&nl)[15]) {
    if constexpr (J < 15) {
        asm volatile("v_fmac_f64_dpp %0, %0, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %2, %2, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, %3, %4 row_newbcast:%5 row_mask:0xf bank_mask:0xf"
                     : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) : "v"(nl[J]), "n"(J));      // (three instructions between a write of x[c] and its next DPP read)
        xrr_steps<J + 1>(x, nl);
    }
}
__device__ __forceinline__ void xrr_helper(int r, int hq, int i, const double* __restrict__ Lk, const double* __restrict__ dinv, double* __restrict__ XT,
                                           double* __restrict__ XTs) {
    const int ii = i & 15;
    double nl[15], x[4];
    const double* Lrow = Lk + (16 * r + ii) * LDT + 16 * r;
#pragma unroll
    for (int j = 0; j < 15; ++j) nl[j] = Lrow[j];                       // (all loads in flight before the first use)
#pragma unroll
    for (int j = 0; j < 15; ++j) nl[j] = (ii > j) ? -nl[j] : 0.0;
#pragma unroll
    for (int c = 0; c < 4; ++c) x[c] = (ii == 4 * hq + c) ? 1.0 : 0.0;
    asm volatile("s_nop 1" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
    xrr_steps<0>(x, nl);
    if (i < 16) {
        const double di = dinv[16 * r + ii];
#pragma unroll
        for (int c = 0; c < 4; ++c) { XT[(16 * r + 4 * hq + c) * LDT + 16 * r + ii] = x[c]; XTs[(16 * r + 4 * hq + c) * LDT + 16 * r + ii] = x[c] * di; }
    }
}
// one column of a diagonal inverse per wavefront (after the last pivot every wavefront is free)
template <int J> __device__ __forceinline__ void xrr1_steps(double& x, const double (&nl)[15]) {
    if constexpr (J < 15) {
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(nl[J]), "n"(J));
        xrr1_steps<J + 1>(x, nl);
    }
}
__device__ __forceinline__ void xrr_column(int r, int c, int i, const double* __restrict__ Lk, const double* __restrict__ dinv, double* __restrict__ XT,
                                           double* __restrict__ XTs) {
    const int ii = i & 15;
    double nl[15];
    const double* Lrow = Lk + (16 * r + ii) * LDT + 16 * r;
#pragma unroll
    for (int j = 0; j < 15; ++j) nl[j] = Lrow[j];
#pragma unroll
    for (int j = 0; j < 15; ++j) nl[j] = (ii > j) ? -nl[j] : 0.0;
    double x = (ii == c) ? 1.0 : 0.0;
    xrr1_steps<0>(x, nl);
    if (i < 16) { XT[(16 * r + c) * LDT + 16 * r + ii] = x; XTs[(16 * r + c) * LDT + 16 * r + ii] = x * dinv[16 * r + ii]; }
}
// t += L_RK X_KC (16 x 16 blocks): lane (fr, fk) holds t[q] = (row 16 R + fk + 4 q, column 16 C + fr).  XT[a][r] = X[r][a], Lk[i][k] = L[i][k], both
// k-fastest with stride LDT: the fragment reads of a 32-lane half hit 32 distinct bank pairs
__device__ __forceinline__ v4d blk_LX(v4d t, int Rr, int K, int Cc, const double* __restrict__ Lk, const double* __restrict__ XT, int fr, int fk) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const double f = Lk[(16 * Rr + fr) * LDT + 16 * K + 4 * kk + fk];
        const double s = XT[(16 * Cc + fr) * LDT + 16 * K + 4 * kk + fk];
        t = __builtin_amdgcn_mfma_f64_16x16x4f64(f, s, t, 0, 0, 0);
    }
    return t;
}
// X_RC = -X_RR W_RC with W still in the accumulator registers of the wavefront that formed it (t[q] = W(16 R + fk + 4 q, 16 C + fr)): the matrix
// cores sum over k in any order, so k-step kk takes k = fk + 4 kk — lane (fr, fk) then supplies t[kk] as it stands, and the X_RR operand is read to match
__device__ __forceinline__ void blk_XW(v4d t, int Rr, int Cc, double* __restrict__ XT, double* __restrict__ XTs, const double* __restrict__ dinv, int fr, int fk) {
    v4d x = (v4d){0.0, 0.0, 0.0, 0.0};
    double f[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) f[kk] = XT[(16 * Rr + fk + 4 * kk) * LDT + 16 * Rr + fr];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) x = __builtin_amdgcn_mfma_f64_16x16x4f64(-f[kk], t[kk], x, 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        XT[(16 * Cc + fr) * LDT + 16 * Rr + fk + 4 * q] = x[q];
        XTs[(16 * Cc + fr) * LDT + 16 * Rr + fk + 4 * q] = x[q] * dinv[16 * Rr + fk + 4 * q];
    }
}

// acc: tile (R, C) = (w >> 2, w & 3) of the block, acc[q] = A(16 R + fr, 16 C + fk + 4 q) with fr = lane & 15, fk = lane >> 4 (tiles above the diagonal are
// ignored).  smem: DIAG_LDS_DOUBLES doubles that no wavefront of the workgroup still reads (the caller has a barrier behind its last LDS read).
__device__ __forceinline__ void diag_block(double* __restrict__ smem, v4d acc, int NP, int nx, int k0, int tb, double* __restrict__ S, double* __restrict__ Dx,
                                           double* __restrict__ Tinv, double* __restrict__ Minv, int* __restrict__ icount) {
    double* Lk = smem;                       // Lk[i][k] = L[i][k] (k fastest)
    double* XT = Lk + NB * LDT;              // XT[a][r] = X[r][a]
    double* XTs = XT + NB * LDT;             // ... scaled by the reciprocal pivot of row r
    double* Yk = XTs + NB * LDT;             // Yk[i][j] = unscaled column 16 r + j of the current round
    double* cp = Yk + NB * YS;               // cp[c][i]: column block r after the updates of the rounds before, for its owner
    double* dpiv = cp + 16 * CPS;            // the 64 pivots
    double* dinv = dpiv + NB;                // and their reciprocals
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));            // (opaque: what is derived from the lane index is formed here, not held in registers by the callers across the block — lfac.hip's item loop)
    const int i = tid & 63, w = tid >> 6;
    const int R = w >> 2, C = w & 3, fr = i & 15, fk = i >> 4;
    v4d xacc = (v4d){0.0, 0.0, 0.0, 0.0};   // a block product of the inverse carried from one phase to the next
    LDL_STAMP(k0 / NB, 2);
#pragma unroll 1
    for (int r = 0; r < 4; ++r) {
        if (C == r && R >= r) {
#pragma unroll
            for (int q = 0; q < 4; ++q) cp[(fk + 4 * q) * CPS + 16 * R + fr] = acc[q];
        }
        lds_barrier_all();
        if (w == 5 * r) {
            double a[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) a[c] = cp[c * CPS + i];
            const double y0 = cp[16 * r + (i & 15)];
            Yk[i * YS] = a[0];
            const int lo = __builtin_amdgcn_readlane(__double2loint(a[0]), 16 * r), hi = __builtin_amdgcn_readlane(__double2hiint(a[0]), 16 * r);
            Pivot<0, true>::run(a, (unsigned)(uintptr_t)(Yk + i * YS), (unsigned)(uintptr_t)(Yk + (16 * r + (i & 15)) * YS), Lk + i * LDT + 16 * r, 16 * r,
                          fast_rcp(__hiloint2double(hi, lo)), y0);
            if (i < 16) { const double d = Yk[(16 * r + i) * YS + i]; dpiv[16 * r + i] = d; dinv[16 * r + i] = fast_rcp(d); }   // (its own stores: the LDS queue of a wavefront is in order)
        }
        // while the owner (wavefront 5 r, SIMD r) is busy — nothing else is put on its SIMD: the diagonal inverse of the previous round and the block
        // products whose operands are visible
        if (r == 1) { const int hq = w == 2 ? 0 : w == 3 ? 1 : w == 6 ? 2 : w == 7 ? 3 : -1; if (hq >= 0) xrr_helper(0, hq, i, Lk, dinv, XT, XTs); }
        if (r == 2) { const int hq = w == 1 ? 0 : w == 3 ? 1 : w == 4 ? 2 : w == 8 ? 3 : -1; if (hq >= 0) xrr_helper(1, hq, i, Lk, dinv, XT, XTs); }
        if (r == 3) {
            const int hq = w == 1 ? 0 : w == 2 ? 1 : w == 6 ? 2 : w == 4 ? 3 : -1;
            if (hq >= 0) xrr_helper(2, hq, i, Lk, dinv, XT, XTs);
            if (w == 9) xacc = blk_LX(xacc, 2, 1, 0, Lk, XT, fr, fk);                                                                  // W_20 += L_21 X_10
            if (w == 8) { xacc = blk_LX(xacc, 3, 0, 0, Lk, XT, fr, fk); xacc = blk_LX(xacc, 3, 1, 0, Lk, XT, fr, fk); }               // W_30' = L_30 X_00 + L_31 X_10
            if (w == 12) xacc = blk_LX(xacc, 3, 1, 1, Lk, XT, fr, fk);                                                                // W_31' = L_31 X_11
        }
        lds_barrier_all();
        if (R >= C && C > r) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const double yf = Yk[(16 * C + fr) * YS + 4 * kk + fk];
                const double lf = Lk[(16 * R + fr) * LDT + 16 * r + 4 * kk + fk];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-yf, lf, acc, 0, 0, 0);
            }
        }
        if (r == 2) {
            if (w == 7) { const v4d t = blk_LX((v4d){0.0, 0.0, 0.0, 0.0}, 1, 0, 0, Lk, XT, fr, fk); blk_XW(t, 1, 0, XT, XTs, dinv, fr, fk); }   // X_10 = -X_11 (L_10 X_00)
            if (w == 9) xacc = blk_LX(xacc, 2, 0, 0, Lk, XT, fr, fk);                                                                 // W_20' = L_20 X_00
            if (w == 4) xacc = blk_LX(xacc, 2, 1, 1, Lk, XT, fr, fk);                                                                 // W_21 = L_21 X_11
        }
    }
    LDL_STAMP(k0 / NB, 3);
    // after the last pivot: X_33 (one column per wavefront), X_20, X_21, W_32; then X_30, X_31, X_32
    if (w != 9 && w != 4 && w != 7) xrr_column(3, w, i, Lk, dinv, XT, XTs);
    if (w == 0) xrr_column(3, 9, i, Lk, dinv, XT, XTs);
    if (w == 1) xrr_column(3, 4, i, Lk, dinv, XT, XTs);
    if (w == 2) xrr_column(3, 7, i, Lk, dinv, XT, XTs);
    if (w == 9) blk_XW(xacc, 2, 0, XT, XTs, dinv, fr, fk);
    if (w == 4) blk_XW(xacc, 2, 1, XT, XTs, dinv, fr, fk);
    if (w == 7) xacc = blk_LX(xacc, 3, 2, 2, Lk, XT, fr, fk);                                                                             // W_32 = L_32 X_22
    lds_barrier_all();
    if (w == 8) { xacc = blk_LX(xacc, 3, 2, 0, Lk, XT, fr, fk); blk_XW(xacc, 3, 0, XT, XTs, dinv, fr, fk); }
    if (w == 12) { xacc = blk_LX(xacc, 3, 2, 1, Lk, XT, fr, fk); blk_XW(xacc, 3, 1, XT, XTs, dinv, fr, fk); }
    if (w == 7) blk_XW(xacc, 3, 2, XT, XTs, dinv, fr, fk);
    lds_barrier_all();
    LDL_STAMP(k0 / NB, 4);
    // M = X' D^-1 X = (L11 D L11')^-1: what the NEXT launch multiplies the raw panel with.  M[a][b] = sum_r X[r][a] X[r][b] / d[r] on the matrix
    // cores: wavefront (wa, wb) forms the 16 x 16 tile (rows a, columns b); both operand fragments are "row a (b), k index r" reads of X'.
    {
        const int wa = R, wb = C;
        v4d m = (v4d){0.0, 0.0, 0.0, 0.0};
        // X[r][a] = 0 for r < a: the k blocks above the later of the two tile origins contribute exact zeros and are skipped (the workgroup's 256
        // MFMAs shrink to 120; the matrix cores of one CU are what bounds this product)
        for (int kk = 4 * (wa > wb ? wa : wb); kk < NB / 4; ++kk) {
            const double xa = XTs[(wa * 16 + fr) * LDT + 4 * kk + fk];
            const double xb = XT[(wb * 16 + fr) * LDT + 4 * kk + fk];
            m = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, xb, m, 0, 0, 0);
        }
        double* Mo = Minv + (size_t)(k0 / NB) * NB * NB;
        LDL_STAMP(k0 / NB, 6);
#pragma unroll
        for (int q = 0; q < 4; ++q) Mo[(wb * 16 + fr) + (size_t)(wa * 16 + fk + 4 * q) * NB] = m[q];   // lane holds M(a = fk + 4 q, b = fr) = M(b, a): 128-byte runs along fr
    }
    // everything that goes to global memory leaves here, after the last barrier: D and the inertia counts (compute_inertia!), the strictly
    // lower L of the block and X = L11^-1 on the diagonal of the triangular-solve inverse block (zeros above), both from their LDS copies:
    // thread (i, w) stores row i of columns 4 w .. 4 w + 3
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
        for (int c = 0; c < 4; ++c) {
            const int k = 4 * w + c;
            T[(o + i) + (size_t)(o + k) * tb] = (i >= k) ? XT[k * LDT + i] : 0.0;
            if (i > k) S[(k0 + i) + (size_t)(k0 + k) * NP] = Lk[i * LDT + k];
        }
    }
    LDL_STAMP(k0 / NB, 5);
}

// ---- panel step: A22 -= (A21 M) A21' ---------------------------------------------------------------------------------------------
// 64 x 64 tile of the lower triangle per workgroup of 1024 threads (16 wavefronts, one 16 x 16 MFMA tile each).  ONE workgroup is resident
// per CU (registers).  Small tiles keep all 256 CUs busy on the shrinking trailing matrix.  The tile is computed transposed (MFMA row <->
// column j of S) so result stores are 128-byte runs.
constexpr int TR_THREADS = 1024;
constexpr int TT = 64;
constexpr int step_lds_doubles(int nh) { return (nh + 2) * TT * LDT > DIAG_LDS_DOUBLES ? (nh + 2) * TT * LDT : DIAG_LDS_DOUBLES; }   // Zs[nh] | Ys | Ms
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
                                       double* __restrict__ Zs, int row, int cb, int wr, int wc, int fr, int fk, v4d* __restrict__ zout = nullptr) {
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
    if (zout) *zout = z;                  // (lfac.hip keeps Z = A(i, panel) M for the later, deferred updates of row i)
    lds_barrier();                        // Z visible; every read of `stage` / Ms is done (they are refilled next)
}

}  // namespace calipso
