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
#define LDL_TRACE_OWNER      // the timeline stamps of -DCALIPSO_LDL_TRACE live in this translation unit
#include "ldl_device.hpp"
#include "host_logic.hpp"

#include <algorithm>
#include <array>
#include <chrono>
#include <mutex>

namespace calipso {


__global__ __launch_bounds__(DIAG_THREADS) void k_ldl_diag(Batch bt, int NP, int nx, int k0, int tb, double* __restrict__ S, double* __restrict__ Dx,
                                                            double* __restrict__ Tinv, double* __restrict__ Minv, int* __restrict__ icount) {
    __shared__ double smem[DIAG_LDS_DOUBLES];
    inst_shift(bt, S, Dx, Tinv, Minv);
    inst_shift_i(bt, icount);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, R = w >> 2, C = w & 3, fr = lane & 15, fk = lane >> 4;
    v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
    if (R >= C) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = S[(k0 + 16 * R + fr) + (size_t)(k0 + 16 * C + fk + 4 * q) * NP];
    }
    diag_block(smem, acc, NP, nx, k0, tb, S, Dx, Tinv, Minv, icount);
}

// ---- factor columns: L21 = A21 X' D^-1 for every panel in one launch -------------------------------------------------------------
// After the last panel step the sub-diagonal tiles of S still hold the RAW panel columns A(i,k) (final Schur-complement values); the
// factor is L(i,k) = A(i,k) X_k' D_k^-1.  D[c][r] = sum_k X[c][k] A21[r][k]: MFMA A operand = X (rows c), B operand = A21' so that the
// 16-lane fast index of the result is the contiguous row index r of the column-major panel.  One workgroup (16 wavefronts) per
// 64 x 64 tile (blockIdx.x = tile row below the panel, blockIdx.y = panel); wavefront (wr, wc) computes the 16 x 16 tile rows 16 wr..,
// columns 16 wc..; X is staged in LDS.  Off the critical path: 780 independent tiles at C3.
__global__ __launch_bounds__(1024) void k_ldl_scale(Batch bt, int NP, int tb, int band_rows, int p0, const double* S, double* Lout, const double* __restrict__ Dx,
                                                     const double* __restrict__ Tinv) {
    __shared__ double Xs[NB * LDT];   // Xs[c][k]
    __shared__ double dinv[NB];
    const int k0 = (p0 + (int)blockIdx.y) * NB;       // panels p0 .. p0 + gridDim.y - 1
    const int rows = min(NP - k0 - NB, band_rows);           // banded S: the panel stops at the band
    if ((int)blockIdx.x * 64 >= rows) return;
    inst_shift(bt, S, Lout, Dx, Tinv);
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
        Lout[row + (size_t)(k0 + c) * NP] = acc[r] * dinv[c];     // (Lout == S: in place — every lane holds its raw entries since the barrier; lfac.hip: a buffer of its own)
    }
}

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
// ONE grid dimension over all instances of the launch.  Workgroup w runs on XCD w % 8 (dispatch order; used for speed only); the first bt.n
// workgroups take tile 0 of one instance each (and then its diagonal block), the others share the remaining (instance, tile) pairs so that every XCD
// owns a CONTIGUOUS eighth of the instance-major, tile-row-major list and every workgroup of that XCD a contiguous run of it: what runs
// concurrently on an XCD then touches the panels of one or two instances and neighbouring tile rows — they stay in that XCD's 4 MB L2.
template <int MODE>
__global__ __launch_bounds__(TR_THREADS) void k_ldl_step(Batch bt, int NP, int nx, int k0, int ntiles, int tb, double* __restrict__ S, double* __restrict__ Minv,
                                                         double* __restrict__ Dx, double* __restrict__ Tinv, int* __restrict__ icount,
                                                         unsigned long long* __restrict__ hprog = nullptr, unsigned long long ptag = 0) {
    constexpr int NH = MODE == 2 ? 2 : 1;   // panels per pass
    // progress word for the host (mapped memory; launch_ldl): this step has started, so everything the steps before it wrote is complete
    if (hprog && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(hprog, ptag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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
        if (lin < nz) {
            off = bt.delta[lin];
        } else {
            if (ntiles < 2) return;
            const int k = lin & 7;
            const int first = nz + ((k - (nz & 7) + 8) & 7);              // first worker of this XCD (the host grid holds one for every XCD)
            const int u = (lin - first) >> 3, Uk = (W - 1 - first) / 8 + 1;
            const long long G = (long long)nz * (ntiles - 1);
            if (G <= 0) return;
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
    // lane offsets inside a tile as 32-bit integers (NP^2 < 2^31), tile origins as wave-uniform pointers: scalar base + vector offset addressing,
    // half the address registers
    const int offC = (wr * 16 + fr) + (wc * 16 + fk) * NP, offY = row + cb * NP;
    double cS[4], yv[NH][4];              // operands of the current tile: the entries of S this lane updates, its share of the raw column panels
#ifdef CALIPSO_LDL_TRACE
    const bool bulk_traced = k0 == 0 && (int)blockIdx.x == (int)bt.n + 16;
    int bulk_tile = 0;
#endif
#pragma unroll
    for (int r = 0; r < 4; ++r) cS[r] = (S + (i0 + (size_t)j0 * NP))[offC + 4 * r * NP];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
#pragma unroll
        for (int it = 0; it < 4; ++it) yv[h][it] = (S + (j0 + (size_t)(k0 + h * NB) * NP))[offY + it * 16 * NP];
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
        BULK_STAMP(0);
        if (newrow) {
#pragma unroll
            for (int h = 0; h < NH; ++h)
                form_Z(S + i0 + (size_t)(k0 + h * NB) * NP, NP, Mk + (size_t)h * NB * NB, Ys, Ms, Zs + h * TT * LDT, row, cb, wr, wc, fr, fk);
            if (t == 0) LDL_STAMP(r0 / NB, 1);
        }
        BULK_STAMP(1);
        if constexpr (NH == 2) {
            // both column panels are staged at once (the second one in the buffer form_Z uses for M: free between two changes of tile row): two
            // barriers per tile instead of four, and the two products run back to back on the matrix cores.  Same arithmetic, same order.
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int it = 0; it < 4; ++it) (h ? Ms : Ys)[row * LDT + cb + it * 16] = yv[h][it];
            }
            BULK_STAMP(2);
            lds_barrier();
            BULK_STAMP(3);
            BULK_WAVE_STAMP(0);
            if (more) {
#pragma unroll
                for (int r = 0; r < 4; ++r) cN[r] = (Sn + (in0 + (size_t)jn0 * NP))[offC + 4 * r * NP];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
#pragma unroll
                    for (int it = 0; it < 4; ++it) yv[h][it] = (Sn + (jn0 + (size_t)(k0 + h * NB) * NP))[offY + it * 16 * NP];
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const v4d acc = frag_product((unsigned)(uintptr_t)(Zs + h * TT * LDT + (wr * 16 + fr) * LDT + fk), (unsigned)(uintptr_t)((h ? Ms : Ys) + (wc * 16 + fr) * LDT + fk));
#pragma unroll
                for (int r = 0; r < 4; ++r) cS[r] -= acc[r];
            }
            BULK_STAMP(4);
            BULK_WAVE_STAMP(1);
        } else {
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            // tile 0 (the next diagonal block): its column operand IS its row operand, which form_Z has just staged in Ys and whose readers it has waited for —
            // nothing to stage, no barrier (0.4 us less on the pivot chain of every panel)
            if (t != 0) {
#pragma unroll
                for (int it = 0; it < 4; ++it) Ys[row * LDT + cb + it * 16] = yv[h][it];
                lds_barrier();
            }
            // the registers just staged are free: fetch the SAME panel's share of the next tile into them (one tile = NH units ahead)
            if (more) {
                if (h + 1 == NH) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) cN[r] = (Sn + (in0 + (size_t)jn0 * NP))[offC + 4 * r * NP];
                }
#pragma unroll
                for (int it = 0; it < 4; ++it) yv[h][it] = (Sn + (jn0 + (size_t)(k0 + h * NB) * NP))[offY + it * 16 * NP];
            }
            const v4d acc = frag_product((unsigned)(uintptr_t)(Zs + h * TT * LDT + (wr * 16 + fr) * LDT + fk), (unsigned)(uintptr_t)(Ys + (wc * 16 + fr) * LDT + fk));
#pragma unroll
            for (int r = 0; r < 4; ++r) cS[r] -= acc[r];
            if (h + 1 < NH) lds_barrier();            // the operand reads of the first panel are done before Ys is refilled
        }
        }
        if (t == 0) {
            // tile 0 = the diagonal block of the next panel: its 16 x 16 tiles are already where the diagonal block wants them (accumulator layout,
            // wavefront (wr, wc)); one barrier: every operand read of Zs / Ys is done before the block reuses the LDS
            lds_barrier();
            Dx += off; Tinv += off; icount += 2 * off;
            diag_block(smem, (v4d){cS[0], cS[1], cS[2], cS[3]}, NP, nx, r0, tb, S, Dx, Tinv, Minv, icount);
            return;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) (S + (i0 + (size_t)j0 * NP))[offC + 4 * r * NP] = cS[r];
        if (!more) {
            return;
        }
        t = tn; i0 = in0; j0 = jn0; newrow = nextrow;
        item += 1; off += doff; S += doff; Minv += doff; Mk += doff;
#pragma unroll
        for (int r = 0; r < 4; ++r) cS[r] = cN[r];
        BULK_STAMP(5);
        lds_barrier();                            // the operand reads of this tile are done before LDS is refilled
        BULK_STAMP(6);
#ifdef CALIPSO_LDL_TRACE
        ++bulk_tile;
#endif
    }
}

// ---- inverses of the (up to) 1024 x 1024 diagonal blocks from the 64 x 64 ones -----------------------------------------------------------
// inv([A 0; B C]) = [A^-1 0; -C^-1 B A^-1  C^-1].  Level 1 joins 64-blocks into 128-blocks, level 2 joins 128-blocks into
// 256-blocks, ... level 4 joins 512-blocks into 1024-blocks (a trailing 512-block of NP stays as it is).  Each level is two launches of one small matrix-core GEMM
// (T = B * A^-1, then X21 = -C^-1 * T) with 32 x 32 output tiles and 256 threads (4 wavefronts, one 16 x 16 MFMA tile each): many small tiles spread over the
// compute units and eight workgroups per unit hide each other's load latency (round 2's 64 x 64 tiles — 13.6 us for one tile at K = 512 — lasted as long as their
// longest tile).  The skipped k ranges are exact zeros of the triangular operand.
// C(32 x 32) = alpha * A(32 x K) * B(K x 32) over k in [kbeg, kend) (multiples of 32), 256 threads; As / Bs: 32 * 66 doubles of LDS each
__device__ __forceinline__ void merge32_tile(const double* __restrict__ A, int lda, const double* __restrict__ B, int ldb, double* __restrict__ Cc, int ldc, int kbeg, int kend,
                                             double alpha, double* __restrict__ As, double* __restrict__ Bs) {
    constexpr int KC = 64, ldk = KC + 2;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int fr = lane & 15, fk = lane >> 4;
    v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
    // K in chunks of 64 (a last chunk of 32: kbeg, kend are multiples of 32); the operands of chunk c + 1 travel from global memory to registers while the
    // matrix cores work on chunk c (round 3: KC = 32 without the prefetch left every stage one exposed memory round trip — a 1024-deep tile took ≈ 50 us)
    double av[8], bv[8];
    auto fetch = [&](int kc) {
        const int kn = kend - kc < KC ? kend - kc : KC;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int ka = (tid >> 5) + 8 * it;                                           // A: lanes along i (32 rows), 8 k per pass
            av[it] = ka < kn ? A[(tid & 31) + (size_t)(kc + ka) * lda] : 0.0;
            const int kb = tid & 63, cb = (tid >> 6) + 4 * it;                            // B: lanes along k (64 consecutive), 4 columns per pass
            bv[it] = kb < kn ? B[(kc + kb) + (size_t)cb * ldb] : 0.0;
        }
    };
    fetch(kbeg);
    for (int kc = kbeg; kc < kend; kc += KC) {
        const int kn = kend - kc < KC ? kend - kc : KC;
        __syncthreads();                                                                 // the previous chunk's fragments are read
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            As[(tid & 31) * ldk + (tid >> 5) + 8 * it] = av[it];
            Bs[((tid >> 6) + 4 * it) * ldk + (tid & 63)] = bv[it];
        }
        __syncthreads();
        if (kc + KC < kend) fetch(kc + KC);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const double a = As[(wi * 16 + fr) * ldk + kk * 4 + fk];
            const double b = Bs[(wj * 16 + fr) * ldk + kk * 4 + fk];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, acc, 0, 0, 0);   // transposed: row <-> j, col <-> i
        }
        if (kn > 32) {
#pragma unroll
            for (int kk = 8; kk < 16; ++kk) {
                const double a = As[(wi * 16 + fr) * ldk + kk * 4 + fk];
                const double b = Bs[(wj * 16 + fr) * ldk + kk * 4 + fk];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, acc, 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int j = wj * 16 + fk + 4 * r, i = wi * 16 + fr;
        Cc[i + (size_t)j * ldc] = alpha * acc[r];
    }
}

__global__ __launch_bounds__(256) void k_tinv_merge32(Batch bt, int NP, int tb, int half, int phase, int pair0, const double* __restrict__ S, double* __restrict__ Tinv,
                                                       double* __restrict__ Ttmp) {
    __shared__ double As[32 * 66];       // As[i][k]
    __shared__ double Bs[32 * 66];       // Bs[j][k]
    inst_shift(bt, S, Tinv, Ttmp);
    const int tiles = half / 32;
    const int pair = pair0 + blockIdx.x / (tiles * tiles);     // pairs pair0 .. of this level (a launch may cover one solve block only)
    const int tt = blockIdx.x % (tiles * tiles);
    const int tiy = tt / tiles, tjx = tt % tiles;
    const int g0 = pair * 2 * half;
    const int q = g0 / tb, o = g0 % tb;
    double* T = Tinv + (size_t)q * tb * tb;
    double* tmp = Ttmp + (size_t)pair * half * half;
    if (phase == 0)          // tmp(half x half) = L21 * X11; X11 is lower triangular: rows k < 32 tjx of its column tile are zero
        merge32_tile(S + (g0 + half + tiy * 32) + (size_t)g0 * NP, NP, T + o + (size_t)(o + tjx * 32) * tb, tb, tmp + tiy * 32 + (size_t)(tjx * 32) * half, half, tjx * 32, half, 1.0, As, Bs);
    else                     // X21 = -X22 * tmp; X22 is lower triangular: columns k >= 32 (tiy + 1) of its row tile are zero
        merge32_tile(T + (o + half + tiy * 32) + (size_t)(o + half) * tb, tb, tmp + (size_t)(tjx * 32) * half, half, T + (o + half + tiy * 32) + (size_t)(o + tjx * 32) * tb, tb, 0,
                     (tiy + 1) * 32, -1.0, As, Bs);
}

// The same product by a FEW persistent workgroups of 1024 threads (64 x 64 output tiles, 16 wavefronts with one 16 x 16 MFMA tile each, K in chunks of 64 with the next
// chunk's operands prefetched into registers): what runs on the second stream BESIDE the pivot chain.  A 1024-thread workgroup owns a compute unit, exactly like a
// worker of k_ldl_step: G of them leave 256 - G compute units to the panel steps, whereas the 1536 small workgroups of the 32 x 32 form land on every compute unit and
// keep the next panel step's workgroups (which need whole units) waiting for 20-35 us.  Tiles in the order of decreasing depth (column tile 0 has K = w).
__global__ __launch_bounds__(1024) void k_wform_product64(Batch bt, int NP, int tb, int kb, int w, int rb, int rows, const double* __restrict__ S, const double* __restrict__ Tinv, double* __restrict__ Wb) {
    constexpr int KC = 64, ldk = KC + 2;
    __shared__ double As[64 * ldk];       // As[i][k]
    __shared__ double Bs[64 * ldk];       // Bs[j][k]
    inst_shift(bt, S, Tinv, Wb);
    const int tr = rows / 64, ntiles = tr * (w / 64), k0 = kb * tb;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 2, wj = wave & 3, fr = lane & 15, fk = lane >> 4;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tiy = t % tr, tjx = t / tr;
        const double* A = S + (k0 + w + tiy * 64) + (size_t)k0 * NP;                                // A[i + k * NP] = L[k0 + w + 64 tiy + i, k0 + k]
        const double* B = Tinv + (size_t)kb * tb * tb + (size_t)(tjx * 64) * tb;                    // B[k + j * tb] = Tinv_kb[k, 64 tjx + j]: zero for k < 64 tjx
        const int kbeg = tjx * 64;
        v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
        double av[4], bv[4];
        auto fetch = [&](int kc) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                av[it] = A[lane + (size_t)(kc + wave + 16 * it) * NP];                              // lanes along i, 16 k per pass
                bv[it] = B[(kc + lane) + (size_t)(wave + 16 * it) * tb];                            // lanes along k, 16 columns per pass
            }
        };
        fetch(kbeg);
        for (int kc = kbeg; kc < w; kc += KC) {
            __syncthreads();                                                                        // the previous chunk's fragments are read
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                As[lane * ldk + wave + 16 * it] = av[it];
                Bs[(wave + 16 * it) * ldk + lane] = bv[it];
            }
            __syncthreads();
            if (kc + KC < w) fetch(kc + KC);
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const double a = As[(wi * 16 + fr) * ldk + kk * 4 + fk];
                const double b = Bs[(wj * 16 + fr) * ldk + kk * 4 + fk];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, acc, 0, 0, 0);                     // transposed: row <-> j, col <-> i
            }
        }
        double* Cc = Wb + tiy * 64 + (size_t)(tjx * 64) * rb;
#pragma unroll
        for (int r = 0; r < 4; ++r) Cc[(wi * 16 + fr) + (size_t)(wj * 16 + fk + 4 * r) * rb] = acc[r];
    }
}

// W-form block of solve block kb (internal.hpp: wform_offset): W = L[k0 + w .. NP, k0 .. k0 + w) * Tinv_kb, rb = NP - k0 - w rows, leading dimension rb.  One 32 x 32 tile per
// workgroup (blockIdx.x = tile row + (rb / 32) * tile column); Tinv_kb is lower triangular: the k range of column tile tjx starts at 32 tjx.
__global__ __launch_bounds__(256) void k_wform_product(Batch bt, int NP, int tb, int kb, int w, int rb, int rows, const double* __restrict__ S, const double* __restrict__ Tinv, double* __restrict__ Wb) {
    __shared__ double As[32 * 66];
    __shared__ double Bs[32 * 66];
    inst_shift(bt, S, Tinv, Wb);
    const int tr = rows / 32, tiy = blockIdx.x % tr, tjx = blockIdx.x / tr, k0 = kb * tb;
    merge32_tile(S + (k0 + w + tiy * 32) + (size_t)k0 * NP, NP, Tinv + (size_t)kb * tb * tb + (size_t)(tjx * 32) * tb, tb, Wb + tiy * 32 + (size_t)(tjx * 32) * rb, rb, tjx * 32, w, 1.0, As, Bs);
}

// The LAST solve block has nothing below it: its forward step u = Tinv b, z = u / D and its backward step v = Tinv' z are two dependent launches on the same small block.
// With Msym = Tinv' D^-1 Tinv (= the inverse of the block's Schur complement, symmetric, w x w) they are ONE mat-vec v = Msym b (k_block_sym): a solve is 2 nb - 1 launches.
// Msym is formed once per factorisation, when the block's inverse is complete (C3: 512 x 512, 0.09 GF): tile (a0, b0) = sum_r T[r][a] T[r][b] / d[r] over r >= max(a0, b0)
// (T is lower triangular; the rows scaled by the reciprocal pivots), both operands fetched with the lanes along r; 32 x 32 tiles, 256 threads, the chunk loop of merge32_tile.
__global__ __launch_bounds__(256) void k_lastblock_sym(Batch bt, int tb, int kb, int w, const double* __restrict__ Tinv, const double* __restrict__ Dx, double* __restrict__ Msym) {
    constexpr int KC = 128, ldk = KC + 2;       // (chunks of 128: a tile is a chain of load -> barrier -> 32 MFMAs per wavefront; the fewer links the better — 21 us with 64, K = 512)
    __shared__ double As[32 * ldk];
    __shared__ double Bs[32 * ldk];
    inst_shift(bt, Tinv, Dx, Msym);
    const int tr = w / 32, ta = blockIdx.x % tr, tbj = blockIdx.x / tr, k0 = kb * tb;
    const double* TA = Tinv + (size_t)kb * tb * tb + (size_t)(ta * 32) * tb;
    const double* TB = Tinv + (size_t)kb * tb * tb + (size_t)(tbj * 32) * tb;
    const double* dd = Dx + k0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1, fr = lane & 15, fk = lane >> 4;
    const int kbeg = 32 * (ta > tbj ? ta : tbj);
    v4d acc = (v4d){0.0, 0.0, 0.0, 0.0}, acc1 = (v4d){0.0, 0.0, 0.0, 0.0};       // (two chains: one wavefront per SIMD, nothing else hides the latency of a dependent MFMA)
    double av[16], bv[16], dv[2];
    auto fetch = [&](int kc) {                                                             // (loads only: the reciprocal pivots are formed when the chunk is parked in LDS)
        const int kn = w - kc < KC ? w - kc : KC;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int kr = lane + 64 * h;                                                  // lanes along r (64 consecutive), 4 columns per pass, two halves of the chunk
            dv[h] = kr < kn ? dd[kc + kr] : 1.0;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int cb = wave + 4 * it;
                av[8 * h + it] = kr < kn ? TA[(kc + kr) + (size_t)cb * tb] : 0.0;
                bv[8 * h + it] = kr < kn ? TB[(kc + kr) + (size_t)cb * tb] : 0.0;
            }
        }
    };
    fetch(kbeg);
    for (int kc = kbeg; kc < w; kc += KC) {
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const double dk = 1.0 / dv[h];                                                 // (one division per row and chunk: the row's reciprocal pivot)
#pragma unroll
            for (int it = 0; it < 8; ++it) { As[(wave + 4 * it) * ldk + lane + 64 * h] = av[8 * h + it] * dk; Bs[(wave + 4 * it) * ldk + lane + 64 * h] = bv[8 * h + it]; }
        }
        __syncthreads();
        if (kc + KC < w) fetch(kc + KC);
#pragma unroll
        for (int kk = 0; kk < KC / 4; ++kk) {                                              // (rows beyond the block are zeros in LDS)
            const double a = As[(wi * 16 + fr) * ldk + kk * 4 + fk];
            const double b = Bs[(wj * 16 + fr) * ldk + kk * 4 + fk];
            if (kk & 1) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, acc1, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_f64_16x16x4f64(b, a, acc, 0, 0, 0);           // transposed: row <-> column index b, col <-> row index a
        }
    }
    acc += acc1;
    double* Cc = Msym + ta * 32 + (size_t)(tbj * 32) * w;
#pragma unroll
    for (int r = 0; r < 4; ++r) Cc[(wi * 16 + fr) + (size_t)(wj * 16 + fk + 4 * r) * w] = acc[r];
}

__global__ void k_publish_inertia(const int* __restrict__ icount, int* __restrict__ hcount, unsigned long long* __restrict__ hseq, unsigned long long seq) {
    if (threadIdx.x < 6) hcount[threadIdx.x] = icount[threadIdx.x];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(hseq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// second stream + events + progress word of the handle (created on first use; lowest priority: the pivot chain's workgroups are scheduled first.  Confining it
// to a quarter of the compute units — hipExtStreamCreateWithCUMask — changed nothing: the chain's workgroup does not wait for a CU)
// CALIPSO_HIP_GRAPH_LDL=1: the panel steps and the finish as captured graphs on ONE stream (no second stream, no left-looking schedule): the reference point of
// tests/test_gpu_ldl_overlap.py
static bool graph_ldl_requested() {
    static const bool env = [] { const char* e = getenv("CALIPSO_HIP_GRAPH_LDL"); return e && atoi(e) != 0; }();
    return env;
}
static bool side_stream(calipso_hip_solver* s) {
    if (s->stream2) return s->hprog_dev != nullptr && s->ev_side[7] != nullptr;
    // (lowest priority.  A CU mask — hipExtStreamCreateWithCUMask, half / three quarters / 15 of 16 of the compute units — is accepted and changes
    // nothing, neither the duration of the second stream's kernels nor the panel steps that wait behind them: see DESIGN 5.0)
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return false;
    hipStream_t st = nullptr;
    if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, least) != hipSuccess) return false;
    bool ok = true;
    if (!s->hprog) {
        if (hipHostMalloc((void**)&s->hprog, 4 * sizeof(unsigned long long), hipHostMallocMapped) != hipSuccess) { s->hprog = nullptr; ok = false; }     // [0] updates, [1] chain
        else {
            s->hprog[0] = 0; s->hprog[1] = 0;
            if (hipHostGetDevicePointer((void**)&s->hprog_dev, s->hprog, 0) != hipSuccess) { s->hprog_dev = nullptr; ok = false; }
        }
    }
    for (auto& e : s->ev_side)
        if (ok && !e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { e = nullptr; ok = false; }
    if (!ok) { (void)hipStreamDestroy(st); return false; }      // no half-made second stream: the factorisation keeps the one-stream schedule
    s->stream2 = st;
    return true;
}
static void enqueue_finish_feed(calipso_hip_solver* s, hipStream_t stream, int f);
static void enqueue_wform(calipso_hip_solver* s, hipStream_t stream, int kb);
static void enqueue_feed(calipso_hip_solver* s, hipStream_t stream, int i, bool beside);
static int wform_rows(const calipso_hip_solver* s, int rb);
static void enqueue_lastblock_sym(calipso_hip_solver* s, hipStream_t stream);
static void ldl_plan_ranges(calipso_hip_solver* s);
// Which finish work runs beside the chain: one instance alone, dense S, a solve block that is complete before the chain ends (otherwise nothing to overlap)
static bool ldl_overlap(calipso_hip_solver* s) {
    static const int env = [] { const char* e = getenv("CALIPSO_HIP_LDL_OVERLAP"); return e ? atoi(e) : 1; }();
    // (groups: measured in round 4 — one group of 12 alone 952 -> 962 steps/s, three groups in flight 1098 -> 1086: the other groups already fill the chain's idle
    // time, and the finish as 60 small launches per group costs them more than the overlap gives: groups keep one stream)
    if (!env || graph_ldl_requested() || s->cur || s->band64 > 0) return false;
    const int NP = s->d.NP, tb = trsv_block(NP, (int)s->solve_block);
    if (NP < 1024 || NP > 8192 || tb < 256) return false;
    return true;
}

// the panel steps (the pivot chain): NP / 64 launches
static void enqueue_ldl_steps(calipso_hip_solver* s) {
    const int NP = s->d.NP, nblk = NP / NB, tb = trsv_block(NP, (int)s->solve_block);
    const Batch bt = batch_of(s).b;
    const unsigned nz = bt.n;
    double* Minv = s->Ypanel;           // NP x 64: M_k of every panel (the buffer held round 2's unscaled panels)
    const bool lfac = lfac_ready(s);      // one dense system alone: the left-looking schedule of lfac.hip (the same chain, the Schur complement's products under it)
    if (!lfac) hipLaunchKernelGGL(k_ldl_diag, dim3(1, 1, nz), dim3(DIAG_THREADS), 0, s->stream, bt, NP, s->d.nx, 0, tb, s->S, s->Dx, s->Tinv, Minv, s->icount);
    const int band = s->band64 > 0 ? s->band64 : nblk;     // 64-row blocks below a diagonal block that can be non-zero (structure.hip)
    // persistent workgroups: one is resident per CU (registers, LDS): 248 workers + the workgroups that carry the diagonal blocks = everything resident
    // at once, 31 + 1 per XCD for one instance.  Since round 3's diagonal block (19 us) the early launches of ONE instance are bound by the trailing update,
    // and a second wave of workgroups costs 5 us per launch (21.5 against 26.4 us at 512); a group of 12 gains 2 % (2.50 against 2.56 ms per factorisation).
    const int resident = std::max(2, 248 / (int)nz);
    // Pair schedule (dense S, several instances per launch): the first tile column of panel k's update (whose tile 0 factors diagonal block
    // k + 1), then BOTH panels in one pass over the rest.  Same arithmetic as the plain schedule (k_ldl_step, MODE 2), half
    // the read-modify-write traffic on the trailing matrix; one instance alone is bound by the pivot chain, not by traffic, and keeps the
    // plain schedule (one launch per panel).
    const bool pairs = s->band64 == 0 && nz >= 4;
    // flattened XCD-aware grid: one workgroup per instance for tile 0 (+ the diagonal block), then workers in multiples of 8 plus 7, so that
    // every XCD (workgroup index mod 8) has at least one worker for its share of the tile list; surplus workgroups leave at once
    auto grid = [&](int tiles) { const int workers = std::min(std::max(tiles - 1, 0), resident) * (int)nz; return dim3(nz + (workers ? (workers + 7) / 8 * 8 + 7 : 0)); };
    const bool overlap = ldl_overlap(s) && side_stream(s);
    s->ldl_overlap_on = overlap;
    if (overlap) ldl_plan_ranges(s);
    int launches = 1;
    unsigned long long* const hprog = overlap ? s->hprog_dev : (unsigned long long*)nullptr;
    const unsigned long long epoch = s->ldl_epoch << 16;         // progress word = epoch | first panel the launch applies: every panel before it is released
    s->lfac_last = lfac;
    if (lfac) launches = lfac_enqueue(s, hprog, epoch);
    for (int kb = 0; !lfac && kb + 1 < nblk;) {
        const int k0 = kb * NB;
        const int rows = std::min(NP - k0 - NB, band * NB);  // banded S: the panel and its trailing update stop at the band
        const int ntr = rows / TT;
        if (pairs && kb + 2 < nblk) {
            hipLaunchKernelGGL((k_ldl_step<1>), grid(ntr), dim3(TR_THREADS), 0, s->stream, bt, NP, s->d.nx, k0, ntr, tb, s->S, Minv, s->Dx,
                               s->Tinv, s->icount, hprog, epoch | (unsigned long long)kb);
            const int ntr2 = ntr - 1, ntiles2 = ntr2 * (ntr2 + 1) / 2;
            hipLaunchKernelGGL((k_ldl_step<2>), grid(ntiles2), dim3(TR_THREADS), 0, s->stream, bt, NP, s->d.nx, k0, ntiles2, tb, s->S, Minv,
                               s->Dx, s->Tinv, s->icount);
            kb += 2; launches += 2;
        } else {
            const int ntiles = ntr * (ntr + 1) / 2;
            // trailing update; its tile 0 also factors the next diagonal block (k0 + 64)
            hipLaunchKernelGGL((k_ldl_step<0>), grid(ntiles), dim3(TR_THREADS), 0, s->stream, bt, NP, s->d.nx, k0, ntiles, tb, s->S, Minv, s->Dx, s->Tinv, s->icount,
                               hprog, epoch | (unsigned long long)kb);
            kb += 1; launches += 1;
        }
    }
    s->ldl_step_launches = launches;
    // A range of columns can be finished on the second stream as soon as a launch that applies none of its panels has STARTED (the raw panel columns are
    // only read by the launch that applies them): launch_ldl watches the progress word for that.  The launches carry the tags 0 .. nblk - 2 (a pair pass the
    // tag of its first panel), so the ranges that end at or before panel nblk - 2 are handed over while the chain runs, the rest after it (enqueue_ldl_finish).
    int forks = 0;
    if (overlap) while (3 * forks < (int)s->ldl_feeds.size() && s->ldl_feeds[3 * forks] <= nblk - 2) ++forks;
    s->ldl_forks = forks;              // (enqueue_ldl_finish joins the second stream)
    // A single handle outside a stream capture: the six inertia counts go to their mapped host words right behind the chain (+ the sequence number the host
    // spins on, api.hip: wait_published) — the host learns the inertia when the pivot chain ends, not after the finish + a copy + a stream synchronisation
    // (as arguments of the panel-step kernel itself the three words cost every launch 0.7 us).
    if (s->ldl_publish && nz == 1) {
        s->ldl_pub_seq = ++s->pub_seq;
        hipLaunchKernelGGL(k_publish_inertia, dim3(1), dim3(64), 0, s->stream, s->icount, s->hicount_dev, s->hseq_dev, s->ldl_pub_seq);
    }
}

// what follows the chain, fully parallel: the factor columns L = A X' D^-1 of every panel, then the inverses of the triangular-solve blocks
static void enqueue_ldl_finish(calipso_hip_solver* s) {
    const int NP = s->d.NP, nblk = NP / NB, tb = trsv_block(NP, (int)s->solve_block);
    const Batch bt = batch_of(s).b;
    const unsigned nz = bt.n;
    const int band = s->band64 > 0 ? s->band64 : nblk;
    {
        if (s->ldl_overlap_on) {
            // the blocks whose last panel a panel step applied were finished beside the chain (launch_ldl); what is left is the last block(s).
            // Join first: the solves need every block (the second stream is long done by now).  (The last block's finish on the second stream too, joined
            // after the inertia read-back, was measured: 0.901 against 0.889 ms — the hand-over between the queues costs more than the overlap gives.)
            if (s->ldl_forks || s->rhs_ahead) {
                (void)hipEventRecord(s->ev_side[7], s->stream2);
                (void)hipStreamWaitEvent(s->stream, s->ev_side[7], 0);
                s->rhs_joined = true;
            }
            for (int f = s->ldl_forks; 3 * f < (int)s->ldl_feeds.size(); ++f) enqueue_feed(s, s->stream, f, false);
            enqueue_lastblock_sym(s, s->stream);
            return;
        }
    }
    if (nblk > 1) {
        const int maxrows = std::min(NP - NB, band * NB);
        hipLaunchKernelGGL(k_ldl_scale, dim3(maxrows / 64, nblk - 1, nz), dim3(1024), 0, s->stream, bt, NP, tb, band * NB, 0, s->S, s->Lf, s->Dx, s->Tinv);
    }
    for (int level = 1; level <= 5; ++level) {
        const int half = 32 << level, tiles = half / 64, pairs = NP / (2 * half);
        if (2 * half > tb) break;
        for (int phase = 0; phase < 2; ++phase)
            hipLaunchKernelGGL(k_tinv_merge32, dim3(pairs * tiles * tiles * 4, 1, nz), dim3(256), 0, s->stream, bt, NP, tb, half, phase, 0, s->Lf, s->Tinv, s->Ttmp);
    }
    if (wform_on(s)) for (int kb = 0; kb * tb < NP; ++kb) enqueue_wform(s, s->stream, kb);
    enqueue_lastblock_sym(s, s->stream);
}

// The same finish for ONE solve block b (columns b tb .. b tb + w - 1) on `stream`: the factor columns of its panels, then the merges of its inverse blocks.
// Everything it reads is final once the panel step that applies the block's LAST panel has completed (the raw panel columns are only read by the step
// that applies them; the X and D of a diagonal block are stored one step earlier), and what it writes — the scaled columns, the off-diagonal parts
// of Tinv_b, its pairs of Ttmp — nothing else touches: one instance alone leaves most of the chip idle during a panel step (one 20 us pivot chain, a
// shrinking trailing update), so the finish of the completed blocks runs THERE, on a second stream, instead of after the chain.
// The finish as a walk over the merge tree of the solve blocks.  The columns are cut into RANGES of FEED columns (a solve block narrower than that is one
// range); when a range is complete (the panel step that applies its last panel has finished) the second stream gets, in this order:
//   the factor columns of its panels; the merges inside the range (both phases, level by level);
//   then up the tree: the range (or the node it has just completed) is either the LEFT child of its parent — the parent's first phase T = L21 X11 needs
//   nothing else, it is queued and the walk stops — or the RIGHT child — the parent's second phase X21 = -X22 T is queued, the parent is complete, the walk
//   goes on with it.
// So when the chain ends only the last range and the second phases along the right edge of the last block are left.  A first phase may wait for its second
// one while other merges run: every level has a scratch area of its own (merge_scratch).  Measured at C3 (ms per factorisation / per step): ranges of 64,
// 128, 256 columns 0.868 / 2.418, 0.872 / 2.417, 0.868 / 2.422 (whole solve blocks of 1024: 0.889 / 2.43); with 2048-wide solve blocks 0.93-0.96 / 2.415-2.445:
// the second phase of the 1024 -> 2048 merge and the right edge below it arrive together after step 31 and the second stream runs late, which costs
// the factorisation what the six-launch solves save (solve + refinement 0.78 against 0.84 ms) — opt.solve_block stays 1024.
constexpr int FEED = 256;
static size_t merge_scratch(int NP, int half) { return (size_t)NP * (size_t)(half - 64) / 2; }   // a level holds NP / (2 half) products of half x half = NP half / 2 doubles; the levels below: NP (32 + 64 + ... + half / 4); all five: NP * 992
static void enqueue_merge(calipso_hip_solver* s, hipStream_t stream, int half, int phase, int pair0, int pairs) {
    const int NP = s->d.NP, tb = trsv_block(NP, (int)s->solve_block), tiles = half / 32;
    const Batch bt = batch_of(s).b;
    hipLaunchKernelGGL(k_tinv_merge32, dim3(pairs * tiles * tiles, 1, bt.n), dim3(256), 0, stream, bt, NP, tb, half, phase, pair0, s->Lf, s->Tinv, s->Ttmp + merge_scratch(NP, half));
}
// workgroups of the W-form product beside the chain (k_wform_product64): it is handed over once the trailing update has shrunk enough to leave them their compute units
constexpr int WFORM_WGS = 128;
static void ldl_plan_ranges(calipso_hip_solver* s) {
    const int NP = s->d.NP, nblk = NP / NB, tb = trsv_block(NP, (int)s->solve_block);
    s->ldl_ranges.clear();
    s->ldl_feeds.clear();
    const bool wf = wform_on(s);
    std::vector<std::array<int, 3>> feeds;
    for (int b0 = 0; b0 < NP; b0 += tb) {
        const int wblk = std::min(tb, NP - b0), wr = std::min(FEED, wblk);
        for (int c = b0; c < b0 + wblk; c += wr) {
            feeds.push_back({(c + wr) / NB, 0, (int)s->ldl_ranges.size() / 2});
            s->ldl_ranges.push_back(c); s->ldl_ranges.push_back(wr);
        }
        if (wf && b0 + wblk < NP) {
            // the block's product: after its last range, and not before the panel step whose trailing update (tiles + the chain's workgroup) leaves WFORM_WGS compute units free
            int step = (b0 + wblk) / NB;
            while (step < nblk - 1) { const int ntr = nblk - 1 - step; if (ntr * (ntr + 1) / 2 + WFORM_WGS <= 248) break; ++step; }
            feeds.push_back({step, 1, b0 / tb});
        }
    }
    std::stable_sort(feeds.begin(), feeds.end(), [](const std::array<int, 3>& a, const std::array<int, 3>& b) { return a[0] < b[0]; });
    for (const auto& f : feeds) { s->ldl_feeds.push_back(f[0]); s->ldl_feeds.push_back(f[1]); s->ldl_feeds.push_back(f[2]); }
}
// feed i of the plan on `stream`; beside: the pivot chain is still running (the product takes its few-workgroup form)
static void enqueue_feed(calipso_hip_solver* s, hipStream_t stream, int i, bool beside) {
    const int kind = s->ldl_feeds[3 * i + 1], arg = s->ldl_feeds[3 * i + 2];
    if (kind == 0) { enqueue_finish_feed(s, stream, arg); return; }
    if (!beside) { enqueue_wform(s, stream, arg); return; }
    const int NP = s->d.NP, tb = trsv_block(NP, (int)s->solve_block);
    const int k0 = arg * tb, w = std::min(tb, NP - k0), rb = NP - k0 - w;
    const Batch bt = batch_of(s).b;
    const int rows = wform_rows(s, rb);
    hipLaunchKernelGGL(k_wform_product64, dim3(std::min(WFORM_WGS, (rows / 64) * (w / 64)), 1, bt.n), dim3(1024), 0, stream, bt, NP, tb, arg, w, rb, rows, s->Lf, s->Tinv, s->Wfac + wform_offset(NP, tb, arg));
}
static void enqueue_finish_feed(calipso_hip_solver* s, hipStream_t stream, int f) {
    const int NP = s->d.NP, nblk = NP / NB, tb = trsv_block(NP, (int)s->solve_block);
    const Batch bt = batch_of(s).b;
    const unsigned nz = bt.n;
    const int c0 = s->ldl_ranges[2 * f], w = s->ldl_ranges[2 * f + 1];
    const int b0 = c0 / tb * tb, wblk = std::min(tb, NP - b0);
    const int p0 = c0 / NB, np = std::min(w / NB, nblk - 1 - p0);                     // (the last panel has no rows below it)
    if (np > 0) hipLaunchKernelGGL(k_ldl_scale, dim3((NP - p0 * NB - NB) / 64, np, nz), dim3(1024), 0, stream, bt, NP, tb, NP, p0, s->S, s->Lf, s->Dx, s->Tinv);
    for (int half = 64; 2 * half <= w; half *= 2)
        for (int phase = 0; phase < 2; ++phase) enqueue_merge(s, stream, half, phase, c0 / (2 * half), w / (2 * half));
    int rel = c0 - b0;
    bool block_done = w == wblk;                            // (a range that is a whole solve block)
    for (int cur = w; 2 * cur <= wblk; cur *= 2) {          // the complete node: columns b0 + rel .. + cur - 1
        const int pair = (b0 + (rel & ~(2 * cur - 1))) / (2 * cur);
        if ((rel & (2 * cur - 1)) == 0) { enqueue_merge(s, stream, cur, 0, pair, 1); break; }
        enqueue_merge(s, stream, cur, 1, pair, 1);
        rel &= ~(2 * cur - 1);
        if (2 * cur == wblk) block_done = true;             // the node just completed is the whole block
    }
    (void)block_done;     // (the block's W-form product is a feed of its own in the plan: ldl_plan_ranges)
}

// ---- triangular solves with blocks of up to 1024 columns ---------------------------------------------------------------------------
// forward:  L u = b.   kernel B_k: u_k = Tinv_k b_k ;  kernel A_k: b_rest -= L[rest, k] u_k
// backward: L' v = z.  kernel B'_k: v_k = Tinv_k' z_k ; kernel A'_k: z_above -= L[k, above]' v_k
// Each output entry is a dot product of a matrix row/column with a block-wide vector; the vector sits in LDS.  Block kb covers columns
// k0 = kb tb .. k0 + w - 1 with w = min(tb, NP - k0): NP = 2560 is 1024 + 1024 + 512.  ROWS rows per workgroup, PARTS column parts per row,
// CPT columns per thread and pass: ROWS * PARTS threads cover PARTS * CPT columns per pass (16 x 32 x 32 for blocks wider than 512, 16 x 16 x 32 otherwise).

// u_k = Tinv_k b_k (lower-triangular mat-vec, lanes along rows); also z_k = u_k / D.  Each lane issues ALL its
// loads before using any (these kernels are latency-bound: one round trip, not four).
template <int ROWS, int PARTS, int CPT>
__global__ __launch_bounds__(ROWS * PARTS) void k_trsv_block_n(Batch bt, int kb, int tb, int w, const double* __restrict__ Tinv, const double* __restrict__ b,
                                                                const double* __restrict__ Dx, double* __restrict__ u, double* __restrict__ z) {
    constexpr int W = PARTS * CPT;                // columns per pass (a 2048-wide block takes two passes of 1024; the rows of its upper half only the first)
    __shared__ double bs[W];
    __shared__ double part[PARTS][ROWS];
    inst_shift(bt, Tinv, b, Dx, u, z);
    const int tid = threadIdx.x, k0 = kb * tb;
    const int r = tid % ROWS, p = tid / ROWS;
    const int row = blockIdx.x * ROWS + r;
    const double* T = Tinv + (size_t)kb * tb * tb + row;
    const int cend = blockIdx.x * ROWS + ROWS;     // lower triangular: columns beyond the workgroup's last row are zero
    double acc = 0.0;
    for (int c0 = 0; c0 < cend; c0 += W) {
        double v[CPT];
#pragma unroll
        for (int q = 0; q < CPT; ++q) { const int c = c0 + p + PARTS * q; v[q] = (c < cend) ? T[(size_t)c * tb] : 0.0; }
        if (c0) __syncthreads();                  // the previous pass has read bs
        for (int i = tid; i < W; i += ROWS * PARTS) bs[i] = c0 + i < w ? b[k0 + c0 + i] : 0.0;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < CPT; ++q) acc += v[q] * bs[p + PARTS * q];
    }
    part[p][r] = acc;
    __syncthreads();
    if (tid < ROWS) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < PARTS; ++q) s += part[q][tid];
        const int gi = k0 + blockIdx.x * ROWS + tid;
        u[gi] = s;
        z[gi] = s / Dx[gi];
    }
}

// b[rows below block kb] -= L[rows, block kb] * u_k      (ROWS rows per workgroup, passes of PARTS * CPT columns over the w columns of the block)
template <int ROWS, int PARTS, int CPT>
__global__ __launch_bounds__(ROWS * PARTS) void k_trsv_update_n(Batch bt, int NP, int k0, int w, const double* __restrict__ S, const double* __restrict__ u, double* __restrict__ b) {
    constexpr int W = PARTS * CPT;
    __shared__ double us[W];
    __shared__ double part[PARTS][ROWS];
    inst_shift(bt, S, u, b);
    const int tid = threadIdx.x;
    const int r = tid % ROWS, p = tid / ROWS;
    const int row = k0 + w + blockIdx.x * ROWS + r;
    double acc = 0.0;
    for (int c0 = 0; c0 < w; c0 += W) {
        const double* Sp = S + row + (size_t)(k0 + c0) * NP;
        double v[CPT];
#pragma unroll
        for (int q = 0; q < CPT; ++q) v[q] = Sp[(size_t)(p + PARTS * q) * NP];
        if (c0) __syncthreads();
        for (int i = tid; i < W; i += ROWS * PARTS) us[i] = u[k0 + c0 + i];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < CPT; ++q) acc += v[q] * us[p + PARTS * q];
    }
    part[p][r] = acc;
    __syncthreads();
    if (tid < ROWS) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < PARTS; ++q) s += part[q][tid];
        b[k0 + w + blockIdx.x * ROWS + tid] -= s;
    }
}

// v_k = Tinv_k' z_k : one wavefront per column (4 columns per workgroup), lanes stride down the column.  TBW = widest block of the layout (1024 or 2048)
template <int TBW>
__global__ __launch_bounds__(256) void k_trsv_block_t(Batch bt, int kb, int tb, int w, const double* __restrict__ Tinv, const double* __restrict__ z, double* __restrict__ v) {
    __shared__ double zs[TBW];
    inst_shift(bt, Tinv, z, v);
    const int tid = threadIdx.x, lane = tid & 63, k0 = kb * tb;
    for (int i = tid; i < TBW; i += 256) zs[i] = i < w ? z[k0 + i] : 0.0;
    __syncthreads();
    const int c = blockIdx.x * 4 + (tid >> 6);
    const double* T = Tinv + (size_t)kb * tb * tb + (size_t)c * tb;
    double tv[TBW / 64];
#pragma unroll
    for (int q = 0; q < TBW / 64; ++q) { const int r = lane + 64 * q; tv[q] = (r >= (c & ~63) && r < w) ? T[r] : 0.0; }   // column c is zero above row c
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < TBW / 64; ++q) acc += tv[q] * zs[lane + 64 * q];
    acc = wave_sum(acc);
    if (lane == 0) v[k0 + c] = acc;
}

// z[columns left of block kb] -= L[block kb, columns]' v_k : one wavefront per column
template <int TBW>
__global__ __launch_bounds__(256) void k_trsv_update_t(Batch bt, int NP, int k0, int w, int cfirst, const double* __restrict__ S, const double* __restrict__ v, double* __restrict__ z) {
    __shared__ double vs[TBW];
    inst_shift(bt, S, v, z);
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < TBW; i += 256) vs[i] = i < w ? v[k0 + i] : 0.0;
    __syncthreads();
    const int c = cfirst + blockIdx.x * 4 + (tid >> 6);     // cfirst <= c < k0 (columns further left are outside the band)
    const double* Lc = S + (size_t)c * NP + k0;
    double lv[TBW / 64];
#pragma unroll
    for (int q = 0; q < TBW / 64; ++q) lv[q] = (lane + 64 * q < w) ? Lc[lane + 64 * q] : 0.0;
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < TBW / 64; ++q) acc += lv[q] * vs[lane + 64 * q];
    acc = wave_sum(acc);
    if (lane == 0) z[c] -= acc;
}

// ---- W-form solves (internal.hpp: wform_offset): one launch per solve block and direction -------------------------------------------------
// forward, block kb: the stacked matrix [Tinv_kb; W_kb] (w + rb rows) times b_kb: rows < w give u_kb (and z_kb = u_kb / D), rows >= w are subtracted from the
// right-hand side below the block.  ROWS rows per workgroup, the first w / ROWS workgroups take the triangular part (as k_trsv_block_n), the others W.
template <int ROWS, int PARTS, int CPT>
__global__ __launch_bounds__(ROWS * PARTS) void k_trsv_fwd(Batch bt, int kb, int tb, int w, int rb, const double* __restrict__ Tinv, const double* __restrict__ Wb, double* __restrict__ b,
                                                            const double* __restrict__ Dx, double* __restrict__ u, double* __restrict__ z, const int* __restrict__ gate = nullptr,
                                                            int gate_epoch = 0, int snap = 0) {
    if (gate && gate[0] == gate_epoch) return;        // a round queued ahead of a refinement that has converged meanwhile (internal.hpp: gate)
    constexpr int W = PARTS * CPT;
    __shared__ double bs[W];
    __shared__ double part[PARTS][ROWS];
    inst_shift(bt, Tinv, Wb, b, Dx, u, z);
    const int tid = threadIdx.x, k0 = kb * tb;
    const int r = tid % ROWS, p = tid / ROWS;
    const int nA = w / ROWS;
    const bool below = (int)blockIdx.x >= nA;
    const int blk = below ? (int)blockIdx.x - nA : (int)blockIdx.x;
    const int row = blk * ROWS + r;
    if (below && snap && blk * ROWS >= snap - 1) {        // (snap = 1 + the rows of W_kb that can be non-zero: a banded S — the rows beyond take no update, only the copy)
        if (tid < ROWS) { const int gi = k0 + w + blk * ROWS + tid; u[gi] = b[gi]; }
        return;
    }
    const double* M = below ? Wb + row : Tinv + (size_t)kb * tb * tb + row;
    const size_t ld = below ? (size_t)rb : (size_t)tb;
    const int cend = below ? w : blk * ROWS + ROWS;     // lower triangular: columns beyond the workgroup's last row are zero
    // what the row's result is combined with (its right-hand side entry below the block, its pivot inside it) travels with the matrix loads: fetched where it is used,
    // behind the last barrier, it was one more memory round trip at the end of a 6 us kernel
    double tail_operand = 0.0;
    if (tid < ROWS) { const int gi0 = k0 + (below ? w : 0) + blk * ROWS + tid; tail_operand = below ? b[gi0] : Dx[gi0]; }
    double acc = 0.0;
    for (int c0 = 0; c0 < cend; c0 += W) {
        double v[CPT];
#pragma unroll
        for (int q = 0; q < CPT; ++q) { const int c = c0 + p + PARTS * q; v[q] = (c < cend) ? M[(size_t)c * ld] : 0.0; }
        if (c0) __syncthreads();
        for (int i = tid; i < W; i += ROWS * PARTS) bs[i] = c0 + i < w ? b[k0 + c0 + i] : 0.0;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < CPT; ++q) acc += v[q] * bs[p + PARTS * q];
    }
    part[p][r] = acc;
    __syncthreads();
    if (tid < ROWS) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < PARTS; ++q) s += part[q][tid];
        const int gi = k0 + (below ? w : 0) + blk * ROWS + tid;
        if (below) {                              // (rows >= k0 + w: no workgroup of this launch reads them)
            const double nv = tail_operand - s;
            b[gi] = nv;
            if (snap) u[gi] = nv;                 // the block before the last one: a copy of the final right-hand side of the last block for k_block_sym (which writes b in place)
        } else { u[gi] = s; z[gi] = s / tail_operand; }
    }
}
// backward, block kb: v_kb = [Tinv_kb; W_kb]' [z_kb; -v_below]: one wavefront per column, lanes stride down the stacked column (w rows of Tinv_kb from the diagonal
// on, then rb rows of W_kb), every load of a lane in flight before the first use.  NCH = 64-row groups of the stacked column (w + rb <= 64 NCH).  x holds v of the
// blocks below (written by the launches before this one); the block's own v goes to x[k0 ..].
template <int NCH>
__global__ __launch_bounds__(256) void k_trsv_bwd(Batch bt, int kb, int tb, int w, int rb, int rows, const double* __restrict__ Tinv, const double* __restrict__ Wb, const double* __restrict__ z,
                                                   double* __restrict__ x, const int* __restrict__ gate = nullptr, int gate_epoch = 0) {
    if (gate && gate[0] == gate_epoch) return;
    __shared__ double zs[NCH * 64];
    inst_shift(bt, Tinv, Wb, z, x);
    const int tid = threadIdx.x, lane = tid & 63, k0 = kb * tb;
    const int c = blockIdx.x * 4 + (tid >> 6);
    const double* T = Tinv + (size_t)kb * tb * tb + (size_t)c * tb;
    const double* Wc = Wb + (size_t)c * rb;
    // the column's loads go out FIRST (they do not depend on the vector): the fill of zs below and its barrier run under their latency
    double tv[NCH];
#pragma unroll
    for (int q = 0; q < NCH; ++q) {
        const int r = lane + 64 * q;
        tv[q] = r < w ? (r >= (c & ~63) ? T[r] : 0.0) : (r < w + rows ? Wc[r - w] : 0.0);     // column c of Tinv_kb is zero above row c; rows <= rb: what of W_kb can be non-zero
    }
    {
        constexpr int PER = NCH * 64 / 256;
        double zv[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) { const int i = tid + 256 * u; zv[u] = i < w ? z[k0 + i] : (i < w + rows ? -x[k0 + i] : 0.0); }
#pragma unroll
        for (int u = 0; u < PER; ++u) zs[tid + 256 * u] = zv[u];
    }
    __syncthreads();
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < NCH; ++q) acc += tv[q] * zs[lane + 64 * q];
    acc = wave_sum(acc);
    if (lane == 0) x[k0 + c] = acc;
}
template <int NCH>
static void launch_trsv_bwd(hipStream_t st, unsigned nz, const Batch& bt, int kb, int tb, int w, int rb, int rows, const double* Tinv, const double* Wb, const double* z, double* x,
                            const int* gate, int ge) {
    hipLaunchKernelGGL(k_trsv_bwd<NCH>, dim3(w / 4, 1, nz), dim3(256), 0, st, bt, kb, tb, w, rb, rows, Tinv, Wb, z, x, gate, ge);
}

// v = Msym b for the last solve block (k_lastblock_sym): 16 rows per workgroup, the column parts of a row combined through LDS as in k_trsv_fwd
template <int ROWS, int PARTS, int CPT>
__global__ __launch_bounds__(ROWS * PARTS) void k_block_sym(Batch bt, int k0, int w, const double* __restrict__ Msym, const double* __restrict__ in, double* __restrict__ x,
                                                             const int* __restrict__ gate = nullptr, int gate_epoch = 0) {
    if (gate && gate[0] == gate_epoch) return;
    constexpr int W = PARTS * CPT;
    __shared__ double bs[W];
    __shared__ double part[PARTS][ROWS];
    inst_shift(bt, Msym, in, x);
    const int tid = threadIdx.x, r = tid % ROWS, p = tid / ROWS;
    const int row = blockIdx.x * ROWS + r;
    const double* M = Msym + row;
    double acc = 0.0;
    for (int c0 = 0; c0 < w; c0 += W) {
        double v[CPT];
#pragma unroll
        for (int q = 0; q < CPT; ++q) { const int c = c0 + p + PARTS * q; v[q] = (c < w) ? M[(size_t)c * w] : 0.0; }
        if (c0) __syncthreads();
        for (int i = tid; i < W; i += ROWS * PARTS) bs[i] = c0 + i < w ? in[k0 + c0 + i] : 0.0;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < CPT; ++q) acc += v[q] * bs[p + PARTS * q];
    }
    part[p][r] = acc;
    __syncthreads();
    if (tid < ROWS) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < PARTS; ++q) s += part[q][tid];
        x[k0 + blockIdx.x * ROWS + tid] = s;
    }
}

bool wform_on(const calipso_hip_solver* s) {
    const int NP = s->d.NP, tb = trsv_block(NP, (int)s->solve_block);
    return s->solve_wform != 0 && !s->compact && !(s->stage_parallel && s->spS) && wform_layout_ok(NP, tb);
}
// the last solve block through its symmetric inverse (k_lastblock_sym / k_block_sym): with the W-form
static const int LASTSYM = [] { const char* e = getenv("CALIPSO_HIP_LASTBLOCK_SYM"); return e ? atoi(e) : 1; }();
bool lastblock_sym_on(const calipso_hip_solver* s) {
    if (!LASTSYM || !wform_on(s)) return false;
    const int NP = s->d.NP, tb = trsv_block(NP, (int)s->solve_block), nb = (NP + tb - 1) / tb;
    return nb >= 2;
}
static void enqueue_lastblock_sym(calipso_hip_solver* s, hipStream_t stream) {
    if (!lastblock_sym_on(s)) return;
    const int NP = s->d.NP, tb = trsv_block(NP, (int)s->solve_block), nb = (NP + tb - 1) / tb;
    const int k0 = (nb - 1) * tb, w = NP - k0;
    const Batch bt = batch_of(s).b;
    hipLaunchKernelGGL(k_lastblock_sym, dim3((w / 32) * (w / 32), 1, bt.n), dim3(256), 0, stream, bt, tb, nb - 1, w, s->Tinv, s->Dx, s->Wfac + wform_offset(NP, tb, nb - 1));
}
// rows of W_kb that can be non-zero: all rb of them, or — banded S (structure.hip) — the rows the block's columns reach (the others are exact zeros: skipping
// them changes no bit, which is what keeps the banded treatment bitwise the dense one)
static int wform_rows(const calipso_hip_solver* s, int rb) {
    if (s->band64 <= 0) return rb;
    return std::min(rb, ((s->half_bandwidth + 63) / 64) * 64);
}
// the W-form products of solve block kb (after its factor columns and its inverse block are complete)
static void enqueue_wform(calipso_hip_solver* s, hipStream_t stream, int kb) {
    const int NP = s->d.NP, tb = trsv_block(NP, (int)s->solve_block);
    const int k0 = kb * tb, w = std::min(tb, NP - k0), rb = NP - k0 - w;
    if (rb <= 0) return;
    const Batch bt = batch_of(s).b;
    const int rows = wform_rows(s, rb);
    hipLaunchKernelGGL(k_wform_product, dim3((rows / 32) * (w / 32), 1, bt.n), dim3(256), 0, stream, bt, NP, tb, kb, w, rb, rows, s->Lf, s->Tinv, s->Wfac + wform_offset(NP, tb, kb));
}
static void enqueue_trsv_wform(calipso_hip_solver* s, double* x) {
    const int NP = s->d.NP, tb = trsv_block(NP, (int)s->solve_block), nb = (NP + tb - 1) / tb;
    double* u = s->zf;
    double* z = s->zf2;
    const Batch bt = batch_of(s).b;
    const unsigned nz = bt.n;
    const int* gate = s->gate_epoch ? s->gate : (const int*)nullptr;
    const int ge = s->gate_epoch;
    const bool sym = lastblock_sym_on(s);            // the last block as ONE launch (k_block_sym)
    for (int kb = 0; kb < nb - (sym ? 1 : 0); ++kb) {
        const int k0 = kb * tb, w = std::min(tb, NP - k0), rb = NP - k0 - w;
        const double* Wb = s->Wfac + wform_offset(NP, tb, kb);
        const int rows = wform_rows(s, rb);
        const int snap = (sym && kb == nb - 2) ? 1 + rows : 0;          // the block before the last one leaves a copy of the last block's final right-hand side in u
        const int grows = snap ? rb : rows;
        if (w > 512) hipLaunchKernelGGL((k_trsv_fwd<16, 32, 32>), dim3((w + grows) / 16, 1, nz), dim3(512), 0, s->stream, bt, kb, tb, w, rb, s->Tinv, Wb, x, s->Dx, u, z, gate, ge, snap);
        else hipLaunchKernelGGL((k_trsv_fwd<16, 16, 32>), dim3((w + grows) / 16, 1, nz), dim3(256), 0, s->stream, bt, kb, tb, w, rb, s->Tinv, Wb, x, s->Dx, u, z, gate, ge, snap);
    }
    if (sym) {
        const int k0 = (nb - 1) * tb, w = NP - k0;
        const double* Msym = s->Wfac + wform_offset(NP, tb, nb - 1);
        if (w > 512) hipLaunchKernelGGL((k_block_sym<16, 32, 32>), dim3(w / 16, 1, nz), dim3(512), 0, s->stream, bt, k0, w, Msym, u, x, gate, ge);
        else hipLaunchKernelGGL((k_block_sym<16, 16, 32>), dim3(w / 16, 1, nz), dim3(256), 0, s->stream, bt, k0, w, Msym, u, x, gate, ge);
    }
    for (int kb = nb - 1 - (sym ? 1 : 0); kb >= 0; --kb) {
        const int k0 = kb * tb, w = std::min(tb, NP - k0), rb = NP - k0 - w;
        const double* Wb = s->Wfac + wform_offset(NP, tb, kb);
        const int rows = wform_rows(s, rb), nch = (w + rows + 63) / 64;
        if (nch <= 8) launch_trsv_bwd<8>(s->stream, nz, bt, kb, tb, w, rb, rows, s->Tinv, Wb, z, x, gate, ge);
        else if (nch <= 16) launch_trsv_bwd<16>(s->stream, nz, bt, kb, tb, w, rb, rows, s->Tinv, Wb, z, x, gate, ge);
        else if (nch <= 24) launch_trsv_bwd<24>(s->stream, nz, bt, kb, tb, w, rb, rows, s->Tinv, Wb, z, x, gate, ge);
        else if (nch <= 32) launch_trsv_bwd<32>(s->stream, nz, bt, kb, tb, w, rb, rows, s->Tinv, Wb, z, x, gate, ge);
        else if (nch <= 40) launch_trsv_bwd<40>(s->stream, nz, bt, kb, tb, w, rb, rows, s->Tinv, Wb, z, x, gate, ge);
        else if (nch <= 48) launch_trsv_bwd<48>(s->stream, nz, bt, kb, tb, w, rb, rows, s->Tinv, Wb, z, x, gate, ge);
        else launch_trsv_bwd<64>(s->stream, nz, bt, kb, tb, w, rb, rows, s->Tinv, Wb, z, x, gate, ge);
    }
}

// x (length NP, padded entries zero) <- S^-1 x
static void enqueue_trsv(calipso_hip_solver* s, double* x) {
    if (wform_on(s)) { enqueue_trsv_wform(s, x); return; }
    const int NP = s->d.NP, tb = trsv_block(NP, (int)s->solve_block), nb = (NP + tb - 1) / tb;
    double* u = s->zf;         // forward result (unscaled), consumed by the updates
    double* z = s->zf2;        // D^-1 u, then overwritten block by block with v
    const Batch bt = batch_of(s).b;
    const unsigned nz = bt.n;
    for (int kb = 0; kb < nb; ++kb) {
        const int k0 = kb * tb, w = std::min(tb, NP - k0);
        int rest = NP - (k0 + w);
        if (s->band64 > 0) rest = std::min(rest, ((s->half_bandwidth + 31) / 32) * 32);      // rows below the block that its columns reach
        // 16 rows per workgroup (128-byte runs down a column), 32 columns per thread in one batch of loads: 64 workgroups per 1024-wide block.  Measured
        // at C3 (7 solves per step): 0.71 ms for the solve + refinement phase against 0.82 with 32 rows x 64 columns per thread (32 workgroups per block)
        // and 0.73 - 0.75 with 8 rows or 16 / 64 columns per thread.
        if (w > 512) {                                                                        // (a block with rows below it is tb = 512, 1024 or 2048 wide)
            hipLaunchKernelGGL((k_trsv_block_n<16, 32, 32>), dim3(w / 16, 1, nz), dim3(512), 0, s->stream, bt, kb, tb, w, s->Tinv, x, s->Dx, u, z);
            if (rest > 0) hipLaunchKernelGGL((k_trsv_update_n<16, 32, 32>), dim3(rest / 16, 1, nz), dim3(512), 0, s->stream, bt, NP, k0, w, s->Lf, u, x);
        } else {
            hipLaunchKernelGGL((k_trsv_block_n<16, 16, 32>), dim3(w / 16, 1, nz), dim3(256), 0, s->stream, bt, kb, tb, w, s->Tinv, x, s->Dx, u, z);
            if (rest > 0) hipLaunchKernelGGL((k_trsv_update_n<16, 16, 32>), dim3(rest / 16, 1, nz), dim3(256), 0, s->stream, bt, NP, k0, w, s->Lf, u, x);
        }
    }
    for (int kb = nb - 1; kb >= 0; --kb) {
        const int k0 = kb * tb, w = std::min(tb, NP - k0);
        if (w > 1024) hipLaunchKernelGGL(k_trsv_block_t<2048>, dim3(w / 4, 1, nz), dim3(256), 0, s->stream, bt, kb, tb, w, s->Tinv, z, x);
        else hipLaunchKernelGGL(k_trsv_block_t<1024>, dim3(w / 4, 1, nz), dim3(256), 0, s->stream, bt, kb, tb, w, s->Tinv, z, x);
        if (kb > 0) {
            const int cfirst = s->band64 > 0 ? std::max(0, ((k0 - s->half_bandwidth) / 4) * 4) : 0;   // columns left of the block that reach into it
            if (w > 1024) hipLaunchKernelGGL(k_trsv_update_t<2048>, dim3((k0 - cfirst) / 4, 1, nz), dim3(256), 0, s->stream, bt, NP, k0, w, cfirst, s->Lf, x, z);
            else hipLaunchKernelGGL(k_trsv_update_t<1024>, dim3((k0 - cfirst) / 4, 1, nz), dim3(256), 0, s->stream, bt, NP, k0, w, cfirst, s->Lf, x, z);
        }
    }
}

// The factorisation of S and the triangular solves are fixed kernel sequences with fixed arguments (118 and 18 launches):
// they are captured once per handle into hipGraphs and replayed, so the host issues one graph launch instead of queueing every
// kernel (the GPU otherwise waits on the host between the many few-microsecond kernels).

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
extern "C" int32_t calipso_hip_debug_ldl_bulk_trace(long long* out) { return hipMemcpyFromSymbol(out, HIP_SYMBOL(calipso::g_ldl_bulk), sizeof(long long) * (32 * 8 + 32 * 16 * 2)) == hipSuccess ? 0 : -1; }
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
// The right-hand side of the first condensed solve (k_residual_symmetric, then b_x += [gx; hx]'(Omega b_m)) needs the cone pivots, not the factor: do_factorize queues it
// on the second stream of a handle whose factorisation will use that stream (and join it before the solves: enqueue_ldl_finish), where it runs on the compute units
// k_schur leaves free instead of between the factorisation and the first solve.  nullptr: no such stream — the operands are formed where they always were.
hipStream_t ldl_rhs_stream(calipso_hip_solver* s) {
    static const bool env = [] { const char* e = getenv("CALIPSO_HIP_RHS_AHEAD"); return !e || atoi(e) != 0; }();
    if (!env || s->cur || s->compact || (s->stage_parallel && s->spS) || (s->use_graphs && graph_ldl_requested())) return nullptr;
    if (!ldl_overlap(s) || !side_stream(s)) return nullptr;
    return s->stream2;
}

void launch_ldl(calipso_hip_solver* s) {
    // state of the previous factorisation that do_factorize / enqueue_ldl_finish would otherwise act on: a blocked factorisation that published its inertia
    // counts followed by a stage-parallel one on the same handle must not leave do_factorize waiting for a sequence number that was consumed long ago
    s->ldl_forks = 0;
    s->ldl_pub_seq = 0;
    s->ldl_overlap_on = false;
    s->ldl_failed = false;
    if (s->stage_parallel && s->spS) {        // stage-parallel: multifrontal LDL^T of S over its nested-dissection tree (sparse.hip)
        const Batch bt = batch_of(s).b;
        if (sparse_factor_from_dense(s->spS, s->stream, bt, s->S, s->spS_src, s->icount, s->compact && s->spS_inv && s->spS_values_current) == CALIPSO_OK) { s->spS_values_current = false; (void)hipEventRecord(s->ev[14], s->stream); return; }
        if (s->compact) {
            s->err = "structured handle: the multifrontal factorisation was refused and there is no blocked one to fall back to";
            s->ldl_failed = true;             // (do_factorize turns it into CALIPSO_ERR_HIP: no stale inertia, no solve with an absent factor)
            (void)hipEventRecord(s->ev[14], s->stream);
            return;
        }
        s->stage_parallel = false;            // (a group larger than the reserved batch: back to the blocked factorisation)
        launch_pad_identity(s);               // launch_schur skipped the padding of S for the multifrontal path: the blocked one needs its unit pivots
    }
    s->Lf = lfac_factor_buffer(s);        // where this factorisation's factor columns go: S (scaled in place) or, under the left-looking schedule, lfac.hip's buffer
    // (a group launch covers a changing set of instances: its kernel arguments differ from call to call, so no graph there)
    // The panel steps are queued launch by launch (the host keeps ahead of a 20 us chain: 0.935 ms per factorisation at C3 against 0.950 as a captured graph), which
    // also lets the host hand the completed solve blocks to the second stream while the chain runs (below): 0.889 ms.  CALIPSO_HIP_GRAPH_LDL=1 brings the
    // graphs back (one stream, no overlap); the solves keep theirs (launch_trsv).
    const bool graphs = !s->cur && s->use_graphs && graph_ldl_requested();
    if (ldl_overlap(s)) (void)side_stream(s);       // (created outside a stream capture)
    static const bool pub_env = [] { const char* e = getenv("CALIPSO_HIP_LDL_PUBLISH"); return !e || atoi(e) != 0; }();     // (experiment switch)
    s->ldl_publish = pub_env && !graphs && !s->cur;       // (a captured launch would replay a stale sequence number)
    s->ldl_epoch += 1;
    if (!graphs || !replay_or_capture(s, s->graph_ldl, s->graph_ldl_tried, [&] { enqueue_ldl_steps(s); })) enqueue_ldl_steps(s);
    (void)hipEventRecord(s->ev[14], s->stream);
    // The host (which would only wait for the factorisation anyway) hands the completed solve blocks to the second stream: it watches the progress word
    // the panel steps store and queues a block's finish when the step after the block's last panel has started.  (A hipStreamWaitEvent on the second
    // stream instead was measured: a queue blocked on a barrier costs every dispatch of the chain's queue ~0.8 us — 30 us per factorisation.)
    for (int f = 0; f < s->ldl_forks; ++f) {
        const unsigned long long want = (s->ldl_epoch << 16) | (unsigned long long)s->ldl_feeds[3 * f];
        (void)host_wait([&] { return __atomic_load_n(s->hprog, __ATOMIC_ACQUIRE) >= want; }, [&] { return hipStreamQuery(s->stream) == hipErrorNotReady; });     // (not alive: the chain is through, or the queue faulted)
        enqueue_feed(s, s->stream2, f, true);
    }
    if (!graphs || !replay_or_capture(s, s->graph_ldl_fin, s->graph_ldl_fin_tried, [&] { enqueue_ldl_finish(s); })) enqueue_ldl_finish(s);
}

void launch_trsv_direct(calipso_hip_solver* s, double* x) {
    if (s->stage_parallel && s->spS) { (void)sparse_solve_inplace(s->spS, s->stream, batch_of(s).b, x); return; }
    enqueue_trsv(s, x);
}
void launch_trsv(calipso_hip_solver* s, double* x) {
    if (s->stage_parallel && s->spS) { (void)sparse_solve_inplace(s->spS, s->stream, batch_of(s).b, x); return; }
    if (s->cur || x != s->xbuf || !s->use_graphs || !replay_or_capture(s, s->graph_trsv, s->graph_trsv_tried, [&] { enqueue_trsv(s, s->xbuf); })) enqueue_trsv(s, x);
}

}  // namespace calipso
