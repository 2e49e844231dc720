// schur.hip — elimination of the constraint block of the condensed KKT matrix.
//
// The condensed matrix of residual_jacobian_variables.jl:110-167 is
//        K = [ Lxx + ep*I   gx'      hx'  ]     factored by the reference as P K P' = L D L' (qdldl.jl:400-589) in a
//            [ gx           k_y*I    0    ]     fill-reducing order.  We use the constraint-first order [z | y | x]:
//            [ hx           0        B_z  ]     the leading (ne+nc) x (ne+nc) block is (block-)diagonal, so its part of
// the LDL' is closed-form per cone and the trailing update of the whole x-block is ONE rank-(ne+nc) update
//        S = Lxx + ep*I + omega_y * gx'gx + hx' (Omega_z hx),   omega_y = -1/k_y,  Omega_z = -B_z^-1  (block diagonal)
// which is the only GEMM-shaped work of the assembly and runs on the fp64 matrix cores (v_mfma_f64_16x16x4_f64).
// The nx x nx remainder S is factored by ldl.hip.  Only triu(K) is used, as in the reference (qdldl.jl:145-147):
// second-order blocks B_z are symmetrised from their upper triangle and Lxx is read through its upper triangle.
#include "internal.hpp"
#include "device_utils.hpp"

#include <algorithm>
#include <mutex>

namespace calipso {

// ---- per-cone pivots / weights ----------------------------------------------------------------------------------------
// nonnegative entry i (residual_jacobian_variables.jl:142-149):  K_zz = -Sbar/(T + Sbar*P) + D,  Sbar = s-ed, T = t, P = ep, D = -ed
// second-order cone (:151-164):  B = -(Cs + Cbar_t P)^-1 Cbar_t + D  column by column with the closed-form arrow inverse
// equality rows (:131-133):      k_y = -1/(rho+ep) + (-ed)
// Also counts the signs of the pivots of this block (compute_inertia!, linear_solver.jl:33-44) into icount[0..2].
__global__ __launch_bounds__(1024) void k_cone_weights(BatchSc bt, Dims d, ConeDev cd, const double* __restrict__ w,
                                                        double* __restrict__ kzz, double* __restrict__ wz, double* __restrict__ Bsoc,
                                                        double* __restrict__ Wsoc, double* __restrict__ work, int* __restrict__ icount) {
    __shared__ int smi[3][16];
    inst_shift(bt.b, w, kzz, wz, Bsoc, Wsoc, work);
    inst_shift_i(bt.b, icount);
    const Scalars sc = bt.scal(blockIdx.z);
    const int tid = threadIdx.x;
    const double* sl = w + d.os();
    const double* t = w + d.ot();
    const double Hss = 0.0 + sc.ep;
    int pos = 0, nonpos = 0, zero = 0;
    for (int i = tid; i < d.q; i += 1024) {
        const double Sb = sl[i] - sc.ed, Ti = t[i];
        const double k = -1.0 * Sb / (Ti + Sb * Hss) + (0.0 - sc.ed);
        kzz[i] = k;
        wz[i] = -1.0 / k;
        pos += k > 0.0; nonpos += k <= 0.0; zero += k == 0.0;
    }
    for (int j = tid; j < d.n_soc; j += 1024) {
        const int st = cd.soc_start[j], dim = cd.soc_dim[j], off = cd.soc_woff[j];
        double* B = Bsoc + off;
        double* W = Wsoc + off;
        if (dim > 4) continue;
        {
            // small cones: the d x d blocks live in registers (loops unrolled to constant indices); the operations and their order are those of the
            // sequential formulation (oracle: oracle_residual_jacobian_variables_symmetric)
            constexpr int MD = 4;
            double ls[MD], lt[MD], u[MD], c[MD], o[MD], Bm[MD * MD], M[MD * MD];
#pragma unroll
            for (int a = 0; a < MD; ++a) { ls[a] = a < dim ? sl[st + a] : 0.0; lt[a] = a < dim ? t[st + a] : 0.0; u[a] = 0.0; c[a] = 0.0; o[a] = 0.0; }
#pragma unroll
            for (int e = 0; e < MD * MD; ++e) { Bm[e] = 0.0; M[e] = 0.0; }
            const double sb1 = ls[0] - sc.ed;
            u[0] = lt[0] + sb1 * Hss;
#pragma unroll
            for (int k = 1; k < MD; ++k) if (k < dim) u[k] = lt[k] + ls[k] * Hss;
#pragma unroll
            for (int col = 0; col < MD; ++col) if (col < dim) {
#pragma unroll
                for (int a = 0; a < MD; ++a) if (a < dim) c[a] = (a == col) ? sb1 : (col == 0 ? ls[a] : (a == 0 ? ls[col] : 0.0));
                arrow_inverse_small<MD>(dim, u, c, o);
#pragma unroll
                for (int a = 0; a < MD; ++a) if (a < dim) Bm[a + col * MD] = 0.0 - o[a];
            }
#pragma unroll
            for (int a = 0; a < MD; ++a) if (a < dim) Bm[a + a * MD] += (0.0 - sc.ed);
#pragma unroll
            for (int b = 0; b < MD; ++b)
#pragma unroll
                for (int a = 0; a < MD; ++a) if (a < dim && b < dim) { B[a + b * dim] = Bm[a + b * MD]; M[a + b * MD] = (a <= b) ? Bm[a + b * MD] : Bm[b + a * MD]; }
#pragma unroll
            for (int jj = 0; jj < MD; ++jj) if (jj < dim) {
                const double dj = M[jj + jj * MD];
                pos += dj > 0.0; nonpos += dj <= 0.0; zero += dj == 0.0;
#pragma unroll
                for (int i = jj + 1; i < MD; ++i) if (i < dim) {
                    const double yij = M[i + jj * MD];
                    const double l = yij / dj;
#pragma unroll
                    for (int k = jj + 1; k < MD; ++k) if (k <= i) M[i + k * MD] -= l * (k == i ? yij : M[jj + k * MD]);
                    M[i + jj * MD] = l;
                    M[jj + i * MD] = yij;
                }
            }
#pragma unroll
            for (int col = 0; col < MD; ++col) if (col < dim) {
#pragma unroll
                for (int a = 0; a < MD; ++a) o[a] = (a == col) ? 1.0 : 0.0;
#pragma unroll
                for (int a = 0; a < MD; ++a) if (a < dim) {
                    double x = o[a];
#pragma unroll
                    for (int k = 0; k < MD; ++k) if (k < a) x -= M[a + k * MD] * o[k];
                    o[a] = x;
                }
#pragma unroll
                for (int a = 0; a < MD; ++a) if (a < dim) o[a] /= M[a + a * MD];
#pragma unroll
                for (int a = MD - 1; a >= 0; --a) if (a < dim) {
                    double x = o[a];
#pragma unroll
                    for (int k = 0; k < MD; ++k) if (k > a && k < dim) x -= M[k + a * MD] * o[k];
                    o[a] = x;
                }
#pragma unroll
                for (int a = 0; a < MD; ++a) if (a < dim) W[a + col * dim] = -o[a];
            }
        }
        // (cones of dimension > 4 are the business of k_cone_weights_wide, soc_wide.hip: one wavefront per cone)
    }
    if (tid == 0 && d.ne > 0) {
        const double ky = -1.0 / (sc.rho + sc.ep) + (0.0 - sc.ed);
        if (ky > 0.0) pos += d.ne;
        if (ky <= 0.0) nonpos += d.ne;
        if (ky == 0.0) zero += d.ne;
    }
    pos = wave_sum_i(pos); nonpos = wave_sum_i(nonpos); zero = wave_sum_i(zero);
    if ((tid & 63) == 0) { smi[0][tid >> 6] = pos; smi[1][tid >> 6] = nonpos; smi[2][tid >> 6] = zero; }
    __syncthreads();
    if (tid < 3) {
        int a = 0;
        for (int k = 0; k < 16; ++k) a += smi[tid][k];
        icount[tid] = a;
        icount[3 + tid] = 0;   // reset the counters of the S part (ldl.hip accumulates into them)
    }
}

void launch_cone_weights(calipso_hip_solver* s) {
    const BatchSc B = batch_of(s);
    hipLaunchKernelGGL(k_cone_weights, dim3(1, 1, B.b.n), dim3(1024), 0, s->stream, B, s->d, s->cone, s->solution, s->kzz, s->wz, s->Bsoc,
                       s->Wsoc, s->socwork, s->icount);
    launch_cone_weights_wide(s);          // cones of dimension > 4 (adds their pivot signs to the counts the kernel above has just written)
}

// WH = Omega_z * hx  (nc x nx): nonnegative rows scaled by -1/K_zz, second-order rows multiplied by the d x d block W
constexpr int SCALE_COLS = 16;
__global__ void k_scale_rows(Batch bt, Dims d, ConeDev cd, const double* __restrict__ hx, const double* __restrict__ wz,
                             const double* __restrict__ Wsoc, double* __restrict__ WH, const int* __restrict__ zrow, double* __restrict__ Spad) {
    inst_shift(bt, hx, wz, Wsoc, WH);
    if (zrow) inst_shift_i(bt, zrow);
    if (Spad) {
        // k_pad_identity rides along (the launch in front of k_schur): the threads of the instance share the (NP - nx) x NP padded rows of S
        inst_shift(bt, Spad);
        const long long total = (long long)gridDim.x * gridDim.y * blockDim.x, count = (long long)(d.NP - d.nx) * d.NP;
        for (long long e = ((long long)blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; e < count; e += total) {
            const int i = d.nx + (int)(e / d.NP), j = (int)(e % d.NP);
            if (j <= i) Spad[i + (size_t)j * d.NP] = (i == j) ? 1.0 : 0.0;
        }
    }
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= d.nc) return;
    int st = 0, dim = 0;
    const double* W = nullptr;
    double w0 = 0.0;
    if (c < d.q) w0 = wz[c];
    else { const int j = cd.entry_soc[c]; st = cd.soc_start[j]; dim = cd.soc_dim[j]; W = Wsoc + cd.soc_woff[j]; }
    int col0 = blockIdx.y * SCALE_COLS, col1 = min(d.nx, col0 + SCALE_COLS);   // a block of columns per workgroup: the cone data is looked up once
    if (zrow) {                          // analysed structure (structure.hip): the row is zero outside its column range — in hx, hence in WH (never written there)
        const int k = d.ne + c;
        col0 = max(col0, zrow[2 * k]); col1 = min(col1, zrow[2 * k + 1]);
    }
    // the columns of a nonnegative row are independent loads: all SCALE_COLS of them are in flight at once (the loop written with its bounds as a predicate so
    // that it unrolls; one column at a time left this kernel at 2 TB/s)
    const int cbase = blockIdx.y * SCALE_COLS;
    if (c < d.q) {
        double hv[SCALE_COLS];
#pragma unroll
        for (int k = 0; k < SCALE_COLS; ++k) { const int col = cbase + k; hv[k] = (col >= col0 && col < col1) ? hx[c + (size_t)col * d.m] : 0.0; }
#pragma unroll
        for (int k = 0; k < SCALE_COLS; ++k) { const int col = cbase + k; if (col >= col0 && col < col1) WH[c + (size_t)col * d.nc] = w0 * hv[k]; }
        return;
    }
    if (dim <= 4) {
        // small cones: the row of W once, then eight columns at a time with their (up to) four entries of hx in flight together; the sum in the order b = 0, 1, ...
        double wr[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) wr[b] = b < dim ? W[(c - st) + b * dim] : 0.0;
        for (int k0 = 0; k0 < SCALE_COLS; k0 += 8) {
            double hv[8][4];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int col = cbase + k0 + k;
                const bool in = col >= col0 && col < col1;
#pragma unroll
                for (int b = 0; b < 4; ++b) hv[k][b] = (in && b < dim) ? hx[st + b + (size_t)col * d.m] : 0.0;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int col = cbase + k0 + k;
                if (col >= col0 && col < col1) {
                    double v = 0.0;
#pragma unroll
                    for (int b = 0; b < 4; ++b) if (b < dim) v += wr[b] * hv[k][b];
                    WH[c + (size_t)col * d.nc] = v;
                }
            }
        }
        return;
    }
    for (int col = col0; col < col1; ++col) {
        const double* h = hx + (size_t)col * d.m;     // hx is the lower part of the stacked Jacobian (ld = m)
        double v = 0.0;
        for (int b = 0; b < dim; ++b) v += W[(c - st) + b * dim] * h[st + b];
        WH[c + (size_t)col * d.nc] = v;
    }
}

void launch_scale_rows(calipso_hip_solver* s) {
    s->pad_done = false;
    if (s->d.nc == 0) return;
    if (s->blocks.on && s->blocks_effective) return;      // stage blocks: Omega is applied while k_schur_blocks stages its operand, WH is not formed
    const BatchSc B = batch_of(s);
    // the unit pivots of the padded rows of S are written by this launch when the dense k_schur follows (launch_schur then skips launch_pad_identity)
    const bool pad = s->d.NP > s->d.nx && !(s->stage_parallel && s->spS);
    s->pad_done = pad;
    hipLaunchKernelGGL(k_scale_rows, dim3((s->d.nc + 255) / 256, (s->d.nx + SCALE_COLS - 1) / SCALE_COLS, B.b.n), dim3(256), 0, s->stream, B.b, s->d, s->cone, s->hx, s->wz, s->Wsoc, s->WH, s->band64 > 0 ? s->zrow : nullptr,
                       pad ? s->S : (double*)nullptr);
}

// ---- Schur complement on the fp64 matrix cores -----------------------------------------------------------------------------
// Workgroup = 256 threads = 4 wavefronts (2 x 2), tile 128 x 128 of the lower triangle of S; each wavefront owns a
// 64 x 64 sub-tile = 4 x 4 MFMA tiles of 16 x 16 (64 accumulator doubles per lane).  The K dimension runs over the
// ne equality rows, then the nc cone rows, KT rows per LDS stage.  Operand tiles are read from HBM with lanes along k
// (the contiguous dimension of the column-major Jacobians), stored k-fastest in LDS with a +2 pad so that the
// ds_read_b64 of an MFMA fragment (16 rows x 4 k) touches 64 distinct banks.
//   v_mfma_f64_16x16x4_f64:  A: lane l holds A[i = l&15][k = l>>4];  B: lane l holds B[k = l>>4][j = l&15];
//                            D: lane l, register r holds D[row = (l>>4) + 4r][col = l&15].
// blockIdx -> tile is XCD-aware: the 8 XCDs each get a contiguous band of tile rows, so the operand columns a band
// needs are shared through that XCD's L2 instead of being fetched by all eight.
// Workgroup = 1024 threads = 16 wavefronts.  Tile = 128 rows x TJ columns of the lower triangle, TJ = 16 nj <= 128.  Wavefront w owns
// row tile w & 7 and the four column tiles 4 (w >> 3) .. + 3: EVERY wavefront runs the same branch-free instruction stream
// (5 ds_read_b64 + 4 MFMAs per k-step); column tiles beyond the tile width are zero-filled in LDS and dropped in the epilogue.
// (Round 1 dealt the column tiles by `if (n < jcnt)`: wave-uniform, but the compiler turned every MFMA into its own exec-masked basic
// block with an s_waitcnt lgkmcnt(0) in front, which serialised LDS reads and matrix instructions — 36 TFLOP/s; the stage loop of this
// kernel in isolation, bench/schur_loop_bench.hip, sustains 61.)  Global addresses are a wave-uniform base (scalar registers,
// advanced per stage by scalar adds) plus a per-thread 32-bit offset that is constant over the stages, and the bounds predicates are
// only evaluated for the stages / tiles that touch an edge: the vector ALU work per stage is a handful of instructions.
// fp64 matrix-core ceiling: bench/mfma444_loop.hip — 75 TFLOP/s for back-to-back v_mfma_f64_16x16x4_f64 with distinct operands
// (the 47-49 of bench/mfma_f64_peak.hip comes from issuing the SAME source registers back to back), 76 for the 4x4x4 form.
constexpr int KT = 32;
constexpr int LDK = KT + 2;
constexpr int SCHUR_THREADS = 1024;
constexpr int SLD = TILE * KT / SCHUR_THREADS;   // doubles per thread per operand per stage
typedef double v4d __attribute__((ext_vector_type(4)));

struct SchurStage { const double* MA; const double* MB; int kmax; int k0; double bscale; int lda, ldb; };

// stage st of a tile: the first nst0 stages walk the equality rows from stage est0 on, the rest the cone rows from stage cst0 on
// (est0 = cst0 = 0 and all stages for a dense problem; structure.hip narrows the ranges for stage-banded ones)
__device__ __forceinline__ SchurStage schur_stage(int st, int nst0, int est0, int cst0, const Dims& d, const double* gx, const double* hx, const double* WH,
                                                  double omega_y) {
    SchurStage g;
    if (st < nst0) { g.MA = gx; g.MB = gx; g.kmax = d.ne; g.k0 = (est0 + st) * KT; g.bscale = omega_y; g.lda = d.m; g.ldb = d.m; }
    else { g.MA = hx; g.MB = WH; g.kmax = d.nc; g.k0 = (cst0 + st - nst0) * KT; g.bscale = 1.0; g.lda = d.m; g.ldb = d.nc; }
    return g;
}

// global -> registers: KT x width block of M (rows k0.., columns c0..c0+width), lanes along k; out-of-range -> 0
__device__ __forceinline__ void stage_fetch(double (&r)[SLD], const double* __restrict__ M, int ldm, int kmax, int ncols, int k0, int c0, int width,
                                            int tid, double scale) {
    const int k = tid % KT, cbase = tid / KT;
#pragma unroll
    for (int it = 0; it < SLD; ++it) {
        const int c = cbase + it * (SCHUR_THREADS / KT);
        const int gk = k0 + k, gc = c0 + c;
        r[it] = (gk < kmax && gc < ncols && c < width) ? scale * M[gk + (size_t)gc * ldm] : 0.0;
    }
}
// the same block when it lies entirely inside the matrix: base = M + k0 + c0 * ldm is wave-uniform, off[it] (in doubles) is the thread's
// constant offset of its it-th element
__device__ __forceinline__ void stage_fetch_full(double (&r)[SLD], const double* __restrict__ base, const unsigned (&off)[SLD], double scale) {
#pragma unroll
    for (int it = 0; it < SLD; ++it) r[it] = scale * base[off[it]];
}
// registers -> LDS, k fastest: dst[c][k]
__device__ __forceinline__ void stage_store(double* dst, const double (&r)[SLD], int tid) {
    const int k = tid % KT, cbase = tid / KT;
#pragma unroll
    for (int it = 0; it < SLD; ++it) dst[(cbase + it * (SCHUR_THREADS / KT)) * LDK + k] = r[it];
}

// tile t -> (row block, column block).  The tiles that intersect { j <= i < nx } are enumerated in SB x SB super-blocks
// (super-rows outermost, then super-columns, row-major inside): the ~27 consecutive tiles an XCD works on at a time then span
// about SB row panels and SB column panels of the operands instead of 1 + 27, which is what its 4 MB L2 can share.
constexpr int SB = 5;
__device__ __forceinline__ int schur_row_count(int bi, int nx, int TJ) { return (min(nx, (bi + 1) * TILE) - 1) / TJ + 1; }
__device__ __forceinline__ void schur_tile(int t, int nx, int TJ, int& bi, int& bj) {
    const int nbi = (nx + TILE - 1) / TILE;
    for (int r0 = 0; r0 < nbi; r0 += SB) {
        const int r1 = min(nbi, r0 + SB);
        const int cmax = schur_row_count(r1 - 1, nx, TJ);          // widest row of this super-row
        for (int c0 = 0; c0 < cmax; c0 += SB) {
            for (int b = r0; b < r1; ++b) {
                const int c = min(max(schur_row_count(b, nx, TJ) - c0, 0), SB);
                if (t < c) { bi = b; bj = c0 + t; return; }
                t -= c;
            }
        }
    }
    bi = 0; bj = 0;   // not reached for t < ntiles
}

// banded S (structure.hip): row-major over the tiles that intersect { j <= i < nx, i - j <= hb }
__device__ __forceinline__ int schur_band_first(int bi, int TJ, int hb) {     // first column block of row block bi inside the band
    const int v = bi * TILE - hb - (TJ - 1);
    return v <= 0 ? 0 : (v + TJ - 1) / TJ;
}
__device__ __forceinline__ void schur_tile_banded(int t, int nx, int TJ, int hb, int& bi, int& bj) {
    const int nbi = (nx + TILE - 1) / TILE;
    for (bi = 0; bi < nbi; ++bi) {
        const int first = schur_band_first(bi, TJ, hb), c = schur_row_count(bi, nx, TJ) - first;
        if (t < c) { bj = first + t; return; }
        t -= c;
    }
    bi = 0; bj = 0;
}
// constraint rows a range of columns [c0, c1) visits: union of the per-group ranges of structure.hip (kr: 4 ints per 16 columns)
__device__ __forceinline__ void schur_rows(const int* __restrict__ kr, int c0, int c1, int which, int& lo, int& hi) {
    lo = 1 << 30; hi = 0;
    for (int g = c0 / 16; g <= (c1 - 1) / 16; ++g) { lo = min(lo, kr[4 * g + 2 * which]); hi = max(hi, kr[4 * g + 2 * which + 1]); }
}

// LDS: two stages of (A tile, B tile), 4 x 128 x LDK doubles = 136 KiB (dynamic).  Stage s+1 is written to the other buffer
// while the matrix cores work on stage s (its operands were fetched to registers one stage earlier), so a stage costs ONE barrier
// and neither the global-load nor the LDS-store latency is exposed.
constexpr size_t SCHUR_LDS_BYTES = 4 * (size_t)TILE * LDK * sizeof(double);
__global__ __launch_bounds__(SCHUR_THREADS) void k_schur(BatchSc bt, Dims d, const double* __restrict__ Lsym, const double* __restrict__ gx,
                                                          const double* __restrict__ hx, const double* __restrict__ WH, double* __restrict__ S,
                                                          const int* __restrict__ kr, int hb, int ntiles, int nj, int flat) {
    extern __shared__ __attribute__((aligned(16))) double schur_lds[];
    // XCD-aware remap over ONE flattened grid: block b runs on XCD b % 8 (observed dispatch order; used for speed only) and every XCD owns a
    // contiguous eighth of the instance-major, tile-row-major list of (instance, tile) pairs: the tiles an XCD works on at a time belong to one
    // instance and to neighbouring tile rows, so the operand columns they need are shared through that XCD's L2 (with blockIdx.z = instance
    // every XCD saw the operands of all the instances in flight).
    int z, t;
    if (flat) {
        const long long G = (long long)bt.b.n * ntiles;
        const int xk = blockIdx.x & 7;
        const long long item = (long long)xk * G / 8 + (blockIdx.x >> 3);
        if (item >= (long long)(xk + 1) * G / 8) return;
        z = (int)(item / ntiles); t = (int)(item % ntiles);
    } else {                      // one grid row per instance (blockIdx.z): every XCD gets a contiguous band of each instance's tile list
        const int chunk = (gridDim.x + 7) / 8;
        z = blockIdx.z; t = (blockIdx.x % 8) * chunk + blockIdx.x / 8;
        if (t >= ntiles) return;
    }
    {
        const long long o = bt.b.delta[z];
        Lsym += o; gx += o; hx += o; WH += o; S += o; kr += 2 * o;
    }
    const Scalars sc = bt.scal(z);
    const int TJ = 16 * nj;
    int bi, bj;
    if (hb > 0) schur_tile_banded(t, d.nx, TJ, hb, bi, bj);
    else schur_tile(t, d.nx, TJ, bi, bj);
    const int i0 = bi * TILE, j0 = bj * TJ;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave & 7;                          // row tile (16 rows) of this wavefront
    const int jt0 = (wave >> 3) * 4;                  // its column tiles jt0 .. jt0 + 3 (those >= nj hold zeros and are dropped at the end)
    const int fr = lane & 15, fk = lane >> 4;

    // acc[n]: MFMA row index <-> column j of S, MFMA column index (the 16-lane fast index) <-> row i of S, so that the
    // epilogue's stores are 128-byte contiguous runs of the column-major S
    v4d acc[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[n] = (v4d){0.0, 0.0, 0.0, 0.0};

    const double omega_y = -1.0 / (-1.0 / (sc.rho + sc.ep) + (0.0 - sc.ed));
    // constraint rows that touch both the tile's rows (as columns of the Jacobians) and its columns
    int est0, nst0, cst0, nst;
    {
        int la, ha, lb, hbb;
        const int i1 = min(d.nx, i0 + TILE), j1 = min(d.nx, j0 + TJ);
        schur_rows(kr, i0, i1, 0, la, ha); schur_rows(kr, j0, j1, 0, lb, hbb);
        int lo = max(la, lb), hi = min(ha, hbb);
        est0 = lo / KT; nst0 = hi > lo ? (hi + KT - 1) / KT - est0 : 0;
        schur_rows(kr, i0, i1, 1, la, ha); schur_rows(kr, j0, j1, 1, lb, hbb);
        lo = max(la, lb); hi = min(ha, hbb);
        cst0 = lo / KT; nst = nst0 + (hi > lo ? (hi + KT - 1) / KT - cst0 : 0);
    }
    double ra[SLD], rb[SLD];
    // per-thread offsets (doubles) of its SLD elements inside a stage block, for the two leading dimensions in use (m for gx / hx, nc for WH)
    unsigned offm[SLD], offc[SLD];
    {
        const int k = tid % KT, cbase = tid / KT;
#pragma unroll
        for (int it = 0; it < SLD; ++it) {
            const unsigned c = (unsigned)(cbase + it * (SCHUR_THREADS / KT));
            offm[it] = (unsigned)k + c * (unsigned)d.m;
            offc[it] = (unsigned)k + c * (unsigned)d.nc;
        }
    }
    const bool rows_full = i0 + TILE <= d.nx, cols_full = (TJ == TILE) && j0 + TJ <= d.nx;      // the tile's operand columns all exist
    auto fetch = [&](int st) {
        const SchurStage g = schur_stage(st, nst0, est0, cst0, d, gx, hx, WH, omega_y);
        const bool kfull = g.k0 + KT <= g.kmax;                                                  // wave-uniform
        if (kfull && rows_full) stage_fetch_full(ra, g.MA + g.k0 + (size_t)i0 * g.lda, offm, 1.0);
        else stage_fetch(ra, g.MA, g.lda, g.kmax, d.nx, g.k0, i0, TILE, tid, 1.0);
        if (kfull && cols_full) {
            if (g.ldb == d.m) stage_fetch_full(rb, g.MB + g.k0 + (size_t)j0 * g.ldb, offm, g.bscale);
            else stage_fetch_full(rb, g.MB + g.k0 + (size_t)j0 * g.ldb, offc, g.bscale);
        } else stage_fetch(rb, g.MB, g.ldb, g.kmax, d.nx, g.k0, j0, TJ, tid, g.bscale);
    };
    if (nst > 0) {
        fetch(0);
        stage_store(schur_lds, ra, tid);
        stage_store(schur_lds + TILE * LDK, rb, tid);
        if (nst > 1) fetch(1);
    }
    __syncthreads();
#pragma unroll 1
    for (int st = 0; st < nst; ++st) {
        const double* As = schur_lds + (size_t)(st & 1) * 2 * TILE * LDK;
        const double* Bs = As + TILE * LDK;
        if (st + 1 < nst) {   // registers hold stage st+1: park it in the other buffer (free since the barrier that ended stage st-1)
            double* An = schur_lds + (size_t)((st + 1) & 1) * 2 * TILE * LDK;
            stage_store(An, ra, tid);
            stage_store(An + TILE * LDK, rb, tid);
        }
        if (st + 2 < nst) fetch(st + 2);
        {
            // operand fragments by explicit ds_read_b64: the compiler pairs plain loads of consecutive k-steps into ds_read2_b64, which
            // is banked modulo 32 and conflicts on this layout (45 % of the LDS cycles were conflict replays, rocprofv3 SQ_LDS_BANK_CONFLICT);
            // ds_read_b64 is conflict-free here.  The loads run one k-step ahead of the matrix instructions; lgkmcnt is tracked by hand
            // (LDS returns in order) and the wait takes the fragments as in/out operands so that their uses cannot move above it.
            const unsigned ab = (unsigned)(uintptr_t)(As + (wi * 16 + fr) * LDK + fk);
            const unsigned bb = (unsigned)(uintptr_t)(Bs + (jt0 * 16 + fr) * LDK + fk);
            double fa[2], fb[2][4];
            auto issue = [&](int kk, int buf) {
                const unsigned ao = ab + kk * 32u, bo = bb + kk * 32u;
                asm volatile("ds_read_b64 %0, %1" : "=v"(fa[buf]) : "v"(ao) : "memory");
                asm volatile("ds_read_b64 %0, %1" : "=v"(fb[buf][0]) : "v"(bo) : "memory");
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(fb[buf][1]) : "v"(bo), "n"(16 * LDK * 8) : "memory");
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(fb[buf][2]) : "v"(bo), "n"(32 * LDK * 8) : "memory");
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(fb[buf][3]) : "v"(bo), "n"(48 * LDK * 8) : "memory");
            };
            issue(0, 0);
#pragma unroll
            for (int kk = 0; kk < KT / 4; ++kk) {
                const int cur = kk & 1;
                if (kk + 1 < KT / 4) {
                    issue(kk + 1, cur ^ 1);
                    asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(fa[cur]), "+v"(fb[cur][0]), "+v"(fb[cur][1]), "+v"(fb[cur][2]), "+v"(fb[cur][3]) :: "memory");
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[cur]), "+v"(fb[cur][0]), "+v"(fb[cur][1]), "+v"(fb[cur][2]), "+v"(fb[cur][3]) :: "memory");
                }
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[cur][n], fa[cur], acc[n], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    // epilogue: + Lxx (through its upper triangle, as triu(K)) + ep on the diagonal; identity in the padding
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        if (jt0 + n >= nj) break;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gj = j0 + (jt0 + n) * 16 + fk + 4 * r;   // MFMA row
            const int gi = i0 + wi * 16 + fr;                  // MFMA column: contiguous rows of S
            if (gi >= d.NP || gj >= d.NP) continue;
            double v;
            if (gi < d.nx && gj < d.nx) {
                v = acc[n][r] + Lsym[gi + (size_t)gj * d.nx];   // = Lxx[min, max]: triu(K) mirrored (k_symmetrize_upper)
                if (gi == gj) v += sc.ep;
            } else {
                v = (gi == gj) ? 1.0 : 0.0;
            }
            S[gi + (size_t)gj * d.NP] = v;
        }
    }
}

// identity in the padded rows [nx, NP) of the lower triangle (the tiles only cover what intersects the real triangle)
__global__ void k_pad_identity(Batch bt, Dims d, double* __restrict__ S) {
    inst_shift(bt, S);
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = d.nx + blockIdx.y;
    if (i < d.NP && j <= i) S[i + (size_t)j * d.NP] = (i == j) ? 1.0 : 0.0;
}

// Lsym[i,j] = Lxx[min(i,j), max(i,j)]: the Hessian as a triu-only factorisation sees it (qdldl.jl:145-147), laid out so the
// Schur epilogue can read its lower triangle with unit stride.  32 x 32 tiles transposed through LDS.
__global__ __launch_bounds__(256) void k_symmetrize_upper(Batch bt, int nx, const double* __restrict__ Lxx, double* __restrict__ Lsym) {
    __shared__ double tile[32][33];
    inst_shift(bt, Lxx, Lsym);
    const int bi = blockIdx.x, bj = blockIdx.y;   // tile (rows bi, cols bj) of Lsym
    if (bi < bj) {                                // strictly upper tile: plain copy
        for (int c = threadIdx.y; c < 32; c += 8) {
            const int i = bi * 32 + threadIdx.x, j = bj * 32 + c;
            if (i < nx && j < nx) Lsym[i + (size_t)j * nx] = Lxx[i + (size_t)j * nx];
        }
        return;
    }
    // lower (or diagonal) tile: read the mirrored upper tile (rows bj, cols bi) and transpose
    for (int c = threadIdx.y; c < 32; c += 8) {
        const int r = bj * 32 + threadIdx.x, cc = bi * 32 + c;
        tile[c][threadIdx.x] = (r < nx && cc < nx) ? Lxx[r + (size_t)cc * nx] : 0.0;
    }
    __syncthreads();
    for (int c = threadIdx.y; c < 32; c += 8) {
        const int i = bi * 32 + threadIdx.x, j = bj * 32 + c;
        if (i < nx && j < nx) {
            // Lsym[i][j] with i in tile rows bi, j in tile cols bj: = Lxx[j][i] when i > j, = Lxx[i][j] otherwise
            double v = tile[threadIdx.x][c];      // = Lxx[row j][col i]
            if (bi == bj && i < j) v = Lxx[i + (size_t)j * nx];
            Lsym[i + (size_t)j * nx] = v;
        }
    }
}

void launch_symmetrize(calipso_hip_solver* s) {
    if (s->blocks.on) return;             // stage blocks: the slab region of Lsym holds the packed blocks; k_schur_blocks mirrors the upper triangle itself
    const int nt = (s->d.nx + 31) / 32;
    const BatchSc B = batch_of(s);
    hipLaunchKernelGGL(k_symmetrize_upper, dim3(nt, nt, B.b.n), dim3(32, 8), 0, s->stream, B.b, s->d.nx, s->Lxx, s->Lsym);
}

// host: tile shape for a launch that covers `instances` problem instances.  Tile = 128 x 16 nj.  Every wavefront issues the same
// instruction stream whatever nj is, so a tile costs the same for every nj > 4 (two column-tile groups) and half for nj <= 4 (one group
// does nothing useful but the other still paces the workgroup): the cost of a launch is rounds over the 256 CUs x that.
static int schur_tiles(int nx, int nj, int hb) {
    const int TJ = 16 * nj, nbi = (nx + TILE - 1) / TILE;
    int cnt = 0;
    for (int bi = 0; bi < nbi; ++bi) {
        const int v = bi * TILE - hb - (TJ - 1);
        const int first = (hb <= 0 || v <= 0) ? 0 : (v + TJ - 1) / TJ;
        cnt += (std::min(nx, (bi + 1) * TILE) - 1) / TJ + 1 - first;
    }
    return cnt;
}
static int schur_choose(int nx, int instances, int hb) {
    long best_cost = -1; int best = 8;
    for (int nj = 4; nj <= 8; ++nj) {
        const long cnt = (long)schur_tiles(nx, nj, hb) * instances;
        const long cost = ((cnt + 255) / 256) * 8;
        if (best_cost < 0 || cost <= best_cost) { best_cost = cost; best = nj; }   // ties: the larger tile
    }
    return best;
}
void schur_plan(calipso_hip_solver* s) { s->schur_nj = schur_choose(s->d.nx, 1, 0); }

// unit pivots in the padded rows of S (what the blocked LDL^T of ldl.hip factors beyond row nx)
void launch_pad_identity(calipso_hip_solver* s) {
    if (s->d.NP <= s->d.nx) return;
    const BatchSc B = batch_of(s);
    hipLaunchKernelGGL(k_pad_identity, dim3((s->d.NP + 255) / 256, s->d.NP - s->d.nx, B.b.n), dim3(256), 0, s->stream, B.b, s->d, s->S);
}

void launch_schur(calipso_hip_solver* s) {
    (void)lds_attribute((const void*)k_schur, (int)SCHUR_LDS_BYTES);      // (per device; several host lanes may arrive concurrently)
    s->spS_values_current = false;        // (only a k_schur_blocks launch of THIS factorisation may have written the multifrontal values: blocks_schur sets it)
    if (blocks_schur(s)) return;          // stage blocks: S by segment pairs from the packed blocks (blocks.hip)
    if (s->hessian_dirty && !s->cur) { launch_symmetrize(s); s->hessian_dirty = false; }   // (a group refreshes its members itself)
    if (lfac_ready(s)) {                  // one dense system alone: the products are slices of the panel launches (lfac.hip), queued by launch_ldl
        if (!s->pad_done) launch_pad_identity(s);
        s->pad_done = false;
        return;
    }
    const BatchSc B = batch_of(s);
    const int hb = s->band64 > 0 ? s->half_bandwidth : 0;       // > 0: only the tiles inside the band (structure.hip)
    const int nj = (B.b.n == 1 && hb == 0) ? s->schur_nj : schur_choose(s->d.nx, B.b.n, hb);
    const int ntiles = schur_tiles(s->d.nx, nj, hb);
    // every XCD gets ceil(n ntiles / 8) workgroups: enough for its share [k G / 8, (k + 1) G / 8) of the list
    const int flat = 1;      // (one flattened grid over all instances; the per-instance grid rows of round 2 remain in the kernel for z-launches)
    const int grid = flat ? (int)(((long long)B.b.n * ntiles + 7) / 8 + 1) * 8 : ((ntiles + 7) / 8) * 8;
    if (!(s->stage_parallel && s->spS) && !s->pad_done) launch_pad_identity(s);   // (the multifrontal path reads S only inside its nx x nx pattern; k_scale_rows may have written the padding already)
    s->pad_done = false;
    hipLaunchKernelGGL(k_schur, dim3(grid, 1, flat ? 1 : B.b.n), dim3(SCHUR_THREADS), SCHUR_LDS_BYTES, s->stream, B, s->d, s->Lsym, s->gx, s->hx, s->WH, s->S, s->krange, hb, ntiles, nj, flat);
}

}  // namespace calipso
