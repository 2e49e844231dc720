// gemv.hip — HBM-bound mat-vecs with the dense Hessian / Jacobian blocks (column-major, fp64).
//   gemv_t: y = alpha * A' x + beta*y   one wavefront per column, lanes stride down the column (512 B per wave load),
//                                       DPP/shuffle reduction; 4 columns per 256-thread workgroup.
//   gemv_n: y = alpha * A  x + beta*y   lanes along rows (coalesced), the columns are split over workgroups into
//                                       chunks; per-chunk partial sums are combined in a fixed order by a second
//                                       tiny kernel (deterministic — no floating-point atomics).
// Roofline: bytes = 8*rows*cols per call (each block is read exactly once).
#include "internal.hpp"
#include "device_utils.hpp"

namespace calipso {

// 64-row blocks of a column that can hold non-zeros: [b0, b1) and [b2, b3) (empty ranges: b0 >= b1)
struct BlockRanges { int b0, b1, b2, b3; };
__device__ __forceinline__ BlockRanges column_blocks(const Sparsity& sp, int col, int rows) {
    BlockRanges r = {0, (rows + 63) / 64, 0, 0};
    if (sp.kind == SP_DENSE) return r;
    if (sp.kind == SP_LXX) { r.b0 = max(0, col - sp.hb) / 64; r.b1 = (min(rows, col + sp.hb + 1) + 63) / 64; return r; }
    const int* k = sp.kr + 4 * (col / 16);
    const int elo = k[0], ehi = k[1], clo = k[2], chi = k[3];
    if (sp.kind == SP_GX) { r.b0 = elo / 64; r.b1 = ehi > elo ? (ehi + 63) / 64 : r.b0; return r; }
    if (sp.kind == SP_HX) { r.b0 = clo / 64; r.b1 = chi > clo ? (chi + 63) / 64 : r.b0; return r; }
    r.b0 = elo / 64; r.b1 = ehi > elo ? (ehi + 63) / 64 : r.b0;                       // SP_Z: equality rows, then the cone rows at offset ne
    r.b2 = (sp.ne + clo) / 64; r.b3 = chi > clo ? (sp.ne + chi + 63) / 64 : r.b2;
    return r;
}
__device__ __forceinline__ bool block_active(const BlockRanges& r, int b) { return (b >= r.b0 && b < r.b1) || (b >= r.b2 && b < r.b3); }

// Structured column (analysed structure): only the 64-row blocks that can hold non-zeros are visited, 16 candidate blocks per round (all loads
// issued before the first use).  The terms go to the two accumulators exactly as the dense loop deals them (block b of a batch of 24 -> acc0 for even,
// acc1 for odd positions), in ascending order, so dropping the blocks that hold structural zeros does not change a bit of the result.
template <int NV>
__device__ __forceinline__ void structured_column(const BlockRanges& br, int rows, int lane, const double* __restrict__ a, const double* const (&x)[NV],
                                                  double (&acc0)[NV], double (&acc1)[NV]) {
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        const int lo = pass ? max(br.b2, br.b1) : br.b0, hi = pass ? br.b3 : br.b1;
        for (int w0 = lo; w0 < hi; w0 += 16) {
            double v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) { const int b = w0 + q, i = b * 64 + lane; v[q] = (b < hi && i < rows) ? a[i] : 0.0; }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int b = w0 + q, i = b * 64 + lane;
                const bool in = b < hi && i < rows;
                const bool odd = ((b % 24) & 1) != 0;
#pragma unroll
                for (int n = 0; n < NV; ++n) {
                    const double xv = in ? x[n][i] : 0.0;
                    acc0[n] = odd ? acc0[n] : fma(v[q], xv, acc0[n]);         // (fused, as the dense loop is)
                    acc1[n] = odd ? fma(v[q], xv, acc1[n]) : acc1[n];
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_gemv_t(Batch bt, Sparsity sp, int rows, int cols, const double* __restrict__ A, int ld, const double* __restrict__ x,
                                                 double* __restrict__ y, double alpha, double beta, const double* __restrict__ add = nullptr) {
    inst_shift(bt, A, x, y);
    if (add) inst_shift(bt, add);
    if (sp.kr) inst_shift_i(bt, sp.kr);
    // (x from global memory: one vector of nx or m doubles stays in the L1 of the compute unit — staged in LDS as k_gemv_t2 stages its TWO vectors, which together do not,
    // this kernel was 13 % slower: 12.1 against 10.7 us at C3)
    const int lane = threadIdx.x & 63;
    const int col = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (col >= cols) return;
    const double* a = A + (size_t)col * ld;
    const BlockRanges br = column_blocks(sp, col, rows);
    double acc0 = 0.0, acc1 = 0.0;
    if (sp.kind != SP_DENSE) {
        const double* const xs[1] = {x};
        double a0[1] = {0.0}, a1[1] = {0.0};
        structured_column<1>(br, rows, lane, a, xs, a0, a1);
        acc0 = a0[0]; acc1 = a1[0];
    } else
    // batches of 24 loads per lane, all issued before the first use (latency-bound otherwise)
    for (int base = 0; base < rows; base += 64 * 24) {
        if (sp.kind != SP_DENSE && !(base / 64 < max(br.b1, br.b3) && base / 64 + 24 > min(br.b0, br.b2 < br.b3 ? br.b2 : br.b0))) continue;   // nothing of this batch is inside the structure
        double v[24];
#pragma unroll
        for (int q = 0; q < 24; ++q) { const int i = base + lane + 64 * q; v[q] = (i < rows && block_active(br, base / 64 + q)) ? a[i] : 0.0; }
#pragma unroll
        for (int q = 0; q < 24; q += 2) {
            const int i = base + lane + 64 * q;
            acc0 = fma(v[q], i < rows ? x[i] : 0.0, acc0);
            acc1 = fma(v[q + 1], i + 64 < rows ? x[i + 64] : 0.0, acc1);
        }
    }
    const double r = wave_sum(acc0 + acc1);
    if (lane == 0) y[col] = add ? alpha * r + add[col] : ((beta == 0.0) ? alpha * r : alpha * r + beta * y[col]);
}

// y1 = A'x1 and y2 = A'x2 with ONE pass over A: the refinement residual needs [gx; hx]'(step_y; step_z) and the condensed solve that follows
// needs [gx; hx]'(Omega b_m) — both input vectors are known at the same time (vectors.hip: k_refine_local), so the largest block of a
// refinement round is read once for the two.  Same lane / summation layout as k_gemv_t.
__device__ __forceinline__ void gemv_t2_body(Batch bt, Sparsity sp, int bx, int rows, int cols, const double* __restrict__ A, int ld, const double* __restrict__ x1,
                                             const double* __restrict__ x2, double* __restrict__ y1, double* __restrict__ y2) {
    inst_shift(bt, A, x1, x2, y1, y2);
    if (sp.kr) inst_shift_i(bt, sp.kr);
    const int lane = threadIdx.x & 63;
    const int col = bx * 4 + (threadIdx.x >> 6);
    if (col >= cols) return;
    const double* a = A + (size_t)col * ld;
    const BlockRanges br = column_blocks(sp, col, rows);
    double p0 = 0.0, p1 = 0.0, q0 = 0.0, q1 = 0.0;
    if (sp.kind != SP_DENSE) {
        const double* const xs[2] = {x1, x2};
        double a0[2] = {0.0, 0.0}, a1[2] = {0.0, 0.0};
        structured_column<2>(br, rows, lane, a, xs, a0, a1);
        p0 = a0[0]; p1 = a1[0]; q0 = a0[1]; q1 = a1[1];
    } else
    for (int base = 0; base < rows; base += 64 * 24) {
        if (sp.kind != SP_DENSE && !(base / 64 < max(br.b1, br.b3) && base / 64 + 24 > min(br.b0, br.b2 < br.b3 ? br.b2 : br.b0))) continue;
        double v[24];
#pragma unroll
        for (int q = 0; q < 24; ++q) { const int i = base + lane + 64 * q; v[q] = (i < rows && block_active(br, base / 64 + q)) ? a[i] : 0.0; }
#pragma unroll
        for (int q = 0; q < 24; q += 2) {
            const int i = base + lane + 64 * q;
            const bool in0 = i < rows, in1 = i + 64 < rows;
            p0 = fma(v[q], in0 ? x1[i] : 0.0, p0); p1 = fma(v[q + 1], in1 ? x1[i + 64] : 0.0, p1);
            q0 = fma(v[q], in0 ? x2[i] : 0.0, q0); q1 = fma(v[q + 1], in1 ? x2[i + 64] : 0.0, q1);
        }
    }
    const double r1 = wave_sum(p0 + p1), r2 = wave_sum(q0 + q1);
    if (lane == 0) { y1[col] = r1; y2[col] = r2; }
}
// The same product of a DENSE block with the two vectors in LDS: every wavefront of a launch reads all of both, and together (40 KB at C3) they do not stay in the 32 KB L1
// of a compute unit beside the matrix stream — from global memory each wavefront issued twice as many loads for them as for its column (k_gemv_t2_and_n: 22.9 -> 21 us).
// The first batch of the column's loads goes out BEFORE the workgroup stages the vectors: they travel under the staging.  Same terms in the same order: same bits.
__device__ __forceinline__ void gemv_t2_dense_lds(Batch bt, int bx, int rows, int cols, const double* __restrict__ A, int ld, const double* __restrict__ x1,
                                                  const double* __restrict__ x2, double* __restrict__ y1, double* __restrict__ y2, double* __restrict__ xs) {
    inst_shift(bt, A, x1, x2, y1, y2);
    const int lane = threadIdx.x & 63;
    const int col = bx * 4 + (threadIdx.x >> 6);
    const bool valid = col < cols;
    const double* a = A + (size_t)(valid ? col : 0) * ld;
    double v[24];
#pragma unroll
    for (int q = 0; q < 24; ++q) { const int i = lane + 64 * q; v[q] = (valid && i < rows) ? a[i] : 0.0; }
    for (int i = threadIdx.x; i < rows; i += 256) { xs[i] = x1[i]; xs[rows + i] = x2[i]; }
    __syncthreads();
    if (!valid) return;
    double p0 = 0.0, p1 = 0.0, q0 = 0.0, q1 = 0.0;
    for (int base = 0; base < rows; base += 64 * 24) {
        if (base) {
#pragma unroll
            for (int q = 0; q < 24; ++q) { const int i = base + lane + 64 * q; v[q] = i < rows ? a[i] : 0.0; }
        }
#pragma unroll
        for (int q = 0; q < 24; q += 2) {
            const int i = base + lane + 64 * q;
            const bool in0 = i < rows, in1 = i + 64 < rows;
            p0 = fma(v[q], in0 ? xs[i] : 0.0, p0); p1 = fma(v[q + 1], in1 ? xs[i + 64] : 0.0, p1);
            q0 = fma(v[q], in0 ? xs[rows + i] : 0.0, q0); q1 = fma(v[q + 1], in1 ? xs[rows + i + 64] : 0.0, q1);
        }
    }
    const double r1 = wave_sum(p0 + p1), r2 = wave_sum(q0 + q1);
    if (lane == 0) { y1[col] = r1; y2[col] = r2; }
}
__global__ __launch_bounds__(256) void k_gemv_t2(Batch bt, Sparsity sp, int rows, int cols, const double* __restrict__ A, int ld, const double* __restrict__ x1,
                                                  const double* __restrict__ x2, double* __restrict__ y1, double* __restrict__ y2, int lds_x = 0) {
    extern __shared__ __attribute__((aligned(16))) double t2s_lds[];
    if (lds_x && sp.kind == SP_DENSE) { gemv_t2_dense_lds(bt, blockIdx.x, rows, cols, A, ld, x1, x2, y1, y2, t2s_lds); return; }
    gemv_t2_body(bt, sp, blockIdx.x, rows, cols, A, ld, x1, x2, y1, y2);
}

// the structure tables of `s` for a block of the given kind (dense unless calipso_hip_analyze_structure found a band)
static Sparsity sparsity_of(const calipso_hip_solver* s, int kind) {
    Sparsity sp;
    if (kind == SP_DENSE || s->band64 == 0) return sp;
    sp.kind = kind; sp.hb = s->half_bandwidth; sp.ne = s->d.ne; sp.kr = s->krange; sp.rowrange = s->zrow;
    return sp;
}

// the vector(s) of a transposed product staged in LDS by every workgroup (dense blocks; what fits the 48 KB a kernel gets without asking)
static int gemv_lds_x(size_t bytes) {
    return bytes <= 48 * 1024 ? 1 : 0;
}
void gemv_t(calipso_hip_solver* s, int rows, int cols, const double* A, int ld, const double* x, double* y, double alpha, double beta, int kind, const double* add) {
    if (cols == 0) return;
    if (blocks_gemv_t(s, kind, x, nullptr, y, nullptr, alpha, beta)) { if (add) launch_add(s, y, add, cols); return; }       // stage blocks (blocks.hip)
    if (s->compact) { s->err = "internal: a dense mat-vec was requested on a structured handle"; s->ldl_failed = true; return; }   // (sticky: the driver reports CALIPSO_ERR_HIP)
    const BatchSc B = batch_of(s);
    hipLaunchKernelGGL(k_gemv_t, dim3((cols + 3) / 4, 1, B.b.n), dim3(256), 0, s->stream, B.b, sparsity_of(s, kind), rows, cols, A, ld, x, y, alpha, beta, add);
}

void gemv_t2(calipso_hip_solver* s, int rows, int cols, const double* A, int ld, const double* x1, const double* x2, double* y1, double* y2, int kind) {
    if (cols == 0) return;
    if (blocks_gemv_t(s, kind, x1, x2, y1, y2, 1.0, 0.0)) return;
    if (s->compact) { s->err = "internal: a dense mat-vec was requested on a structured handle"; s->ldl_failed = true; return; }   // (sticky: the driver reports CALIPSO_ERR_HIP)
    const BatchSc B = batch_of(s);
    const int lds_x = gemv_lds_x(2 * sizeof(double) * (size_t)rows);
    hipLaunchKernelGGL(k_gemv_t2, dim3((cols + 3) / 4, 1, B.b.n), dim3(256), lds_x ? 2 * sizeof(double) * (size_t)rows : 0, s->stream, B.b, sparsity_of(s, kind), rows, cols, A, ld, x1, x2, y1, y2, lds_x);
}

constexpr int GN_ROWS = 256;    // rows per workgroup
constexpr int GN_MAXCHUNK = 64; // column chunks

__device__ __forceinline__ void gemv_n_partial_body(Batch bt, Sparsity sp, int bx, int by, int row_off, int rows, int cols, int chunk, const double* __restrict__ A, int ld,
                                                    const double* __restrict__ x, double* __restrict__ partial) {
    inst_shift(bt, A, x, partial);
    if (sp.rowrange) inst_shift_i(bt, sp.rowrange);
    const int i = bx * GN_ROWS + threadIdx.x;
    const int c0 = by * chunk;
    const int c1 = min(cols, c0 + chunk);
    // the chunk's entries of x: every lane needs all of them, at the same time.  As loads they were as many vector-memory instructions as the matrix itself (the
    // compiler does not turn them into scalar loads); a chunk of at most 64 columns is fetched ONCE — lane l holds x[c0 + l], loaded while every lane is still
    // active (v_readlane reads a lane's register whether or not the lane is) — and handed out by v_readlane
    const bool xb = c1 - c0 <= 64;
    const int ln = threadIdx.x & 63;
    double xr = (xb && c0 + ln < c1) ? x[c0 + ln] : 0.0;
    asm volatile("" : "+v"(xr));                  // (the chunk of x is in the registers of ALL 64 lanes before any lane stands aside: the load cannot sink below the predicates)
    // No lane leaves before the end: a lane beyond the last row (or whose row has nothing in this chunk) runs the loop with its loads predicated off, so that
    // every v_readlane below reads a lane that is still there.
    const bool inrow = i < rows;
    // columns of row i that can be non-zero (loads outside are predicated off; the summation order is that of the dense kernel)
    int jlo = 0, jhi = cols;
    if (inrow) {
        if (sp.kind == SP_LXX) { jlo = i - sp.hb; jhi = i + sp.hb + 1; }
        else if (sp.kind != SP_DENSE) { jlo = sp.rowrange[2 * (row_off + i)]; jhi = sp.rowrange[2 * (row_off + i) + 1]; }
    } else { jlo = 0; jhi = 0; }                  // (an empty range: every load of the lane is off)
    const bool any = inrow && !(sp.kind != SP_DENSE && (c1 <= jlo || c0 >= jhi));      // something of this chunk lies inside the row's range (else its partial sum is an exact 0.0)
    double acc0 = 0.0, acc1 = 0.0;
    auto xat = [&](int j) -> double {          // j is wave-uniform
        if (!xb) return x[j];
        const int l = j - c0;
        return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(xr), l), __builtin_amdgcn_readlane(__double2loint(xr), l));
    };
    // batches of 24 loads per lane, all issued before the first use
    for (int j0 = c0; j0 < c1; j0 += 24) {
        const bool batch_in = any && (sp.kind == SP_DENSE || !(j0 + 24 <= jlo || j0 >= jhi));                             // (outside: a batch of exact zeros, not added)
        double v[24];
#pragma unroll
        for (int q = 0; q < 24; ++q) { const int j = j0 + q; v[q] = (batch_in && j < c1 && j >= jlo && j < jhi) ? A[i + (size_t)j * ld] : 0.0; }
#pragma unroll
        for (int q = 0; q < 24; q += 2) {
            const double x0 = j0 + q < c1 ? xat(j0 + q) : 0.0, x1 = j0 + q + 1 < c1 ? xat(j0 + q + 1) : 0.0;        // (read by every lane, used by those inside)
            if (batch_in) { acc0 += v[q] * x0; acc1 += v[q + 1] * x1; }
        }
    }
    if (inrow) partial[(size_t)by * rows + i] = acc0 + acc1;
}
__global__ __launch_bounds__(GN_ROWS) void k_gemv_n_partial(Batch bt, Sparsity sp, int row_off, int rows, int cols, int chunk, const double* __restrict__ A, int ld,
                                                             const double* __restrict__ x, double* __restrict__ partial) {
    gemv_n_partial_body(bt, sp, blockIdx.x, blockIdx.y, row_off, rows, cols, chunk, A, ld, x, partial);
}
// The two mat-vecs a refinement residual needs are independent of each other — [gx; hx]'(v_y; v_z) and [gx; hx]'(Omega b_m) in one pass (gemv_t2), Lxx v_x
// (gemv_n) — and each is a 12 - 16 us kernel that does not fill the device for long: ONE launch runs both (the first nt2 workgroups take the columns of the
// transposed product, the others the (row block, column chunk) pairs of the plain one), the same code and the same sums as the two kernels.
__global__ __launch_bounds__(256) void k_gemv_t2_and_n(Batch bt, Sparsity spz, int rowsz, int colsz, const double* __restrict__ Z, int ldz, const double* __restrict__ x1,
                                                        const double* __restrict__ x2, double* __restrict__ y1, double* __restrict__ y2, int nt2, Sparsity spl, int rb,
                                                        int rowsl, int colsl, int chunk, const double* __restrict__ L, int ldl, const double* __restrict__ xl,
                                                        double* __restrict__ partial, const int* __restrict__ gate = nullptr, int gate_epoch = 0, int lds_x = 0) {
    static_assert(GN_ROWS == 256, "one block size for both bodies");
    if (gate && gate[0] == gate_epoch) return;        // (internal.hpp: gate)
    extern __shared__ __attribute__((aligned(16))) double t2_lds[];
    const int b = blockIdx.x;
    if (b < nt2 && lds_x && spz.kind == SP_DENSE) {
        gemv_t2_dense_lds(bt, b, rowsz, colsz, Z, ldz, x1, x2, y1, y2, t2_lds);
        return;
    }
    if (b < nt2) gemv_t2_body(bt, spz, b, rowsz, colsz, Z, ldz, x1, x2, y1, y2);
    else gemv_n_partial_body(bt, spl, (b - nt2) % rb, (b - nt2) / rb, 0, rowsl, colsl, chunk, L, ldl, xl, partial);
}

// y = alpha * sum_chunks partial + beta*y: 64 rows per workgroup, 4 lanes per row each summing every 4th chunk in a fixed order
__global__ __launch_bounds__(256) void k_gemv_n_reduce(Batch bt, int rows, int nchunk, const double* __restrict__ partial, double* __restrict__ y, double alpha, double beta,
                                                       const double* __restrict__ add = nullptr) {
    __shared__ double part[4][64];
    inst_shift(bt, partial, y);
    if (add) inst_shift(bt, add);
    const int r = threadIdx.x & 63, p = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + r;
    double acc = 0.0;
    if (i < rows) {
        // all of this thread's partials (at most GN_MAXCHUNK / 4 = 16) in flight together, summed in the same order
        static_assert(GN_MAXCHUNK <= 64, "16 partials per thread");
        for (int c0 = p; c0 < nchunk; c0 += 64) {
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) { const int c = c0 + 4 * u; v[u] = c < nchunk ? partial[(size_t)c * rows + i] : 0.0; }
#pragma unroll
            for (int u = 0; u < 16; ++u) if (c0 + 4 * u < nchunk) acc += v[u];
        }
    }
    part[p][r] = acc;
    __syncthreads();
    if (threadIdx.x < 64 && i < rows) {
        const double v = (part[0][r] + part[1][r]) + (part[2][r] + part[3][r]);
        y[i] = add ? alpha * v + add[i] : ((beta == 0.0) ? alpha * v : alpha * v + beta * y[i]);
    }
}

// Both products with ONE pass over A (rows x cols, column-major):   yt[j] = beta_t*yt[j] + sum_i A[i][j] u[i]   (A' u)
//                                                                    yn[i] = sum_j A[i][j] x[j]                  (A x)
// used by the matrix-free H*v of the refinement, where the stacked Jacobian [gx; hx] enters as [gx; hx] v_x and [gx; hx]' v_yz
// (residual_jacobian_variables.jl:19-47): the block is the largest HBM stream of a refinement round, so it is read once.
// One wavefront walks BOTH_CW consecutive columns of a row range of 64*BOTH_RQ rows: lanes stride down the column (512-byte
// loads), the A'u entry of a column is a wave reduction, the A x contributions stay in BOTH_RQ per-lane accumulators and are
// written once per wavefront as a partial row vector; partials are combined in a fixed order by k_gemv_n_reduce.
constexpr int BOTH_RQ = 16;   // rows per lane  -> 1024 rows per row range
constexpr int BOTH_CW = 16;   // columns per wavefront
__global__ __launch_bounds__(256) void k_gemv_both(Batch bt, Sparsity sp, int rows, int cols, int cw, const double* __restrict__ A, int ld, const double* __restrict__ x,
                                                    const double* __restrict__ u, double* __restrict__ part_n, double* __restrict__ part_t) {
    inst_shift(bt, A, x, u, part_n, part_t);
    if (sp.kr) inst_shift_i(bt, sp.kr);
    const int lane = threadIdx.x & 63;
    const int cg = blockIdx.x * 4 + (threadIdx.x >> 6);        // column group of this wavefront
    const int c0 = cg * cw;
    if (c0 >= cols) return;
    const int r0 = blockIdx.y * 64 * BOTH_RQ;
    double uu[BOTH_RQ], acc[BOTH_RQ];
#pragma unroll
    for (int q = 0; q < BOTH_RQ; ++q) { const int i = r0 + lane + 64 * q; uu[q] = i < rows ? u[i] : 0.0; acc[q] = 0.0; }
    const int c1 = min(cols, c0 + cw);
    const BlockRanges br = column_blocks(sp, c0, rows);        // (a column group of the structure tables is 16 = BOTH_CW columns)
    unsigned active = 0;
#pragma unroll
    for (int q = 0; q < BOTH_RQ; ++q) if (block_active(br, r0 / 64 + q)) active |= 1u << q;
    for (int j = c0; j < c1; ++j) {
        const double* a = A + (size_t)j * ld;
        double z[BOTH_RQ];
#pragma unroll
        for (int q = 0; q < BOTH_RQ; ++q) { const int i = r0 + lane + 64 * q; z[q] = (i < rows && ((active >> q) & 1u)) ? a[i] : 0.0; }   // one batch of loads
        const double xj = x[j];
        double t0 = 0.0, t1 = 0.0;
#pragma unroll
        for (int q = 0; q < BOTH_RQ; q += 2) {
            t0 += z[q] * uu[q]; t1 += z[q + 1] * uu[q + 1];
            acc[q] += z[q] * xj; acc[q + 1] += z[q + 1] * xj;
        }
        const double t = wave_sum(t0 + t1);
        if (lane == 0) part_t[(size_t)blockIdx.y * cols + j] = t;
    }
#pragma unroll
    for (int q = 0; q < BOTH_RQ; ++q) { const int i = r0 + lane + 64 * q; if (i < rows) part_n[(size_t)cg * rows + i] = acc[q]; }
}

// yt = A'u + beta_t*yt and yn = A x in one pass over A
void gemv_both(calipso_hip_solver* s, int rows, int cols, const double* A, int ld, const double* x, const double* u, double* yn, double* yt, double beta_t, int kind) {
    if (rows == 0 || cols == 0) return;
    if (s->compact) { s->err = "internal: a dense mat-vec was requested on a structured handle"; s->ldl_failed = true; return; }   // (sticky: the driver reports CALIPSO_ERR_HIP)
    const BatchSc B = batch_of(s);
    const int cw = BOTH_CW;   // (8 is 1 % faster for a single cache-resident instance, 16 for groups streaming from HBM)
    const int ncg = (cols + cw - 1) / cw, nrr = (rows + 64 * BOTH_RQ - 1) / (64 * BOTH_RQ);
    double* part_n = s->gemv_partial;                       // ncg x rows
    double* part_t = s->gemv_partial + (size_t)ncg * rows;  // nrr x cols
    hipLaunchKernelGGL(k_gemv_both, dim3((ncg + 3) / 4, nrr, B.b.n), dim3(256), 0, s->stream, B.b, sparsity_of(s, kind), rows, cols, cw, A, ld, x, u, part_n, part_t);
    hipLaunchKernelGGL(k_gemv_n_reduce, dim3((rows + 63) / 64, 1, B.b.n), dim3(256), 0, s->stream, B.b, rows, ncg, part_n, yn, 1.0, 0.0);
    hipLaunchKernelGGL(k_gemv_n_reduce, dim3((cols + 63) / 64, 1, B.b.n), dim3(256), 0, s->stream, B.b, cols, nrr, part_t, yt, 1.0, beta_t);
}

static void gemv_n_chunks(int rows, int cols, int& rb, int& nchunk, int& chunk) {
    // enough workgroups to cover the chip: rows/256 row blocks x nchunk column chunks ~ 1024 workgroups
    rb = (rows + GN_ROWS - 1) / GN_ROWS;
    nchunk = (768 + rb - 1) / rb;
    if (nchunk > GN_MAXCHUNK) nchunk = GN_MAXCHUNK;
    if (nchunk > (cols + 15) / 16) nchunk = (cols + 15) / 16;
    if (nchunk < 1) nchunk = 1;
    chunk = (cols + nchunk - 1) / nchunk;
    nchunk = (cols + chunk - 1) / chunk;
    if (cols == 0) nchunk = 0;
}

// y1 = Z'x1, y2 = Z'x2 (Z = [gx; hx], m x nx) and yl = Lxx xl in one launch + the reduction of the second (api.hip: refine_residual)
int gemv_refine_pair(calipso_hip_solver* s, const double* x1, const double* x2, double* y1, double* y2, const double* xl, double* yl, bool defer_reduce) {
    const Dims& d = s->d;
    if (d.m == 0 || s->blocks.on || s->compact) {              // no constraints, or the block kernels (blocks.hip): the two calls as they were
        if (d.m) gemv_t2(s, d.m, d.nx, s->Z, d.m, x1, x2, y1, y2, SP_Z);
        gemv_n(s, d.nx, d.nx, s->Lxx, d.nx, xl, yl, 1.0, 0.0, SP_LXX);
        return 0;
    }
    int rb, nchunk, chunk;
    gemv_n_chunks(d.nx, d.nx, rb, nchunk, chunk);
    const int nt2 = (d.nx + 3) / 4;
    const BatchSc B = batch_of(s);
    const bool timed = s->time_matvec && !s->cur;
    if (timed) (void)hipEventRecord(s->ev[5], s->stream);
    const size_t xbytes = 2 * sizeof(double) * (size_t)d.m;
    const int lds_x = gemv_lds_x(xbytes);
    hipLaunchKernelGGL(k_gemv_t2_and_n, dim3(nt2 + rb * nchunk, 1, B.b.n), dim3(256), lds_x ? xbytes : 0, s->stream, B.b, sparsity_of(s, SP_Z), d.m, d.nx, s->Z, d.m, x1, x2, y1, y2, nt2,
                       sparsity_of(s, SP_LXX), rb, d.nx, d.nx, chunk, s->Lxx, d.nx, xl, s->gemv_partial, s->gate_epoch ? s->gate : (const int*)nullptr, s->gate_epoch, lds_x);
    if (timed) { (void)hipEventRecord(s->ev[6], s->stream); s->time_matvec = false; s->matvec_timed = true; }
    if (defer_reduce) return nchunk;
    hipLaunchKernelGGL(k_gemv_n_reduce, dim3((d.nx + 63) / 64, 1, B.b.n), dim3(256), 0, s->stream, B.b, d.nx, nchunk, s->gemv_partial, yl, 1.0, 0.0);
    return 0;
}

void gemv_n(calipso_hip_solver* s, int rows, int cols, const double* A, int ld, const double* x, double* y, double alpha, double beta, int kind, const double* add) {
    if (rows == 0) return;
    if (blocks_gemv_n(s, kind, x, y, alpha, beta)) { if (add) launch_add(s, y, add, rows); return; }
    if (s->compact) { s->err = "internal: a dense mat-vec was requested on a structured handle"; s->ldl_failed = true; return; }   // (sticky: the driver reports CALIPSO_ERR_HIP)
    int rb, nchunk, chunk;
    gemv_n_chunks(rows, cols, rb, nchunk, chunk);
    const BatchSc B = batch_of(s);
    if (nchunk > 0)
        hipLaunchKernelGGL(k_gemv_n_partial, dim3(rb, nchunk, B.b.n), dim3(GN_ROWS), 0, s->stream, B.b, sparsity_of(s, kind), kind == SP_HX ? s->d.ne : 0, rows, cols, chunk, A, ld,
                           x, s->gemv_partial);
    hipLaunchKernelGGL(k_gemv_n_reduce, dim3((rows + 63) / 64, 1, B.b.n), dim3(256), 0, s->stream, B.b, rows, nchunk, s->gemv_partial, y, alpha, beta, add);
}

}  // namespace calipso
