// blocks.hip — stage blocks: packed storage and block kernels for stage-structured problems (the reference keeps such problems
// sparse end to end: src/trajectory_optimization/sparsity.jl:28-129, indices.jl:41-180, the nz -> dense copy of src/solver/evaluate.jl:37-121).
//
// calipso_hip_analyze_structure finds, per row of the stacked Jacobian [gx; hx], the range of columns that can be non-zero, and the reach of
// every column of the Lagrangian Hessian.  calipso_hip_set_stage_blocks turns that into
//   Z blocks   maximal runs of consecutive constraint rows with the SAME column range (a dynamics constraint between stages t and t + 1: nd rows x
//              (n_t + n_t+1) columns; the cone rows of a stage: rows x n_t columns), each stored twice, contiguously: column-major (for Z x: lanes
//              along the rows) and row-major (for Z'u: lanes along the columns);
//   L blocks   the diagonal blocks of Lxx (columns that no Hessian entry couples to an earlier block start a new one), both orientations;
//   segments   the partition of the columns by all block boundaries (at most 64 wide): the tiles of the Schur complement
// and from then on the mat-vecs of the Newton step and the Schur-complement kernel work on the packed blocks:
//   k_bgemv_n / k_bgemv_t   one workgroup per block / per segment, every load a full 512-byte run of the packed data, no reductions across lanes
//   k_schur_blocks          one workgroup per pair of segments that some block couples: S[a][b] = Lsym[a][b] + ep I + sum over the blocks that cover
//                           both of B[:, a]' Omega B[:, b] on the fp64 matrix cores, the blocks in ascending order; Omega (omega_y, the nonnegative
//                           weights, the W block of a second-order cone) is applied while the operand is staged, so WH = Omega hx is never formed
// instead of the dense-layout kernels with predicated loads (gemv.hip, schur.hip), which spend their time on the zeros between the blocks: the
// 128 x 128 tiles of k_schur do 14x the useful flops at BASELINE config C4's trajectory structure (41 stages of 56 variables).
// The dense buffers stay the interchange format of the uploads (set_field, scatter, device evaluators write them as before; a pack kernel refreshes the
// blocks on the same stream right behind every such write).  The packed values live in the slab region of Lsym (unused in this mode: the Schur kernel
// symmetrises from the packed Hessian blocks itself), so a group addresses them like every other per-instance buffer.  Results agree with the dense
// treatment to rounding, not bitwise (the sums run block by block): the mode is opt-in and a member of a group gets the bits of the same handle alone.
#include <algorithm>
#include <map>
#include <numeric>
#include <string>
#include <vector>

#include "internal.hpp"
#include "device_utils.hpp"

namespace calipso {

typedef double v4d __attribute__((ext_vector_type(4)));

// ---- pack --------------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_blocks_pack_z(Batch bt, const ZBlock* __restrict__ blk, int m, const double* __restrict__ Z, double* __restrict__ pk, int rlo = 0, int rhi = 1 << 30) {
    inst_shift(bt, Z, pk);
    const ZBlock b = blk[blockIdx.x];
    const int total = b.nrows * b.ncols;
    for (int idx = threadIdx.x; idx < total; idx += 256) {
        const int i = idx % b.nrows, j = idx / b.nrows;
        if (b.row0 + i < rlo || b.row0 + i >= rhi) continue;      // (only the rows of the Jacobian that was written: blocks_pack_from)
        const double v = Z[(size_t)(b.row0 + i) + (size_t)(b.col0 + j) * m];
        pk[b.off_c + idx] = v;                                 // column-major, ld = nrows
        pk[b.off_r + (size_t)i * b.ncols + j] = v;             // row-major, ld = ncols
    }
}
__global__ __launch_bounds__(256) void k_blocks_pack_l(Batch bt, const LBlock* __restrict__ blk, int nx, const double* __restrict__ L, double* __restrict__ pk) {
    inst_shift(bt, L, pk);
    const LBlock b = blk[blockIdx.x];
    const int total = b.n * b.n;
    for (int idx = threadIdx.x; idx < total; idx += 256) {
        const int i = idx % b.n, j = idx / b.n;
        const double v = L[(size_t)(b.c0 + i) + (size_t)(b.c0 + j) * nx];
        pk[b.off_c + idx] = v;
        pk[b.off_r + (size_t)i * b.n + j] = v;
    }
}

// ---- y[rows of a block] = B x[columns of the block]: one workgroup per block, lanes along the rows, four column parts ---------------------------
// rows [rlo, rhi) of the stacked Jacobian take part (gx only, hx only, or both); vectors indexed from row rlo
// (blockIdx.y: right-hand-side column of the multi-column use of differentiate!, strides ldx / ldy; one column otherwise)
__global__ __launch_bounds__(256) void k_bgemv_n(Batch bt, const ZBlock* __restrict__ blk, int rlo, int rhi, const double* __restrict__ pk, const double* __restrict__ x,
                                                  double* __restrict__ y, long long ldx = 0, long long ldy = 0) {
    __shared__ double part[4][64];
    inst_shift(bt, pk, x, y);
    x += (long long)blockIdx.y * ldx; y += (long long)blockIdx.y * ldy;
    const ZBlock b = blk[blockIdx.x];
    if (b.row0 < rlo || b.row0 >= rhi) return;
    const int lane = threadIdx.x & 63, p = threadIdx.x >> 6;
    for (int i0 = 0; i0 < b.nrows; i0 += 64) {
        const int i = i0 + lane;
        double acc = 0.0;
        if (i < b.nrows) {
            const double* a = pk + b.off_c + i;
            // eight of this lane's columns in flight together (the blocks are small: the kernel is a chain of memory round trips, not a stream); same order of the sum
            for (int j0 = p; j0 < b.ncols; j0 += 32) {
                double av[8], xv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int j = j0 + 4 * u; const bool in = j < b.ncols; av[u] = in ? a[(size_t)j * b.nrows] : 0.0; xv[u] = in ? x[b.col0 + j] : 0.0; }
#pragma unroll
                for (int u = 0; u < 8; ++u) if (j0 + 4 * u < b.ncols) acc += av[u] * xv[u];
            }
        }
        part[p][lane] = acc;
        __syncthreads();
        if (p == 0 && i < b.nrows) y[b.row0 - rlo + i] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
        __syncthreads();
    }
}
// y[columns of a segment] = alpha * sum over the covering blocks (ascending) of B[:, segment]' u[rows of the block] + beta y: lanes along the columns
template <int NV>
__global__ __launch_bounds__(256) void k_bgemv_t(Batch bt, const Segment* __restrict__ seg, const int* __restrict__ segblk, const ZBlock* __restrict__ blk, int rlo, int rhi,
                                                  const double* __restrict__ pk, const double* __restrict__ u1, const double* __restrict__ u2, double* __restrict__ y1,
                                                  double* __restrict__ y2, double alpha, double beta, long long ldu = 0, long long ldy = 0) {
    __shared__ double part[NV][4][64];
    inst_shift(bt, pk, u1, y1);
    if (NV == 2) inst_shift(bt, u2, y2);
    u1 += (long long)blockIdx.y * ldu; y1 += (long long)blockIdx.y * ldy;        // (blockIdx.y: right-hand-side column, NV = 1 only)
    const Segment sg = seg[blockIdx.x];
    const int lane = threadIdx.x & 63, p = threadIdx.x >> 6;
    double a1 = 0.0, a2 = 0.0;
    if (lane < sg.nc) {
        for (int q = 0; q < sg.count; ++q) {
            const ZBlock b = blk[segblk[sg.first + q]];
            if (b.row0 < rlo || b.row0 >= rhi) continue;
            const double* a = pk + b.off_r + (sg.c0 - b.col0) + lane;
            const double* v1 = u1 + (b.row0 - rlo);
            const double* v2 = NV == 2 ? u2 + (b.row0 - rlo) : nullptr;
            for (int i0 = p; i0 < b.nrows; i0 += 32) {          // eight rows in flight together, same order of the sums
                double ev[8], w1[8], w2[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + 4 * u; const bool in = i < b.nrows;
                    ev[u] = in ? a[(size_t)i * b.ncols] : 0.0; w1[u] = in ? v1[i] : 0.0; w2[u] = (NV == 2 && in) ? v2[i] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) if (i0 + 4 * u < b.nrows) { a1 += ev[u] * w1[u]; if (NV == 2) a2 += ev[u] * w2[u]; }
            }
        }
    }
    part[0][p][lane] = a1;
    if (NV == 2) part[1][p][lane] = a2;
    __syncthreads();
    if (p == 0 && lane < sg.nc) {
        const double r1 = (part[0][0][lane] + part[0][1][lane]) + (part[0][2][lane] + part[0][3][lane]);
        const int j = sg.c0 + lane;
        y1[j] = beta == 0.0 ? alpha * r1 : alpha * r1 + beta * y1[j];
        if (NV == 2) { const double r2 = (part[1][0][lane] + part[1][1][lane]) + (part[1][2][lane] + part[1][3][lane]); y2[j] = r2; }
    }
}
// y = Lxx x (trans = 0) or Lxx' x (1) over the diagonal blocks: one workgroup per block and 64 rows of it
__global__ __launch_bounds__(256) void k_bgemv_l(Batch bt, const LBlock* __restrict__ blk, int trans, const double* __restrict__ pk, const double* __restrict__ x,
                                                  double* __restrict__ y, double alpha, double beta) {
    __shared__ double part[4][64];
    inst_shift(bt, pk, x, y);
    const LBlock b = blk[blockIdx.x];
    const int lane = threadIdx.x & 63, p = threadIdx.x >> 6;
    const int i = blockIdx.y * 64 + lane;
    if ((int)blockIdx.y * 64 >= b.n) return;
    double acc = 0.0;
    if (i < b.n) {
        const double* a = pk + (trans ? b.off_r : b.off_c) + i;        // row-major copy read "down the rows" = the transpose
        for (int j0 = p; j0 < b.n; j0 += 32) {                  // eight columns in flight together, same order of the sum
            double av[8], xv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int j = j0 + 4 * u; const bool in = j < b.n; av[u] = in ? a[(size_t)j * b.n] : 0.0; xv[u] = in ? x[b.c0 + j] : 0.0; }
#pragma unroll
            for (int u = 0; u < 8; ++u) if (j0 + 4 * u < b.n) acc += av[u] * xv[u];
        }
    }
    part[p][lane] = acc;
    __syncthreads();
    if (p == 0 && i < b.n) {
        const double r = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
        y[b.c0 + i] = beta == 0.0 ? alpha * r : alpha * r + beta * y[b.c0 + i];
    }
}

// ---- Schur complement by segment pairs ---------------------------------------------------------------------------------------------------------
// S[a][b] (na x nb <= 64 x 64) = Lsym[a][b] + ep I + sum over the blocks covering both segments of B[:, a]' Omega B[:, b].  256 threads: wavefront w forms
// rows 16 w .. 16 w + 15 of the tile (four 16 x 16 MFMA tiles).  Rows of a block are taken in chunks of at most SB_KC that never split a second-order
// cone.  LDS panels are k-fastest with stride SB_KC + 2 (the conflict-free fragment layout of schur.hip / ldl.hip).
// SB_KC = 32 (3 panels x 64 x 34 doubles = 51 KB of LDS: three workgroups per CU — the launch bound holds the registers to 168 for that: 2 spilled, +3 % on C4T with
// 192 instances against two per CU) unless a cone is wider than that (then 64: one workgroup per CU)
template <int SB_KC>
__global__ __launch_bounds__(256, SB_KC == 32 ? 3 : 1) void k_schur_blocks(BatchSc bt, Dims d, ConeDev cd, const SegPair* __restrict__ pairs, const int* __restrict__ pairblk, const Segment* __restrict__ seg,
                                                       const ZBlock* __restrict__ blk, const LBlock* __restrict__ lblk, const double* __restrict__ pk, const double* __restrict__ wz,
                                                       const double* __restrict__ Wsoc, double* __restrict__ S, int packed_S, double* __restrict__ Aval, long long sA,
                                                       const int* __restrict__ inv) {
    constexpr int SB_LD = SB_KC + 2;
    __shared__ double As[64 * SB_LD];      // As[i][k] = B[k][a-column i]
    __shared__ double Bs[64 * SB_LD];      // Bs[j][k] = (Omega B)[k][b-column j]
    __shared__ double Rs[64 * SB_LD];      // raw rows of a second-order cone before its W block is applied
    inst_shift(bt.b, pk, wz, Wsoc, S);
    const Scalars sc = bt.scal(blockIdx.z);
    const SegPair pr = pairs[blockIdx.x];
    const Segment sa = seg[pr.a], sb = seg[pr.b];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fk = lane >> 4;
    const double omega_y = -1.0 / (-1.0 / (sc.rho + sc.ep) + (0.0 - sc.ed));
    v4d acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (v4d){0.0, 0.0, 0.0, 0.0};
    for (int q = 0; q < pr.count; ++q) {
        const ZBlock b = blk[pairblk[pr.first + q]];
        const int oa = sa.c0 - b.col0, ob = sb.c0 - b.col0;
        for (int k0 = 0; k0 < b.nrows;) {
            // chunk [k0, k1): at most SB_KC rows, cones whole
            int k1 = min(b.nrows, k0 + SB_KC);
            if (k1 < b.nrows && b.row0 + k1 >= d.ne + d.q) {
                const int e = b.row0 + k1 - d.ne;                        // cone-local index of the first row after the chunk
                const int j = cd.entry_soc[e];
                if (j >= 0 && cd.soc_start[j] < e) k1 = d.ne + cd.soc_start[j] - b.row0;   // the cone that would be split goes to the next chunk
            }
            const int kn = k1 - k0;
            __syncthreads();
            {
                // every load of the chunk in flight before the first use (64 * SB_KC / 256 entries of each operand per thread): the blocks are small and the
                // kernel is a chain of memory round trips — one entry at a time made it eight times as long as it has to be
                constexpr int NIT = 64 * SB_KC / 256;
                double va[NIT], rw[NIT], wv[NIT];
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int idx = tid + 256 * it, k = idx % SB_KC, i = idx / SB_KC;
                    const bool inb = k < kn;
                    const int row = b.row0 + k0 + k;
                    va[it] = (inb && i < sa.nc) ? pk[b.off_c + (size_t)(oa + i) * b.nrows + k0 + k] : 0.0;
                    rw[it] = (inb && i < sb.nc) ? pk[b.off_c + (size_t)(ob + i) * b.nrows + k0 + k] : 0.0;
                    wv[it] = (inb && i < sb.nc && row >= d.ne && row < d.ne + d.q) ? wz[row - d.ne] : 0.0;
                }
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int idx = tid + 256 * it, k = idx % SB_KC, i = idx / SB_KC;
                    double vb = 0.0, vr = 0.0;
                    if (k < kn && i < sb.nc) {
                        const int row = b.row0 + k0 + k;                 // row of the stacked Jacobian
                        if (row < d.ne) vb = omega_y * rw[it];
                        else if (row < d.ne + d.q) vb = wv[it] * rw[it];
                        else vr = rw[it];                                 // a second-order cone row: W is applied below
                    }
                    As[i * SB_LD + k] = va[it]; Bs[i * SB_LD + k] = vb; Rs[i * SB_LD + k] = vr;
                }
            }
            __syncthreads();
            if (b.row0 + k1 > d.ne + d.q) {                              // (Omega_z B)[k][j] = sum_k' W[k][k'] B[k'][j] inside every cone of the chunk
                for (int idx = tid; idx < 64 * SB_KC; idx += 256) {
                    const int k = idx % SB_KC, i = idx / SB_KC;
                    const int row = b.row0 + k0 + k;
                    if (k < kn && i < sb.nc && row >= d.ne + d.q) {
                        const int e = row - d.ne, j = cd.entry_soc[e];
                        const int st = cd.soc_start[j], dim = cd.soc_dim[j];
                        const double* W = Wsoc + cd.soc_woff[j];
                        const int kc = d.ne + st - b.row0 - k0;          // chunk-local index of the cone's first row
                        double s = 0.0;
                        for (int c = 0; c < dim; ++c) s += W[(e - st) + c * dim] * Rs[i * SB_LD + kc + c];
                        Bs[i * SB_LD + k] = s;
                    }
                }
                __syncthreads();
            }
            for (int kk = 0; kk < (kn + 3) / 4; ++kk) {
                const double a = As[(wave * 16 + fr) * SB_LD + 4 * kk + fk];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const double bb = Bs[(t * 16 + fr) * SB_LD + 4 * kk + fk];
                    acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bb, acc[t], 0, 0, 0);    // D[row i][col j]: lane holds (i = fk + 4 r, j = fr)
                }
            }
            k0 = k1;
        }
    }
    // epilogue: + Lsym (the Hessian as a triu-only factorisation sees it: entry (i, j) = Lxx[min][max], qdldl.jl:145-147) + ep on the diagonal
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = wave * 16 + fk + 4 * r, j = t * 16 + fr;
            if (i < sa.nc && j < sb.nc) {
                const int gi = sa.c0 + i, gj = sb.c0 + j;
                double v = acc[t][r];
                if (pr.lblock >= 0) {
                    const LBlock lb = lblk[pr.lblock];
                    const int li = gi - lb.c0, lj = gj - lb.c0;
                    v += pk[lb.off_c + (size_t)min(li, lj) + (size_t)max(li, lj) * lb.n];
                }
                if (gi == gj) v += sc.ep;
                if (packed_S) {                                                // structured handles: the tile, contiguous (column-major, ld = rows of segment a)
                    const size_t cell = pr.soff + i + (size_t)j * sa.nc;
                    S[cell] = v;
                    if (inv) { const int e = inv[cell]; if (e >= 0) Aval[(size_t)bt.b.slot[blockIdx.z] * (size_t)sA + (size_t)e] = v; }      // ... and the multifrontal factorisation's copy (no gather launch)
                }
                else S[(size_t)gi + (size_t)gj * d.NP] = v;
            }
        }
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------------------
void blocks_release(calipso_hip_solver* s) {
    StageBlocks& B = s->blocks;
    for (void* p : {(void*)B.d_blk, (void*)B.d_lblk, (void*)B.d_seg, (void*)B.d_segblk, (void*)B.d_pairs, (void*)B.d_pairblk, (void*)B.d_colrange, (void*)B.d_rowcov, (void*)B.d_jdesc, (void*)B.d_hdesc}) if (p) (void)hipFree(p);
    B = StageBlocks();
}

// refresh the packed copies from the dense buffers (on the handle's stream, right behind whatever wrote them)
void blocks_pack(calipso_hip_solver* s, bool z, bool l) {
    StageBlocks& B = s->blocks;
    if (!B.on || s->compact) return;                                    // (a structured handle has no dense buffers to pack from: its uploads go into the blocks directly)
    const Batch one;                                                    // the handle itself (uploads are per handle, also for members of a group)
    if (z && B.nblk) hipLaunchKernelGGL(k_blocks_pack_z, dim3(B.nblk), dim3(256), 0, s->stream, one, B.d_blk, s->d.m, s->Z, s->Lsym);
    if (l && B.nlb) hipLaunchKernelGGL(k_blocks_pack_l, dim3(B.nlb), dim3(256), 0, s->stream, one, B.d_lblk, s->d.nx, s->Lxx, s->Lsym);
}

// A device evaluator on a structured handle writes the dense ProblemData layout into scratch arrays of the handle (api.hip: device_evaluate); from there the values go
// into the blocks, and what lies outside the declared structure must be zero: one pass over the dense array per check (the price of a dense interchange format).
__global__ __launch_bounds__(256) void k_check_z_outside(const ZBlock* __restrict__ blk, int m, int nx, const double* __restrict__ Z, int* __restrict__ flag, int rlo, int rhi) {
    const ZBlock b = blk[blockIdx.x];
    const int lane = threadIdx.x & 63, p = threadIdx.x >> 6;
    int bad = 0;
    for (int i0 = 0; i0 < b.nrows; i0 += 64) {
        const int i = i0 + lane;
        if (i >= b.nrows || b.row0 + i < rlo || b.row0 + i >= rhi) continue;
        for (int j = p; j < nx; j += 4) if ((j < b.col0 || j >= b.col0 + b.ncols) && Z[(b.row0 + i) + (size_t)j * m] != 0.0) bad = 1;
    }
    if (bad) atomicOr(flag, 1);
}
// rows of [gx; hx] that NO block covers must be zero altogether (k_check_z_outside only visits the rows of a block)
__global__ __launch_bounds__(256) void k_check_z_uncovered(const int* __restrict__ rowcov, int m, int nx, const double* __restrict__ Z, int* __restrict__ flag, int rlo, int rhi) {
    const int r = rlo + (int)blockIdx.x * 64 + (threadIdx.x & 63), p = threadIdx.x >> 6;
    if (r >= rhi || rowcov[r]) return;
    int bad = 0;
    for (int j = p; j < nx; j += 4) if (Z[r + (size_t)j * m] != 0.0) bad = 1;
    if (bad) atomicOr(flag, 1);
}
__global__ void k_check_l_outside(int nx, const int* __restrict__ colrange, const double* __restrict__ L, int* __restrict__ flag) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (size_t)nx * nx) return;
    const int i = (int)(e % nx), j = (int)(e / nx);
    if ((i < colrange[2 * j] || i >= colrange[2 * j + 1]) && L[e] != 0.0) atomicOr(flag, 2);
}
int blocks_pack_from(calipso_hip_solver* s, const double* L, const double* Z, bool l, bool zg, bool zh) {
    StageBlocks& B = s->blocks;
    if (!B.on) return CALIPSO_ERR_ARGUMENT;
    const Dims& d = s->d;
    const Batch one;
    int* flag = s->icount + 60;
    CK(hipMemsetAsync(flag, 0, sizeof(int), s->stream));
    // only the rows of the Jacobians the evaluator was asked for are checked and packed (gx: [0, ne), hx: [ne, m)): the others keep what set_field / scatter / an
    // earlier evaluation put into their blocks
    const int rlo = zg ? 0 : d.ne, rhi = zh ? d.m : d.ne;
    if ((zg || zh) && rhi > rlo) {
        if (!B.d_rowcov) {          // which rows some block covers (host copy of the blocks: once per handle)
            std::vector<int> cov((size_t)std::max(1, d.m), 0);
            for (const ZBlock& b : B.h_blk) for (int i = 0; i < b.nrows; ++i) if (b.row0 + i < d.m) cov[b.row0 + i] = 1;
            CK(hipMalloc((void**)&B.d_rowcov, cov.size() * sizeof(int)));
            CK(hipMemcpyAsync(B.d_rowcov, cov.data(), cov.size() * sizeof(int), hipMemcpyHostToDevice, s->stream));
            CK(hipStreamSynchronize(s->stream));      // (cov leaves scope)
        }
        hipLaunchKernelGGL(k_check_z_uncovered, dim3((rhi - rlo + 63) / 64), dim3(256), 0, s->stream, B.d_rowcov, d.m, d.nx, Z, flag, rlo, rhi);
        if (B.nblk) {
            hipLaunchKernelGGL(k_check_z_outside, dim3(B.nblk), dim3(256), 0, s->stream, B.d_blk, d.m, d.nx, Z, flag, rlo, rhi);
            hipLaunchKernelGGL(k_blocks_pack_z, dim3(B.nblk), dim3(256), 0, s->stream, one, B.d_blk, d.m, Z, s->Lsym, rlo, rhi);
        }
    }
    if (l && B.nlb) {
        if (B.d_colrange) hipLaunchKernelGGL(k_check_l_outside, dim3((unsigned)(((size_t)d.nx * d.nx + 255) / 256)), dim3(256), 0, s->stream, d.nx, B.d_colrange, L, flag);
        hipLaunchKernelGGL(k_blocks_pack_l, dim3(B.nlb), dim3(256), 0, s->stream, one, B.d_lblk, d.nx, L, s->Lsym);
    }
    CK(hipMemcpyAsync(s->hicount + 60, flag, sizeof(int), hipMemcpyDeviceToHost, s->stream));
    CK(hipStreamSynchronize(s->stream));
    if (s->hicount[60] != 0) { s->err = "the device evaluator wrote non-zeros outside the declared structure of the handle"; return CALIPSO_ERR_ARGUMENT; }
    return CALIPSO_OK;
}

// ---- a device evaluator that writes the blocks (calipso_device_block_eval_fn): descriptors of the column-major copies, and the row-major copies behind it ----------
int blocks_descriptors(calipso_hip_solver* s) {
    StageBlocks& B = s->blocks;
    if (!B.on) return CALIPSO_ERR_ARGUMENT;
    if (!B.h_jdesc.empty() || !B.h_hdesc.empty()) return CALIPSO_OK;
    for (const ZBlock& b : B.h_blk) B.h_jdesc.push_back({b.row0, b.nrows, b.col0, b.ncols, s->Lsym + b.off_c, b.nrows});
    for (const LBlock& b : B.h_lblk) B.h_hdesc.push_back({b.c0, b.n, b.c0, b.n, s->Lsym + b.off_c, b.n});
    auto up = [&](const std::vector<calipso_device_block>& h, calipso_device_block** d) {
        if (hipMalloc((void**)d, std::max<size_t>(1, h.size()) * sizeof(calipso_device_block)) != hipSuccess) return false;
        return h.empty() || hipMemcpy(*d, h.data(), h.size() * sizeof(calipso_device_block), hipMemcpyHostToDevice) == hipSuccess;
    };
    if (!up(B.h_jdesc, &B.d_jdesc) || !up(B.h_hdesc, &B.d_hdesc)) { s->err = "block descriptors: device allocation failed"; return CALIPSO_ERR_HIP; }
    return CALIPSO_OK;
}
__global__ __launch_bounds__(256) void k_blocks_mirror_z(Batch bt, const ZBlock* __restrict__ blk, double* __restrict__ pk, int rlo, int rhi) {
    inst_shift(bt, pk);
    const ZBlock b = blk[blockIdx.x];
    const int total = b.nrows * b.ncols;
    for (int idx = threadIdx.x; idx < total; idx += 256) {
        const int i = idx % b.nrows, j = idx / b.nrows;
        if (b.row0 + i < rlo || b.row0 + i >= rhi) continue;
        pk[b.off_r + (size_t)i * b.ncols + j] = pk[b.off_c + idx];
    }
}
__global__ __launch_bounds__(256) void k_blocks_mirror_l(Batch bt, const LBlock* __restrict__ blk, double* __restrict__ pk) {
    inst_shift(bt, pk);
    const LBlock b = blk[blockIdx.x];
    const int total = b.n * b.n;
    for (int idx = threadIdx.x; idx < total; idx += 256) { const int i = idx % b.n, j = idx / b.n; pk[b.off_r + (size_t)i * b.n + j] = pk[b.off_c + idx]; }
}
void blocks_mirror(calipso_hip_solver* s, bool l, bool zg, bool zh) {
    StageBlocks& B = s->blocks;
    if (!B.on) return;
    const Batch one;
    const int rlo = zg ? 0 : s->d.ne, rhi = zh ? s->d.m : s->d.ne;
    if ((zg || zh) && rhi > rlo && B.nblk) hipLaunchKernelGGL(k_blocks_mirror_z, dim3(B.nblk), dim3(256), 0, s->stream, one, B.d_blk, s->Lsym, rlo, rhi);
    if (l && B.nlb) hipLaunchKernelGGL(k_blocks_mirror_l, dim3(B.nlb), dim3(256), 0, s->stream, one, B.d_lblk, s->Lsym);
}

static bool blocks_usable(const calipso_hip_solver* s) { return s->blocks.on && s->blocks_effective; }

bool blocks_gemv_n(calipso_hip_solver* s, int kind, const double* x, double* y, double alpha, double beta) {
    if (!blocks_usable(s) || alpha != 1.0 || beta != 0.0) return false;
    const StageBlocks& B = s->blocks;
    const BatchSc bs = batch_of(s);
    const Dims& d = s->d;
    if (kind == SP_LXX) {
        hipLaunchKernelGGL(k_bgemv_l, dim3(B.nlb, (B.max_lb + 63) / 64, bs.b.n), dim3(256), 0, s->stream, bs.b, B.d_lblk, 0, s->Lsym, x, y, alpha, beta);
        return true;
    }
    const int rlo = kind == SP_HX ? d.ne : 0, rhi = kind == SP_GX ? d.ne : d.m;
    if (kind != SP_Z && kind != SP_GX && kind != SP_HX) return false;
    hipLaunchKernelGGL(k_bgemv_n, dim3(B.nblk, 1, bs.b.n), dim3(256), 0, s->stream, bs.b, B.d_blk, rlo, rhi, s->Lsym, x, y);
    return true;
}
bool blocks_gemv_t(calipso_hip_solver* s, int kind, const double* u1, const double* u2, double* y1, double* y2, double alpha, double beta) {
    if (!blocks_usable(s)) return false;
    const StageBlocks& B = s->blocks;
    const BatchSc bs = batch_of(s);
    const Dims& d = s->d;
    if (kind == SP_LXX) {
        if (u2) return false;
        hipLaunchKernelGGL(k_bgemv_l, dim3(B.nlb, (B.max_lb + 63) / 64, bs.b.n), dim3(256), 0, s->stream, bs.b, B.d_lblk, 1, s->Lsym, u1, y1, alpha, beta);
        return true;
    }
    if (kind != SP_Z && kind != SP_GX && kind != SP_HX) return false;
    const int rlo = kind == SP_HX ? d.ne : 0, rhi = kind == SP_GX ? d.ne : d.m;
    if (u2) hipLaunchKernelGGL(k_bgemv_t<2>, dim3(B.nseg, 1, bs.b.n), dim3(256), 0, s->stream, bs.b, B.d_seg, B.d_segblk, B.d_blk, rlo, rhi, s->Lsym, u1, u2, y1, y2, 1.0, 0.0);
    else hipLaunchKernelGGL(k_bgemv_t<1>, dim3(B.nseg, 1, bs.b.n), dim3(256), 0, s->stream, bs.b, B.d_seg, B.d_segblk, B.d_blk, rlo, rhi, s->Lsym, u1, (const double*)nullptr, y1, (double*)nullptr, alpha, beta);
    return true;
}
// Y(:, c) = [gx; hx] X(:, c) and Y(:, c) = [gx; hx]' U(:, c) + beta Y(:, c) for p columns in one launch each (differentiate! on a handle that works on blocks)
bool blocks_gemm_n(calipso_hip_solver* s, const double* X, long long ldx, double* Y, long long ldy, int p) {
    if (!blocks_usable(s) || p < 1) return false;
    const StageBlocks& B = s->blocks;
    const BatchSc bs = batch_of(s);
    hipLaunchKernelGGL(k_bgemv_n, dim3(B.nblk, p, 1), dim3(256), 0, s->stream, bs.b, B.d_blk, 0, s->d.m, s->Lsym, X, Y, ldx, ldy);
    return true;
}
bool blocks_gemm_t(calipso_hip_solver* s, const double* U, long long ldu, double* Y, long long ldy, int p, double beta) {
    if (!blocks_usable(s) || p < 1) return false;
    const StageBlocks& B = s->blocks;
    const BatchSc bs = batch_of(s);
    hipLaunchKernelGGL(k_bgemv_t<1>, dim3(B.nseg, p, 1), dim3(256), 0, s->stream, bs.b, B.d_seg, B.d_segblk, B.d_blk, 0, s->d.m, s->Lsym, U, (const double*)nullptr, Y, (double*)nullptr, 1.0, beta,
                       ldu, ldy);
    return true;
}
bool blocks_schur(calipso_hip_solver* s) {
    if (!blocks_usable(s)) return false;
    const StageBlocks& B = s->blocks;
    const BatchSc bs = batch_of(s);
    if (!s->compact && !(s->stage_parallel && s->spS)) {
        // the blocked LDL^T factors S in place: what it left between the pair tiles (fill-in) must read as zero again, and the padded rows as identity
        // (the multifrontal path gathers S into storage of its own and needs neither)
        for (int k = 0; k < bs.b.n; ++k) (void)hipMemsetAsync(s->S + bs.b.delta[k], 0, sizeof(double) * (size_t)s->d.NP * s->d.NP, s->stream);
        launch_pad_identity(s);
    }
    // a structured handle whose S goes through the multifrontal factorisation: the kernel writes that factorisation's values too (structure.hip: spS_inv)
    double* Aval = nullptr; long long sA = 0;
    bool direct = s->compact && s->stage_parallel && s->spS && s->spS_inv && bs.b.n <= sparse_batch(s->spS);
    // (the kernel writes Aval[slot * sA + e]: every member SLOT of the launch — not only their number — must lie inside the reserved batch)
    for (int k = 0; direct && k < bs.b.n; ++k) if (bs.b.slot[k] < 0 || bs.b.slot[k] >= sparse_batch(s->spS)) direct = false;
    if (direct) sparse_values(s->spS, &Aval, &sA);
    const int* inv = direct ? s->spS_inv : nullptr;
    s->spS_values_current = direct;
    if (s->d.max_dim <= 32) hipLaunchKernelGGL(k_schur_blocks<32>, dim3(B.npairs, 1, bs.b.n), dim3(256), 0, s->stream, bs, s->d, s->cone, B.d_pairs, B.d_pairblk, B.d_seg, B.d_blk, B.d_lblk,
                                               s->Lsym, s->wz, s->Wsoc, s->S, s->compact ? 1 : 0, Aval, sA, inv);
    else hipLaunchKernelGGL(k_schur_blocks<64>, dim3(B.npairs, 1, bs.b.n), dim3(256), 0, s->stream, bs, s->d, s->cone, B.d_pairs, B.d_pairblk, B.d_seg, B.d_blk, B.d_lblk, s->Lsym, s->wz,
                            s->Wsoc, s->S, s->compact ? 1 : 0, Aval, sA, inv);
    return true;
}

}  // namespace calipso

namespace calipso {

// unpack the blocks into dense column-major matrices (structured handles only hold the blocks: the rare dense consumers — the pivoted LU fallback, the
// dense K for inspection — get temporaries)
__global__ __launch_bounds__(256) void k_blocks_unpack_z(const ZBlock* __restrict__ blk, int m, const double* __restrict__ pk, double* __restrict__ Z) {
    const ZBlock b = blk[blockIdx.x];
    const int total = b.nrows * b.ncols;
    for (int idx = threadIdx.x; idx < total; idx += 256) {
        const int i = idx % b.nrows, j = idx / b.nrows;
        Z[(size_t)(b.row0 + i) + (size_t)(b.col0 + j) * m] = pk[b.off_c + idx];
    }
}
__global__ __launch_bounds__(256) void k_blocks_unpack_l(const LBlock* __restrict__ blk, int nx, const double* __restrict__ pk, double* __restrict__ L) {
    const LBlock b = blk[blockIdx.x];
    const int total = b.n * b.n;
    for (int idx = threadIdx.x; idx < total; idx += 256) {
        const int i = idx % b.n, j = idx / b.n;
        L[(size_t)(b.c0 + i) + (size_t)(b.c0 + j) * nx] = pk[b.off_c + idx];
    }
}
// Lxx (nx x nx) and [gx; hx] (m x nx, ld m) of a structured handle as dense temporaries (zero outside the blocks); the caller frees them
int blocks_unpack_dense(calipso_hip_solver* s, double** Lxx, double** Z) {
    const Dims& d = s->d;
    const StageBlocks& B = s->blocks;
    *Lxx = nullptr; *Z = nullptr;
    CK(hipMalloc((void**)Lxx, sizeof(double) * (size_t)d.nx * d.nx));
    CK(hipMalloc((void**)Z, sizeof(double) * std::max<size_t>((size_t)d.m * d.nx, 1)));
    CK(hipMemsetAsync(*Lxx, 0, sizeof(double) * (size_t)d.nx * d.nx, s->stream));
    CK(hipMemsetAsync(*Z, 0, sizeof(double) * std::max<size_t>((size_t)d.m * d.nx, 1), s->stream));
    if (B.nlb) hipLaunchKernelGGL(k_blocks_unpack_l, dim3(B.nlb), dim3(256), 0, s->stream, B.d_lblk, d.nx, s->Lsym, *Lxx);
    if (B.nblk) hipLaunchKernelGGL(k_blocks_unpack_z, dim3(B.nblk), dim3(256), 0, s->stream, B.d_blk, d.m, s->Lsym, *Z);
    return CALIPSO_OK;
}

// The block tables of a structure: zrow = per row of [gx; hx] its [first, last + 1) column (0-based), lreach = per column of Lxx the last row/column its
// entries reach.  Pure host work (no device): structured handles size their slab from it before anything is allocated.
bool blocks_plan(const Dims& d, const std::vector<int>& zrow, const std::vector<int>& lreach, BlockPlan& P, std::string& err) {
    const int nx = d.nx, m = d.m, ne = d.ne;
    if ((int)zrow.size() != 2 * m || (int)lreach.size() != nx) { err = "calipso_hip_set_stage_blocks: call calipso_hip_analyze_structure first"; return false; }
    if (d.max_dim > 64) { err = "calipso_hip_set_stage_blocks: the block kernels take second-order cones up to dimension 64 (wider cones: the dense treatment)"; return false; }
    std::vector<LBlock>& lb = P.lb; std::vector<ZBlock>& zb = P.zb;
    lb.clear(); zb.clear();
    {   // Hessian blocks: a new block starts at column p when no entry of columns < p reaches p or beyond
        int start = 0, reach = -1;
        for (int j = 0; j < nx; ++j) {
            if (j > start && reach < j) { lb.push_back({start, j - start, 0, 0}); start = j; }
            reach = std::max(reach, std::max(j, lreach[(size_t)j]));
        }
        lb.push_back({start, nx - start, 0, 0});
    }
    for (int k = 0; k < m;) {    // Z blocks: runs of consecutive rows with one column range (never across the equality / cone boundary)
        const int lo = zrow[2 * (size_t)k], hi = zrow[2 * (size_t)k + 1];
        int e = k + 1;
        while (e < m && e != ne && zrow[2 * (size_t)e] == lo && zrow[2 * (size_t)e + 1] == hi) ++e;
        zb.push_back({k, e - k, hi > lo ? lo : 0, hi > lo ? hi - lo : 0, 0, 0});
        k = e;
    }
    if (lb.size() < 2 || zb.size() > (size_t)std::max(64, m / 2)) { err = "calipso_hip_set_stage_blocks: no block structure to exploit (one Hessian block, or a column range per row)"; return false; }
    size_t off = 0;
    P.max_lb = 0;
    for (ZBlock& b : zb) { const size_t n = (size_t)b.nrows * b.ncols; b.off_c = (long long)off; b.off_r = (long long)(off + n); off += 2 * n; }
    for (LBlock& b : lb) { const size_t n = (size_t)b.n * b.n; b.off_c = (long long)off; b.off_r = (long long)(off + n); off += 2 * n; P.max_lb = std::max(P.max_lb, b.n); }
    P.packed = off;
    // segments: every block boundary, at most 64 columns each
    std::vector<int> cut = {0, nx};
    for (const ZBlock& b : zb) if (b.ncols) { cut.push_back(b.col0); cut.push_back(b.col0 + b.ncols); }
    for (const LBlock& b : lb) { cut.push_back(b.c0); cut.push_back(b.c0 + b.n); }
    std::sort(cut.begin(), cut.end());
    cut.erase(std::unique(cut.begin(), cut.end()), cut.end());
    std::vector<Segment>& seg = P.seg; seg.clear(); P.segblk.clear();
    for (size_t c = 0; c + 1 < cut.size(); ++c)
        for (int a = cut[c]; a < cut[c + 1]; a += 64) seg.push_back({a, std::min(64, cut[c + 1] - a), 0, 0});
    P.seg_of_col.assign((size_t)nx, 0);
    for (size_t g = 0; g < seg.size(); ++g) {
        for (int c = seg[g].c0; c < seg[g].c0 + seg[g].nc; ++c) P.seg_of_col[(size_t)c] = (int)g;
        seg[g].first = (int)P.segblk.size();
        for (size_t q = 0; q < zb.size(); ++q) if (zb[q].ncols && zb[q].col0 <= seg[g].c0 && seg[g].c0 + seg[g].nc <= zb[q].col0 + zb[q].ncols) P.segblk.push_back((int)q);
        seg[g].count = (int)P.segblk.size() - seg[g].first;
    }
    // pairs of segments (a >= b) that a Z block or a Hessian block couples
    std::map<std::pair<int, int>, std::pair<std::vector<int>, int>> pm;
    for (size_t q = 0; q < zb.size(); ++q) {
        if (!zb[q].ncols) continue;
        const int g0 = P.seg_of_col[(size_t)zb[q].col0], g1 = P.seg_of_col[(size_t)(zb[q].col0 + zb[q].ncols - 1)];
        for (int a = g0; a <= g1; ++a) for (int b2 = g0; b2 <= a; ++b2) { auto it = pm.find({a, b2}); if (it == pm.end()) it = pm.insert({{a, b2}, {{}, -1}}).first; it->second.first.push_back((int)q); }
    }
    for (size_t q = 0; q < lb.size(); ++q) {
        const int g0 = P.seg_of_col[(size_t)lb[q].c0], g1 = P.seg_of_col[(size_t)(lb[q].c0 + lb[q].n - 1)];
        for (int a = g0; a <= g1; ++a) for (int b2 = g0; b2 <= a; ++b2) { auto it = pm.find({a, b2}); if (it == pm.end()) pm.insert({{a, b2}, {{}, (int)q}}); else it->second.second = (int)q; }
    }
    P.pairs.clear(); P.pairblk.clear();
    size_t soff = 0;
    for (auto& kv : pm) {
        SegPair p{kv.first.first, kv.first.second, (int)P.pairblk.size(), (int)kv.second.first.size(), kv.second.second, (long long)soff};
        soff += (size_t)seg[(size_t)p.a].nc * seg[(size_t)p.b].nc;
        P.pairblk.insert(P.pairblk.end(), kv.second.first.begin(), kv.second.first.end());
        P.pairs.push_back(p);
    }
    P.spacked = soff + 32;                        // (+ a cell that stays zero: what the skyline holds between the tiles points there)
    P.colrange.assign(2 * (size_t)nx, 0);
    for (const LBlock& b : lb) for (int c = b.c0; c < b.c0 + b.n; ++c) { P.colrange[2 * (size_t)c] = b.c0; P.colrange[2 * (size_t)c + 1] = b.c0 + b.n; }
    unsigned long long h = 1469598103934665603ULL;
    auto mix = [&](long long v) { h ^= (unsigned long long)v; h *= 1099511628211ULL; };
    for (const ZBlock& b : zb) { mix(b.row0); mix(b.nrows); mix(b.col0); mix(b.ncols); }
    for (const LBlock& b : lb) { mix(b.c0); mix(b.n); }
    P.signature = h;
    return true;
}

// the device side of a plan
int blocks_install(calipso_hip_solver* s, const BlockPlan& P) {
    StageBlocks& B = s->blocks;
    auto up = [&](const auto& vec, auto** dptr) -> hipError_t {
        typedef typename std::remove_const<typename std::remove_reference<decltype(vec)>::type>::type V;
        typedef typename V::value_type T;
        hipError_t e = hipMalloc((void**)dptr, sizeof(T) * std::max<size_t>(vec.size(), 1));
        if (e == hipSuccess && !vec.empty()) e = hipMemcpy(*dptr, vec.data(), sizeof(T) * vec.size(), hipMemcpyHostToDevice);
        return e;
    };
    hipError_t e = up(P.zb, &B.d_blk);
    if (e == hipSuccess) e = up(P.lb, &B.d_lblk);
    if (e == hipSuccess) e = up(P.seg, &B.d_seg);
    if (e == hipSuccess) e = up(P.segblk, &B.d_segblk);
    if (e == hipSuccess) e = up(P.pairs, &B.d_pairs);
    if (e == hipSuccess) e = up(P.pairblk, &B.d_pairblk);
    if (e == hipSuccess) e = up(P.colrange, &B.d_colrange);
    if (e != hipSuccess) { blocks_release(s); return calipso::check(s, e, "stage blocks: device tables"); }
    B.nblk = (int)P.zb.size(); B.nlb = (int)P.lb.size(); B.nseg = (int)P.seg.size(); B.npairs = (int)P.pairs.size(); B.max_lb = P.max_lb; B.packed = P.packed;
    B.signature = P.signature;
    B.h_blk = P.zb; B.h_lblk = P.lb; B.h_pairs = P.pairs; B.h_seg = P.seg; B.h_seg_of_col = P.seg_of_col;
    B.schur_flops = 0.0;
    for (const SegPair& pr : P.pairs)
        for (int q = 0; q < pr.count; ++q) B.schur_flops += 2.0 * (double)P.zb[(size_t)P.pairblk[(size_t)pr.first + q]].nrows * (double)P.seg[(size_t)pr.a].nc * (double)P.seg[(size_t)pr.b].nc;
    B.on = true;
    return CALIPSO_OK;
}

// ---- structured handles: the dense HOST arrays of the reference's ProblemData <-> the packed blocks ---------------------------------------------
// which: 0 = lagrangian_hessian (nx x nx), 1 = equality_jacobian_variables (ne x nx), 2 = cone_jacobian_variables (nc x nx), all column-major.
// Entries outside the declared structure must be zero (there is nowhere to put them): CALIPSO_ERR_ARGUMENT otherwise.
int blocks_upload_dense(calipso_hip_solver* s, int which, const double* data, double scale) {
    const Dims& d = s->d;
    const StageBlocks& B = s->blocks;
    std::vector<double>& hb = s->hstage;
    long long lo = -1, hi = -1;                 // the region of the packed buffer this field owns
    auto claim = [&](long long a, long long b) { if (lo < 0 || a < lo) lo = a; if (b > hi) hi = b; };
    if (which == 0) for (const LBlock& b : B.h_lblk) claim(b.off_c, b.off_r + (long long)b.n * b.n);
    else for (const ZBlock& b : B.h_blk) if ((which == 1) == (b.row0 < d.ne)) claim(b.off_c, b.off_r + (long long)b.nrows * b.ncols);
    if (lo < 0) return CALIPSO_OK;
    hb.assign((size_t)(hi - lo), 0.0);
    if (which == 0) {
        const size_t nx = (size_t)d.nx;
        size_t inside = 0, total = 0;
        for (size_t q = 0; q < nx * nx; ++q) total += data[q] != 0.0;
        for (const LBlock& b : B.h_lblk)
            for (int j = 0; j < b.n; ++j) for (int i = 0; i < b.n; ++i) {
                const double v = data[(size_t)(b.c0 + i) + (size_t)(b.c0 + j) * nx];
                inside += v != 0.0;
                hb[(size_t)(b.off_c - lo) + i + (size_t)j * b.n] = scale * v;
                hb[(size_t)(b.off_r - lo) + (size_t)i * b.n + j] = scale * v;
            }
        if (inside != total) { s->err = "lagrangian_hessian has non-zero entries outside the Hessian blocks declared at calipso_hip_create_structured"; return CALIPSO_ERR_ARGUMENT; }
    } else {
        const int r0 = which == 1 ? 0 : d.ne, rows = which == 1 ? d.ne : d.nc;
        size_t inside = 0, total = 0;
        for (size_t q = 0; q < (size_t)rows * d.nx; ++q) total += data[q] != 0.0;
        for (const ZBlock& b : B.h_blk) {
            if ((which == 1) != (b.row0 < d.ne)) continue;
            for (int j = 0; j < b.ncols; ++j) for (int i = 0; i < b.nrows; ++i) {
                const double v = data[(size_t)(b.row0 - r0 + i) + (size_t)(b.col0 + j) * rows];
                inside += v != 0.0;
                hb[(size_t)(b.off_c - lo) + i + (size_t)j * b.nrows] = scale * v;
                hb[(size_t)(b.off_r - lo) + (size_t)i * b.ncols + j] = scale * v;
            }
        }
        if (inside != total) { s->err = "a constraint Jacobian has non-zero entries outside the column ranges declared at calipso_hip_create_structured"; return CALIPSO_ERR_ARGUMENT; }
    }
    CK(hipMemcpyAsync(s->Lsym + lo, hb.data(), sizeof(double) * hb.size(), hipMemcpyHostToDevice, s->stream));
    CK(hipStreamSynchronize(s->stream));        // (hstage is reused)
    return CALIPSO_OK;
}
int blocks_download_dense(calipso_hip_solver* s, int which, double* data) {
    const Dims& d = s->d;
    const StageBlocks& B = s->blocks;
    std::vector<double> hb(B.packed);
    CK(hipMemcpyAsync(hb.data(), s->Lsym, sizeof(double) * B.packed, hipMemcpyDeviceToHost, s->stream));
    CK(hipStreamSynchronize(s->stream));
    if (which == 0) {
        std::fill(data, data + (size_t)d.nx * d.nx, 0.0);
        for (const LBlock& b : B.h_lblk) for (int j = 0; j < b.n; ++j) for (int i = 0; i < b.n; ++i) data[(size_t)(b.c0 + i) + (size_t)(b.c0 + j) * d.nx] = hb[(size_t)b.off_c + i + (size_t)j * b.n];
    } else {
        const int r0 = which == 1 ? 0 : d.ne, rows = which == 1 ? d.ne : d.nc;
        std::fill(data, data + (size_t)rows * d.nx, 0.0);
        for (const ZBlock& b : B.h_blk) {
            if ((which == 1) != (b.row0 < d.ne)) continue;
            for (int j = 0; j < b.ncols; ++j) for (int i = 0; i < b.nrows; ++i) data[(size_t)(b.row0 - r0 + i) + (size_t)(b.col0 + j) * rows] = hb[(size_t)b.off_c + i + (size_t)j * b.nrows];
        }
    }
    return CALIPSO_OK;
}
// packed offsets (column-major copy, row-major copy) of entry (row, col) — 0-based; row: of the stacked Jacobian (field 1 / 2) or of Lxx (field 0);
// false if the entry lies outside the structure
bool blocks_entry_offsets(const calipso_hip_solver* s, int which, int row, int col, long long* off_c, long long* off_r) {
    const StageBlocks& B = s->blocks;
    if (which == 0) {
        for (const LBlock& b : B.h_lblk) if (col >= b.c0 && col < b.c0 + b.n) {
            if (row < b.c0 || row >= b.c0 + b.n) return false;
            *off_c = b.off_c + (row - b.c0) + (long long)(col - b.c0) * b.n; *off_r = b.off_r + (long long)(row - b.c0) * b.n + (col - b.c0);
            return true;
        }
        return false;
    }
    for (const ZBlock& b : B.h_blk) if (row >= b.row0 && row < b.row0 + b.nrows) {
        if (col < b.col0 || col >= b.col0 + b.ncols) return false;
        *off_c = b.off_c + (row - b.row0) + (long long)(col - b.col0) * b.nrows; *off_r = b.off_r + (long long)(row - b.row0) * b.ncols + (col - b.col0);
        return true;
    }
    return false;
}

}  // namespace calipso

using namespace calipso;

extern "C" {

// After calipso_hip_analyze_structure: derive the blocks, pack the current contents of the dense buffers and route the mat-vecs and the Schur complement
// of this handle through them (on = 0: back to the dense-layout kernels).  Refused (CALIPSO_ERR_ARGUMENT, the handle stays as it was) when the
// structure has no blocks to speak of (fewer than two Hessian blocks) or the packed data would not fit the slab region it borrows.
// info (may be NULL) = [Z blocks, Hessian blocks, column segments, packed doubles per instance (both orientations)].
int32_t calipso_hip_set_stage_blocks(calipso_hip_solver* s, int32_t on, int64_t info[4]) {
    if (!s) return CALIPSO_ERR_ARGUMENT;
    if (s->compact) { s->err = "calipso_hip_set_stage_blocks: a structured handle always works on its blocks"; return on ? CALIPSO_OK : CALIPSO_ERR_ARGUMENT; }
    CK(hipSetDevice(s->device));
    CK(hipStreamSynchronize(s->stream));
    blocks_release(s);
    if (!on) { s->hessian_dirty = true; return CALIPSO_OK; }           // (Lsym was borrowed: the dense Schur kernel needs it rebuilt)
    const Dims& d = s->d;
    BlockPlan P;
    if (!blocks_plan(d, s->h_zrow, s->h_lreach, P, s->err)) return CALIPSO_ERR_ARGUMENT;
    if (P.packed > (size_t)d.nx * d.nx) { s->err = "calipso_hip_set_stage_blocks: the packed blocks exceed the slab region they borrow"; return CALIPSO_ERR_ARGUMENT; }
    const int rc = blocks_install(s, P);
    if (rc < 0) { s->hessian_dirty = true; return rc; }
    CK(hipMemsetAsync(s->S, 0, sizeof(double) * (size_t)d.NP * d.NP, s->stream));     // what no pair covers must read as zero
    launch_pad_identity(s);
    blocks_pack(s, true, true);
    CK(hipStreamSynchronize(s->stream));
    if (info) { info[0] = s->blocks.nblk; info[1] = s->blocks.nlb; info[2] = s->blocks.nseg; info[3] = (int64_t)P.packed; }
    return CALIPSO_OK;
}

}  // extern "C"
