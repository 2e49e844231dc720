// sparse.hip — sparse LDL^T on the device for the LinearSolver seam: what the reference does with its vendored QDLDL
// (src/solver/qdldl.jl:134-188 analyse, :400-589 factor, :330-351,592-640 solve) for a sparse symmetric quasi-definite matrix, WITHOUT the dense
// n x n storage of ldlsolver.hip: memory is O(nnz(L)).
//
//   analyse (host, once per pattern)   order (ordering.hip: natural / RCM / minimum degree / nested dissection / the caller's), P A P' as a
//                                      LOWER CSC with a gather map from the caller's nzval, elimination tree, the row-sorted pattern of L, its
//                                      row view (which columns k < j update column j), and the LEVELS of the tree: level(j) = 1 + max level of
//                                      j's children.  Every column that updates column j lies in j's subtree, hence on a lower level.
//   factor  (device)                   left-looking by levels: one launch per level, one workgroup per column.  Column j is gathered into a
//                                      dense accumulator in LDS (n <= 20 000; global scratch beyond), the updates  w -= L(:,k) D_k L(j,k)  are
//                                      applied in ascending k (a fixed order: no atomics, bit-reproducible), then D_j = w_j, L(:,j) = w / D_j.
//                                      Runs of levels that hold a single column (the separators of a nested dissection, a dense tail) are
//                                      merged into ONE single-workgroup launch that walks the chain.  No pivoting (quasi-definite: any symmetric
//                                      order has an LDL^T, qdldl.jl:134-143); inertia from the signs of D, an exact zero pivot as qdldl.jl:456,579.
//   solve   (device)                   L y = b by levels (one wavefront per row, pull form over the row view), x = L^-T (y ./ D) by levels in
//                                      reverse (one wavefront per column); all right-hand sides of a call in the same launches.
// QDLDL is up-looking (row by row); the factor is unique, the summation order differs, so L and D agree with the oracle's restatement to
// rounding (tests/test_gpu_sparse.py: 1e-11 relative), not bit for bit.
#include <algorithm>
#include <numeric>
#include <string>
#include <vector>

#include "internal.hpp"
#include "device_utils.hpp"
#include "pivot16.hpp"
#include "ldl_device.hpp"

using calipso::i64;

namespace {

constexpr int SP_THREADS = 256;
constexpr int SP_REC = 256;                // update records staged in LDS at a time
constexpr int SP_LDS_MAX_N = 19400;        // 155 200 B of the CU's 160 KiB LDS for the accumulator (4 KiB of record stage beside it)

struct SpDev {
    int n;
    const int* order;                      // columns sorted by (level, index)
    const int *Alp, *Ali, *Asrc;           // P A P' lower CSC (diagonal included), Asrc = index into the caller's nzval
    const double* Aval;                    // the caller's nzval on the device
    const int *Lp, *Li;                    // strictly lower pattern of L, rows ascending
    double *Lx, *D;
    const int *Rp, *Rk, *Rpos, *Rend;      // row view: for row j the columns k < j with L(j,k) != 0 (ascending), the slot of L(j,k), the end of column k
    double* work;                          // global accumulators (gridDim.x x n) when n > SP_LDS_MAX_N
};

// barrier that orders the accumulator only: LDS traffic when the accumulator is in LDS (outstanding global PREFETCHES must not be waited for),
// everything otherwise
template <bool LDSW>
__device__ __forceinline__ void acc_barrier() {
    if (LDSW) { __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_barrier(); }   // lgkmcnt(0)
    else __syncthreads();
}

template <bool LDSW>
__global__ __launch_bounds__(SP_THREADS) void k_sp_factor(SpDev d, int first, int ncols) {
    extern __shared__ __attribute__((aligned(16))) double wl[];
    __shared__ int rpos[SP_REC], rend[SP_REC];
    __shared__ double rf[SP_REC];
    double* w = LDSW ? wl : d.work + (size_t)blockIdx.x * d.n;
    const int tid = threadIdx.x;
    for (int c = blockIdx.x; c < ncols; c += gridDim.x) {
        const int j = d.order[first + c];
        const int l0 = d.Lp[j], l1 = d.Lp[j + 1];
        for (int p = l0 + tid; p < l1; p += SP_THREADS) w[d.Li[p]] = 0.0;
        if (tid == 0) w[j] = 0.0;
        __syncthreads();
        for (int p = d.Alp[j] + tid; p < d.Alp[j + 1]; p += SP_THREADS) w[d.Ali[p]] = d.Aval[d.Asrc[p]];
        const int r0 = d.Rp[j], r1 = d.Rp[j + 1];
        __syncthreads();
        // The updates of column j are a dependent chain (they all touch w_j), one barrier each.  What must NOT be on that chain is global-memory
        // latency: the records (column k, slot of L(j,k), end of column k) and the factors L(j,k) D_k of up to SP_REC updates are staged in LDS
        // with one parallel load, and the first 256 (row, value) pairs of update q + 4 travel while update q is applied.
        for (int qc = r0; qc < r1; qc += SP_REC) {
            const int m = min(SP_REC, r1 - qc);
            if (tid < m) {
                const int k = d.Rk[qc + tid], pos = d.Rpos[qc + tid];
                rpos[tid] = pos; rend[tid] = d.Rend[qc + tid];
                rf[tid] = d.Lx[pos] * d.D[k];
            }
            __syncthreads();
            int li[4]; double lx[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                li[u] = 0; lx[u] = 0.0;
                if (u < m) { const int p = rpos[u] + tid; if (p < rend[u]) { li[u] = d.Li[p]; lx[u] = d.Lx[p]; } }
            }
            for (int q = 0; q < m; q += 4) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (q + u < m) {                                                        // (wave-uniform)
                        const int pos = rpos[q + u], end = rend[q + u];
                        const double f = rf[q + u];
                        const int cli = li[u]; const double clx = lx[u];
                        if (q + u + 4 < m) { const int p = rpos[q + u + 4] + tid; if (p < rend[q + u + 4]) { li[u] = d.Li[p]; lx[u] = d.Lx[p]; } }
                        if (pos + tid < end) w[cli] -= clx * f;                             // rows >= j of column k (the slot of L(j,k) is the first of them)
                        for (int p = pos + tid + SP_THREADS; p < end; p += SP_THREADS) w[d.Li[p]] -= d.Lx[p] * f;
                        acc_barrier<LDSW>();
                    }
                }
            }
            __syncthreads();                                                                // the record stage is refilled
        }
        const double dj = w[j];
        const double dinv = 1.0 / dj;                                               // L = y * Dinv as qdldl.jl:560-566
        for (int p = l0 + tid; p < l1; p += SP_THREADS) d.Lx[p] = w[d.Li[p]] * dinv;
        if (tid == 0) d.D[j] = dj;
        __syncthreads();                                                            // a chain's next column reads what this one wrote
    }
}

// forward substitution, rows of one level (or a chain, sequentially): x_j -= sum_k L(j,k) x_k.  One wavefront per row; blockIdx.y = right-hand side.
__global__ __launch_bounds__(64) void k_sp_forward(SpDev d, int first, int ncols, double* __restrict__ X) {
    double* x = X + (size_t)blockIdx.y * d.n;
    const int lane = threadIdx.x;
    for (int c = blockIdx.x; c < ncols; c += gridDim.x) {
        const int j = d.order[first + c];
        double acc = 0.0;
        for (int q = d.Rp[j] + lane; q < d.Rp[j + 1]; q += 64) acc += d.Lx[d.Rpos[q]] * x[d.Rk[q]];
        acc = calipso::wave_sum(acc);
        if (lane == 0) x[j] -= acc;
        __syncthreads();
    }
}
// backward substitution with the diagonal scaling folded in: x_j = y_j / D_j - sum_i L(i,j) x_i, levels (and chains) in reverse
__global__ __launch_bounds__(64) void k_sp_backward(SpDev d, int first, int ncols, double* __restrict__ X) {
    double* x = X + (size_t)blockIdx.y * d.n;
    const int lane = threadIdx.x;
    for (int c = blockIdx.x; c < ncols; c += gridDim.x) {
        const int j = d.order[first + ncols - 1 - c];
        double acc = 0.0;
        for (int p = d.Lp[j] + lane; p < d.Lp[j + 1]; p += 64) acc += d.Lx[p] * x[d.Li[p]];
        acc = calipso::wave_sum(acc);
        if (lane == 0) x[j] = x[j] / d.D[j] - acc;
        __syncthreads();
    }
}
__global__ void k_sp_permute_in(const double* __restrict__ b, const int* __restrict__ perm, int n, double* __restrict__ x) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[(size_t)blockIdx.y * n + i] = b[(size_t)blockIdx.y * n + perm[i]];          // permute!(x, perm)  qdldl.jl:333
}
__global__ void k_sp_permute_out(const double* __restrict__ x, const int* __restrict__ perm, int n, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[(size_t)blockIdx.y * n + perm[i]] = x[(size_t)blockIdx.y * n + i];        // ipermute!(x, perm) qdldl.jl:349
}


// ---- multifrontal path (nested-dissection orders) ---------------------------------------------------------------------------------------------
// The pieces of the dissection (leaf pieces of <= 48 vertices, separators) are the SUPERNODES: node s owns the contiguous columns
// [first, first + c) and the rows R_s below them (r of them); its frontal matrix F_s (m = c + r) is assembled in the LDS of one workgroup from
// the entries of A in its columns and the update matrices U of its children (extend-add through precomputed relative indices, children in
// ascending order: fixed summation order), the first c columns are eliminated by a dense LDL^T in LDS (one barrier per pivot column), the
// m x c panel of L goes to global memory and U_s = F_s[c.., c..] to the node's slot of the update pool.  One launch per LEVEL of the node tree:
// for a T-stage trajectory problem that is ~log2 T launches for the whole factorisation — the stages are eliminated in parallel, level by level.
// One record per node in LAUNCH order and one per (node, child): what a workgroup needs to find its front comes with ONE load each instead of a chain of
// dependent table look-ups (order -> nfirst / ncols / nrows -> childptr -> children -> upd_off / rowptr ...: 0.7 us per hop on a cold launch).
struct MfNode { int s, f, c, r; int rowptr, chfirst, nch, alp0; int alp1, pad0, pad1, pad2; long long panel_off, upd_off, u_off, foff; int xa0, xa1; };
// pad0: first row record of a front factored by many workgroups (sparse_wide.hpp; -1: none), pad1: launch position of the parent (-1: a root), pad2: the level's ypan
struct MfChild { int rc, rowptr; long long upd_off, u_off; int pos, pad; };   // pos: the child's launch position
// one entry of a child's update matrix on its way into the parent's LDS front (MfNode::xa0 .. xa1, children in ascending order, a child's lower triangle row by row):
// usrc = its offset in the instance's update pool, tq = its place in the parent's packed front | the child's ordinal << 16
struct MfXItem { unsigned usrc, tq; };
struct MfRowItem { long long uoff; int relptr, a, uo, pad; };   // row a of a child's update matrix (offset in the update pool), the child's relative indices, the row's entry of the child's vector (solves)
struct MfDev {
    int nnodes;
    const MfNode* nrec;                       // [launch position]
    const MfChild* crec;                      // [chfirst + q], children in ascending order
    const int* order;                         // nodes sorted by level
    const int *nfirst, *ncols, *nrows;        // per node: first column, c, r
    const int *rowptr, *rows;                 // R_s (global, permuted row indices), ascending
    const int* rel;                           // per node, aligned with rows: local index of R_s[k] in the PARENT's front
    const int *childptr, *children;
    const long long *panel_off, *upd_off, *u_off;
    const int* Aloc;                          // per entry of the permuted lower CSC: offset inside its node's packed front (row (row + 1) / 2 + column)
    const int *Alp, *Asrc;
    const double* Aval;
    double *panel, *upd, *D, *uvec;
    double* fpool;                            // global-memory fronts (levels whose fronts exceed the LDS): packed lower triangles, per instance
    const long long* foff;                    // per node: offset of its front in fpool (levels reuse the pool)
    long long sPool;
    long long sA, sPanel, sUpd, sD;           // instance strides (batched factorisation of matrices with one pattern)
    // fronts factored by many workgroups (sparse_wide.hpp): per row of such a front (MfNode::pad0 + row) its entries of A and its child rows
    const int *wptrE, *wptrC;                 // [pad0 + row .. + 1]: ranges in wEcol / wEsrc and in wC
    const int *wEcol, *wEsrc;                 // local column in the front, index into Aval
    const MfRowItem* wC;                      // children in ascending order
    const MfXItem* xit;                       // extend-add items of the LDS fronts (nullptr: none)
    double* wscr;                             // X, M, L11, D of the diagonal blocks of one level: (matrix, node of the level)
};

// storage slot of launch instance z: z itself, or — for the members of a group's (shrinking) active set — slot[z].  A separate read-only kernel
// argument: indexing an array inside a struct the kernel also modifies would push the whole struct to scratch memory.
struct MfSlots { int use; int slot[calipso::MAX_BATCH]; };

typedef double calipso_v4d __attribute__((ext_vector_type(4)));
// threads per front: 256 for small fronts (several workgroups share a CU), 512 for fronts of more than MF_BIG rows (one front fills the CU's LDS
// anyway; more waves shorten the panel, assembly and matrix-core phases)
constexpr int MF_BIG = 96;
constexpr int MF_MAX_FRONT = 196;             // (m (m + 1) / 2 + 2 m) doubles <= 160 KiB: the front's lower triangle, packed, + the pivot-column buffers
constexpr int MF_PY = 18;                     // row stride of the exchange rows of an in-register panel (pivot16.hpp)
constexpr int MF_MAX_FRONT_WIDE = 8192;       // fronts factored and solved by many workgroups (sparse_wide.hpp); beyond: the column method
constexpr int MF_MAX_FRONT_GLOBAL = 4095;     // the ONE-workgroup kernels on fronts in global memory (calipso_hip_debug_wide_fronts(0): the A/B reference of
                                              // sparse_wide.hpp): same algorithm as in LDS, every access a memory access (a 1500-row front is ~2 ms); 4095 is where
                                              // the 24-bit index arithmetic of the packed triangle ends (tri0)

// The front is symmetric: only its lower triangle is held, packed row by row (row i starts at i (i + 1) / 2), which lets fronts of up to 196 rows
// fit the 160 KiB of LDS (a full square would stop at 141).
// i (i + 1) / 2 for a row index below 4096 by the full-rate 24-bit multiply (a 32-bit integer multiply issues at a quarter of the rate, and the trailing
// update of a front is bound by exactly this index arithmetic)
__device__ __forceinline__ int tri0(int i) { return (int)(__umul24((unsigned)i, (unsigned)(i + 1)) >> 1); }
__device__ __forceinline__ int tri(int i, int k) { return tri0(i) + k; }     // i >= k

// Optional timeline of one front per level (build with -DCALIPSO_LDL_TRACE: `make trace`; bench/mf_trace.py reads it through calipso_hip_debug_mf_trace):
// 100 MHz wall-clock stamps of workgroup (0, 0) of every k_mf_factor launch, keyed by the launch's `first` node index.
#ifdef CALIPSO_LDL_TRACE
__device__ long long g_mf_trace[64 * 12];
__device__ int g_mf_trace_n;
#define MF_STAMP(slot) do { if (mf_traced && threadIdx.x == 0) { g_mf_trace[(mf_tr & 63) * 12 + (slot)] = wall_clock64(); if ((slot) == 0) g_mf_trace[(mf_tr & 63) * 12 + 6] = __builtin_readcyclecounter(); if ((slot) == 5) g_mf_trace[(mf_tr & 63) * 12 + 7] = __builtin_readcyclecounter(); } } while (0)   // ([6], [7]: the core clock's counter at the first and the last stamp)
// ... and of workgroup (0, 0) of the sweeps' launches (bench/mf_solve_trace.py, calipso_hip_debug_mfs_trace): [0 .. 5] stamps, [6] 0 forward / 1 backward, [7] c | m << 16
__device__ long long g_mfs_trace[128 * 8];
__device__ int g_mfs_trace_n;
#define MFS_BEGIN(kind) __shared__ int mfs_tr_s; const bool mfs_on = blockIdx.x == 0 && blockIdx.y == 0; if (threadIdx.x == 0) mfs_tr_s = mfs_on ? (atomicAdd(&g_mfs_trace_n, 1) & 127) : 0; \
    __syncthreads(); const int mfs_tr = mfs_tr_s; if (mfs_on && threadIdx.x == 0) { g_mfs_trace[mfs_tr * 8 + 6] = (kind); g_mfs_trace[mfs_tr * 8 + 7] = nd.c | ((nd.c + nd.r) << 16); }
#define MFS_STAMP(slot) do { if (mfs_on && threadIdx.x == 0) g_mfs_trace[mfs_tr * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define MF_STAMP(slot) do { } while (0)
#define MFS_STAMP(slot) do { } while (0)
#endif

// The trailing update of one panel (columns kb .. pe - 1, pe - kb = 4 NK) by one wavefront: its 16 x 16 tiles of the lower triangle beyond pe, row-major, every NW-th.
// F[i][j] -= sum_k (y_ik / d_k) y_jk on the fp64 matrix cores (first operand = scaled rows of the i tile, second = rows of the j tile; two accumulator chains per tile:
// a dependent v_mfma_f64_16x16x4 issues every 64 cycles).  (bi, bj) advance on scalar registers; a tile needs ONE triangular index per operand and one for its
// results (the others follow by i -> i + 4: + 4 i + 10); everything up to the stores is branch-free — rows past the front are clamped for the loads and masked for the stores.  (Requesting the NEXT tile's operands ahead of the
// current tile's matrix instructions was measured: nothing.)
struct MfTile { double af[4], bf[4], old[4]; int o[4]; bool s[4]; };
// LDSF: the front lives in LDS.  Its operand reads are explicit ds_read_b64 then: left to the compiler, the four k-steps of an operand become ds_read2_b64 pairs — half the
// rate and other lane groups (16 contiguous lanes = 16 different rows of the packed triangle at one k: bank conflicts), which is what the update phase was waiting for
template <int NK, bool LDSF>
__device__ __forceinline__ void mf_tile_load(MfTile& T, const double* __restrict__ F, int m, int kb, int pe, int bi, int bj, int fr, int fk) {
    const int i0 = pe + 16 * bi + fk, j = pe + 16 * bj + fr;
    const double* Fa = F + tri0(min(pe + 16 * bi + fr, m - 1)) + kb + fk;
    const double* Fb = F + tri0(min(j, m - 1)) + kb + fk;
    int t = tri0(i0);
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int i = i0 + 4 * rr;
        T.s[rr] = j <= i && i < m;
        T.o[rr] = T.s[rr] ? t + j : 0;
        t += 4 * i + 10;
    }
    if constexpr (LDSF) {
        const unsigned ea = (unsigned)(uintptr_t)Fa, eb = (unsigned)(uintptr_t)Fb;
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(T.af[kk]) : "v"(ea), "n"(32 * kk) : "memory");
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(T.bf[kk]) : "v"(eb), "n"(32 * kk) : "memory");
        }
    } else {
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) { T.af[kk] = Fa[4 * kk]; T.bf[kk] = Fb[4 * kk]; }
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) T.old[rr] = F[T.o[rr]];
}
// (what the explicit reads of a tile return is there after this: the compiler does not count them)
template <int NK> __device__ __forceinline__ void mf_tile_wait(MfTile& T) {
    if constexpr (NK == 4) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(T.af[0]), "+v"(T.af[1]), "+v"(T.af[2]), "+v"(T.af[3]), "+v"(T.bf[0]), "+v"(T.bf[1]), "+v"(T.bf[2]), "+v"(T.bf[3]) :: "memory");
    else if constexpr (NK == 3) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(T.af[0]), "+v"(T.af[1]), "+v"(T.af[2]), "+v"(T.bf[0]), "+v"(T.bf[1]), "+v"(T.bf[2]) :: "memory");
    else if constexpr (NK == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(T.af[0]), "+v"(T.af[1]), "+v"(T.bf[0]), "+v"(T.bf[1]) :: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(T.af[0]), "+v"(T.bf[0]) :: "memory");
}
template <int NK, int NW, bool LDSF>
__device__ __forceinline__ void mf_update_tiles(double* __restrict__ F, const double (&rfh)[4], int m, int kb, int pe, int wv, int ntile, int fr, int fk) {
    if (wv >= ntile) return;
    int bi = 0, bj = wv;
    while (bj > bi) { bj -= bi + 1; ++bi; }
    for (int t = wv; t < ntile; t += NW) {
        MfTile A;
        mf_tile_load<NK, LDSF>(A, F, m, kb, pe, bi, bj, fr, fk);
        bj += NW;
        while (bj > bi) { bj -= bi + 1; ++bi; }
        if constexpr (LDSF) mf_tile_wait<NK>(A);
        calipso_v4d acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A.af[0] * rfh[0], A.bf[0], acc, 0, 0, 0);
        if constexpr (NK > 1) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(A.af[1] * rfh[1], A.bf[1], acc2, 0, 0, 0);
        if constexpr (NK > 2) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(A.af[2] * rfh[2], A.bf[2], acc, 0, 0, 0);
        if constexpr (NK > 3) acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(A.af[3] * rfh[3], A.bf[3], acc2, 0, 0, 0);
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) if (A.s[rr]) F[A.o[rr]] = A.old[rr] - (acc[rr] + acc2[rr]);
    }
}
// the barriers of the factorisation loop: a front in LDS needs the LDS traffic ordered, not the global stores in flight (the panel of L leaves column block by
// column block under the loop, and __syncthreads() would wait for every store at every barrier); a front in global memory needs the full one
template <bool GF> __device__ __forceinline__ void mf_barrier() {
    if constexpr (GF) __syncthreads();
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// one front: assembly, the partial LDL^T of its first c columns, the panel and the update matrix to global memory (every thread of the workgroup arrives)
template <int MF_THREADS, bool GF>
__device__ __forceinline__ void mf_factor_node(const MfDev& d, const MfNode& nd, const size_t z, const int ypan, double* __restrict__ Flds, const bool mf_traced, const int mf_tr) {
    MF_STAMP(0);
    constexpr int MF_RC = MF_THREADS / 16;        // row classes of the panel step (16 panel columns x MF_RC rows at a time)
    __shared__ int relS[256];                                                  // relative indices of the child being extend-added (LDS fronts: r <= 196)
    __shared__ double rinvS[64];                                               // GF: reciprocal pivots (a node has at most 64 columns)
    const int f = nd.f, c = nd.c, r = nd.r, m = c + r, nt = tri0(m);
    const int tid = threadIdx.x;
    double* F = GF ? d.fpool + z * d.sPool + nd.foff : Flds;                    // GF: the front lives in global memory (L2-resident)
    double* ycol = GF ? rinvS : F + nt;
    const double* Aval = d.Aval + z * d.sA;
    double* upd = d.upd + z * d.sUpd; double* panel = d.panel + z * d.sPanel; double* Dg = d.D + z * d.sD;
    // Assembly.  LDS fronts whose node carries extend-add items (MfXItem, built with the pattern): EVERY global load of the assembly is issued before anything waits —
    // the node's entries of A (Aloc / Asrc, then the values they point to) and the children's update matrices (items, then the pool entries they name): two memory
    // round trips for the whole front, the zero fill of the front under the first.  (Before: a round trip per 16 rows of a child, two for the own entries, one after
    // the other: 10 of a front's 30 us.)  Same operations on every entry in the same order (children ascending; a child's entries land on distinct places), so the
    // assembled front has the same bits.
    bool fast = false;
    if constexpr (!GF) fast = nd.xa0 >= 0;
    if (fast) {
        constexpr int AK = 8, XK = 8;
        const unsigned* __restrict__ xit = reinterpret_cast<const unsigned*>(d.xit);
        int al[AK], as[AK];
        uint2 it[XK];
        const int alast = max(nd.alp1 - 1, 0), xlast = max(nd.xa1 - 1, 0);
#pragma unroll
        for (int u = 0; u < AK; ++u) { const int p = min(nd.alp0 + tid + u * MF_THREADS, alast); al[u] = d.Aloc[p]; as[u] = d.Asrc[p]; }
#pragma unroll
        for (int u = 0; u < XK; ++u) { const int e = min(nd.xa0 + tid + u * MF_THREADS, xlast); it[u] = reinterpret_cast<const uint2*>(xit)[e]; }
        unsigned qw0 = xit[2 * (size_t)min(nd.xa0, xlast) + 1], qw1 = xit[2 * (size_t)min(nd.xa0 + XK * MF_THREADS - 1, xlast) + 1];   // first / last child of the chunk
        for (int e = tid; e < nt; e += MF_THREADS) F[e] = 0.0;
        double av[AK], uv[XK];
#pragma unroll
        for (int u = 0; u < AK; ++u) av[u] = Aval[as[u]];
#pragma unroll
        for (int u = 0; u < XK; ++u) uv[u] = upd[it[u].x];
        __syncthreads();
        MF_STAMP(1);
#pragma unroll
        for (int u = 0; u < AK; ++u) if (nd.alp0 + tid + u * MF_THREADS < nd.alp1) F[al[u]] = av[u];
        for (int p = nd.alp0 + AK * MF_THREADS + tid; p < nd.alp1; p += MF_THREADS) F[d.Aloc[p]] = Aval[d.Asrc[p]];
        __syncthreads();
        MF_STAMP(2);
        for (int cb = nd.xa0; cb < nd.xa1; cb += XK * MF_THREADS) {
            if (cb > nd.xa0) {
#pragma unroll
                for (int u = 0; u < XK; ++u) { const int e = min(cb + tid + u * MF_THREADS, xlast); it[u] = reinterpret_cast<const uint2*>(xit)[e]; }
#pragma unroll
                for (int u = 0; u < XK; ++u) uv[u] = upd[it[u].x];
                qw0 = xit[2 * (size_t)cb + 1]; qw1 = xit[2 * (size_t)min(cb + XK * MF_THREADS - 1, xlast) + 1];
            }
            // the children present in this chunk, in turn (a barrier between two children: both may add to one place, and the order of the sum is the children's)
            const int qlo = (int)(qw0 >> 16), qhi = (int)(qw1 >> 16);
            for (int q = qlo; q <= qhi; ++q) {
                double cur[XK];                                                   // (all reads of the round before its writes: the places of ONE child are distinct)
#pragma unroll
                for (int u = 0; u < XK; ++u) cur[u] = F[it[u].y & 0xffffu];
#pragma unroll
                for (int u = 0; u < XK; ++u)
                    if (cb + tid + u * MF_THREADS < nd.xa1 && (int)(it[u].y >> 16) == q) F[it[u].y & 0xffffu] = cur[u] + uv[u];
                __syncthreads();
            }
        }
    } else {
        for (int e = tid; e < nt; e += MF_THREADS) F[e] = 0.0;
        __syncthreads();
        MF_STAMP(1);
        for (int p = nd.alp0 + tid; p < nd.alp1; p += MF_THREADS) F[d.Aloc[p]] = Aval[d.Asrc[p]];
        __syncthreads();
        MF_STAMP(2);
        for (int q = 0; q < nd.nch; ++q) {                                         // extend-add, children in ascending order
            const MfChild cr = d.crec[nd.chfirst + q];
            const int rc = cr.rc;
            const double* U = upd + cr.upd_off;
            const int* rel = d.rel + cr.rowptr;
            if (!GF) { for (int a = tid; a < rc; a += MF_THREADS) relS[a] = rel[a]; }
            __syncthreads();
            if (GF) {
                for (int a = tid >> 5; a < rc; a += MF_THREADS / 32) {             // a row of the child's update matrix per 32 lanes, coalesced along b
                    const int rla = rel[a];
                    const int ra = tri0(rla);                            // rel is increasing: the lower triangle lands in the lower triangle
                    const double* Ua = U + (size_t)a * rc;
                    for (int b = tid & 31; b <= a; b += 32) F[ra + rel[b]] += Ua[b];
                }
            } else {
                // a row of the child's update matrix per 32 lanes (rc <= 196: at most seven chunks of 32 columns); the values of the NEXT row of this lane group
                // travel while the current one is added into the front (one memory round trip per row otherwise: 7 rows x 0.8 us per child)
                const int lb = tid & 31;
                int a = tid >> 5;
                double uv[7];
    #pragma unroll
                for (int q = 0; q < 7; ++q) { const int b = lb + 32 * q; uv[q] = (a < rc && b <= a) ? U[(size_t)a * rc + b] : 0.0; }
                while (a < rc) {
                    const int an = a + MF_THREADS / 32;
                    double un[7];
    #pragma unroll
                    for (int q = 0; q < 7; ++q) { const int b = lb + 32 * q; un[q] = (an < rc && b <= an) ? U[(size_t)an * rc + b] : 0.0; }
                    const int rla = relS[a];
                    const int ra = tri0(rla);                            // rel is increasing: the lower triangle lands in the lower triangle
    #pragma unroll
                    for (int q = 0; q < 7; ++q) { const int b = lb + 32 * q; if (b <= a) F[ra + relS[b]] += uv[q]; }
    #pragma unroll
                    for (int q = 0; q < 7; ++q) uv[q] = un[q];
                    a = an;
                }
            }
            __syncthreads();
        }
    }
    // Partial dense LDL^T of the first c columns, blocked: panels of 16 columns.
    //   panel   column j is left UNSCALED in F (y_ij = l_ij d_j; nothing overwrites what the other rows still read, so ONE barrier per column);
    //           256 threads as 16 row classes x 16 panel columns apply  F[i][k] -= y_ij (y_kj / d_j)  to the panel columns k > j, rows i >= k.
    //   update  F[i][j] -= sum_k (y_ik / d_k) y_jk over the panel, for every 16 x 16 tile of the trailing lower triangle on the fp64 matrix cores
    //           (v_mfma_f64_16x16x4: first operand = scaled rows of the i tile, second = rows of the j tile, K = 16 = one panel).
    double* rinv = ycol;                                                       // c reciprocal pivots (the 2 m doubles behind the front)
    double* P = panel + nd.panel_off;                                          // the node's panel of L, column-major m x c: column k contiguous over the rows
    int p_done = 0;                                                            // columns of it already written
    const int lane = tid & 63, wave = tid >> 6, fr = lane & 15, fk = lane >> 4;
    MF_STAMP(3);
#ifdef CALIPSO_LDL_TRACE
    long long mf_pan = 0, mf_upd = 0, mf_t0 = 0;
#endif
    for (int kb = 0; kb < c; kb += 16) {
        const int pe = min(kb + 16, c);
#ifdef CALIPSO_LDL_TRACE
        mf_t0 = wall_clock64();
#endif
        if (!GF && ypan && (pe - kb == 16 || pe - kb == 8) && m >= 32) {
            // A full panel through registers (pivot16.hpp; ldl.hip: diag_block has the design notes): wavefront 0 takes rows kb .. kb + 63 with lane = row and
            // factors the 16 columns alone — no barrier between pivots.  The rows further down go to the wavefronts behind it AT THE SAME TIME: wavefront e >= 1 puts the
            // 16 rows of the diagonal block in its lanes 0 .. 15 and rows kb + 64 + 48 (e - 1) ... in the other 48, and runs the same instruction sequence — the diagonal
            // block is factored redundantly, to the same bits, so nobody waits for wavefront 0's pivots (m <= 196: at most three such wavefronts; they share the exchange
            // rows: every writer of a diagonal row writes the same value, the other rows are never read).  Two barriers per panel instead of sixteen; the columns stay
            // UNSCALED in the front, as the loop below leaves them.  (Before: the rows beyond the first 64 FOLLOWED in a phase of their own, 0.9 us and a barrier per panel.)
            double* Yp = F + nt + 2 * m;                                       // 64 exchange rows of MF_PY doubles (the pivot columns, read back replicated)
            const bool half = pe - kb == 8;                                    // (a node of 56 columns ends in a half panel)
            mf_barrier<GF>();
            const int xrow0 = kb + 64 + 48 * (wave - 1);
            if (wave == 0 || xrow0 < m) {
                const int row = (wave == 0 || lane < 16) ? kb + lane : xrow0 + lane - 16;
                const bool in = row < m, mine = wave == 0 || lane >= 16;
                const int rowc = min(row, m - 1);
                const int trow = tri0(rowc) + kb;                   // (loads unconditional from a valid address, then selected: no exec-masked blocks)
                const int below = rowc - kb;                        // columns 0 .. min(below, 15) of the panel exist in this row
                double a[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) a[q] = F[trow + min(q, below)];
#pragma unroll
                for (int q = 0; q < 16; ++q) a[q] = (in && below >= q && (q < 8 || !half)) ? a[q] : 0.0;
                const int drow = min(kb + (lane & 15), m - 1);
                const double y0 = F[tri0(drow) + kb];
                const int lo = __builtin_amdgcn_readlane(__double2loint(a[0]), 0), hi = __builtin_amdgcn_readlane(__double2hiint(a[0]), 0);
                if (half) calipso::Pivot<0, false, 8>::run(a, (unsigned)(uintptr_t)(Yp + lane * MF_PY), (unsigned)(uintptr_t)(Yp + (lane & 15) * MF_PY), nullptr, 0,
                                                           calipso::fast_rcp(__hiloint2double(hi, lo)), y0);
                else calipso::Pivot<0, false, 16>::run(a, (unsigned)(uintptr_t)(Yp + lane * MF_PY), (unsigned)(uintptr_t)(Yp + (lane & 15) * MF_PY), nullptr, 0,
                                                       calipso::fast_rcp(__hiloint2double(hi, lo)), y0);
                if (in && mine) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) if (below >= q && (q < 8 || !half)) F[trow + q] = a[q];
                }
                if (wave == 0 && lane < (half ? 8 : 16)) {
                    double dd = a[0];
#pragma unroll
                    for (int q = 1; q < 16; ++q) dd = (lane == q) ? a[q] : dd;
                    rinv[kb + lane] = calipso::fast_rcp(dd);
                    Dg[f + kb + lane] = dd;
                }
            } else {
                // the wavefronts without rows of this panel send the columns of the panels BEFORE it to the panel of L (they are final, and this step does not touch
                // them): most of the write-out runs under the pivots instead of in the tail of the kernel
                const int first_idle = 1 + max(0, (m - kb - 64 + 47) / 48), nidle = MF_THREADS / 64 - first_idle;
                for (int k = p_done + (wave - first_idle); k < kb; k += nidle) {
                    const double rk = rinv[k];
                    for (int i = lane; i < m; i += 64) P[i + (size_t)k * m] = i > k ? F[tri0(i) + k] * rk : 0.0;
                }
            }
            if (1 + max(0, (m - kb - 64 + 47) / 48) < MF_THREADS / 64) p_done = kb;
        } else
        for (int j = kb; j < pe; ++j) {
            mf_barrier<GF>();
            const double dj = F[tri(j, j)];
            const double rj = 1.0 / dj;
            if (tid == 0) { Dg[f + j] = dj; rinv[j] = rj; }
            const int k = j + 1 + (tid & 15);                                  // MF_RC x 16 threads over (row, panel column)
            if (k < pe) {
                const double ykj = F[tri(k, j)] * rj;
                int i = j + 1 + (tid >> 4);                                      // rows i >= k of this thread's residue class
                if (i < k) i += ((k - i + MF_RC - 1) / MF_RC) * MF_RC;
                for (; i < m; i += MF_RC) {
                    double* Fi = F + tri0(i);
                    Fi[k] -= Fi[j] * ykj;
                }
            }
        }
        mf_barrier<GF>();
#ifdef CALIPSO_LDL_TRACE
        { const long long t1 = wall_clock64(); mf_pan += t1 - mf_t0; mf_t0 = t1; }
#endif
        const int ntl = (m - pe + 15) / 16;                                    // 16-row tiles of the trailing part
        // the tiles of the lower triangle (bi >= bj), row-major, dealt round-robin to the wavefronts: every wavefront gets the same number of tiles to within
        // one (whole tile rows per wavefront left the one with the longest rows 1.7 x the average).  Two accumulator chains per tile (a dependent
        // v_mfma_f64_16x16x4 issues every 64 cycles).
        const int ntile = tri0(ntl);
        const int wv = __builtin_amdgcn_readfirstlane(wave);
        const int nk4 = ((pe - kb) & 3) == 0 ? (pe - kb) >> 2 : 0;            // k-steps of the panel when every lane of a step is inside it (0: the general body below)
        if (nk4) {
            double rfh[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) rfh[kk] = rinv[min(kb + 4 * kk + fk, pe - 1)];
            if (nk4 == 4) mf_update_tiles<4, MF_THREADS / 64, !GF>(F, rfh, m, kb, pe, wv, ntile, fr, fk);
            else if (nk4 == 2) mf_update_tiles<2, MF_THREADS / 64, !GF>(F, rfh, m, kb, pe, wv, ntile, fr, fk);
            else if (nk4 == 3) mf_update_tiles<3, MF_THREADS / 64, !GF>(F, rfh, m, kb, pe, wv, ntile, fr, fk);
            else mf_update_tiles<1, MF_THREADS / 64, !GF>(F, rfh, m, kb, pe, wv, ntile, fr, fk);
        } else
        for (int t = wave; t < ntile; t += MF_THREADS / 64) {
            // panels whose width is not a multiple of four: every read is unconditional, from a clamped address, the k steps outside the panel are zeroed
            int bi = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
            while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
            while (tri0(bi) > t) --bi;
            const int bj = t - tri0(bi);
            const int ia = pe + 16 * bi + fr;                                  // the row this lane feeds to the first operand
            const int jbr = pe + 16 * bj + fr;                                  // ... to the second operand
            // every LDS read below is unconditional, from a clamped (always valid) address, and all of them are issued before the first use: a load inside a
            // conditional becomes an exec-masked block with its own wait (measured: 2400 cycles per tile that way)
            const int iac = min(ia, m - 1), jbc = min(jbr, m - 1);
            const int ra = tri0(iac), rb = tri0(jbc);
            double af[4], bf[4], rf[4], old[4];
            int oidx[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int kc = min(kb + 4 * kk + fk, pe - 1);
                af[kk] = F[ra + kc]; bf[kk] = F[rb + kc]; rf[kk] = rinv[kc];
            }
            const int j = pe + 16 * bj + fr;                                    // result column (second operand's tile)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int i = pe + 16 * bi + fk + 4 * rr;                       // result row (first operand's tile)
                const int ic = min(i, m - 1), jc = min(j, ic);
                oidx[rr] = tri0(ic) + jc;
                old[rr] = F[oidx[rr]];
            }
            double av[4], bv[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const bool kin = kb + 4 * kk + fk < pe;
                av[kk] = (kin && ia < m) ? af[kk] * rf[kk] : 0.0;
                bv[kk] = (kin && jbr < m) ? bf[kk] : 0.0;
            }
            calipso_v4d acc = {0.0, 0.0, 0.0, 0.0}, acc2 = {0.0, 0.0, 0.0, 0.0};
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[0], bv[0], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[1], bv[1], acc2, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[2], bv[2], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[3], bv[3], acc2, 0, 0, 0);
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int i = pe + 16 * bi + fk + 4 * rr;
                if (i < m && j <= i) F[oidx[rr]] = old[rr] - (acc[rr] + acc2[rr]);
            }
        }
#ifdef CALIPSO_LDL_TRACE
        mf_barrier<GF>();
        mf_upd += wall_clock64() - mf_t0;
#endif
    }
    mf_barrier<GF>();
    MF_STAMP(4);
#ifdef CALIPSO_LDL_TRACE
    if (mf_traced && tid == 0) { g_mf_trace[(mf_tr & 63) * 12 + 8] = mf_pan; g_mf_trace[(mf_tr & 63) * 12 + 9] = mf_upd; }
#endif
    // write-out: what is left of the panel (a wavefront per column, lanes along the rows: contiguous stores) and the update matrix (a wavefront per row, lanes along
    // the columns)
    for (int k = p_done + wave; k < c; k += MF_THREADS / 64) {
        const double rk = rinv[k];
        for (int i = lane; i < m; i += 64) P[i + (size_t)k * m] = i > k ? F[tri0(i) + k] * rk : 0.0;
    }
    double* U = upd + nd.upd_off;
    for (int a = wave; a < r; a += MF_THREADS / 64) {
        const double* Fa = F + tri0(c + a) + c;
        for (int b = lane; b <= a; b += 64) U[(size_t)a * r + b] = Fa[b];
    }
    MF_STAMP(5);
}

template <int MF_THREADS, bool GF>
__global__ __launch_bounds__(MF_THREADS) void k_mf_factor(const MfDev d, const MfSlots sl, int first, int ypan) {
    extern __shared__ __attribute__((aligned(16))) double Flds[];
    bool mf_traced = false; int mf_tr = 0;
#ifdef CALIPSO_LDL_TRACE
    __shared__ int mf_tr_s;
    mf_traced = blockIdx.x == 0 && blockIdx.y == 0;
    if (threadIdx.x == 0) mf_tr_s = mf_traced ? atomicAdd(&g_mf_trace_n, 1) : 0;
    __syncthreads();
    mf_tr = mf_tr_s;
    if (mf_traced && threadIdx.x == 0) { g_mf_trace[(mf_tr & 63) * 12 + 10] = d.nrec[first].c; g_mf_trace[(mf_tr & 63) * 12 + 11] = d.nrec[first].c + d.nrec[first].r; }
#endif
    const MfNode nd = d.nrec[first + blockIdx.x];
    const size_t z = sl.use ? (size_t)sl.slot[blockIdx.y] : (size_t)blockIdx.y;   // storage slot of this instance of the batch
    mf_factor_node<MF_THREADS, GF>(d, nd, z, ypan, Flds, mf_traced, mf_tr);
}

// forward: v = [b_C ; 0] + children's contributions;  y_C = L11^-1 v_C;  v_R -= L21 y_C  -> the node's contribution to its ancestors.
// The c <= 64 dependent steps of the triangular solve run in ONE wavefront (lane = row, y_k by v_readlane, no barrier: ~20 cycles per step instead of a
// workgroup barrier); the product with L21 is spread over all threads (four k-slices per row, combined in a fixed order).  The panel is read where it
// lies (every entry is used once): the LDS holds only the vector.  (GP, the former "panel too large for the LDS" variant, is the same code now.)
__device__ __forceinline__ double mf_readlane_d(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}
// (one node and right-hand side: yi = instance * nrhs + right-hand side, zs = the instance's storage slot; every thread of the workgroup arrives)
template <int MF_THREADS>
__device__ __forceinline__ void mf_forward_node(const MfDev& d, const MfNode& nd, const size_t yi, const size_t zs, int n, long long usum, double* __restrict__ X,
                                                double* __restrict__ sm, const bool mfs_on = false, const int mfs_tr = 0) {
    MFS_STAMP(0);
    const int f = nd.f, c = nd.c, r = nd.r, m = c + r;
    double* v = sm;                                                            // m
    double* part = sm + m;                                                     // 4 r
    double* x = X + yi * n;
    double* ubase = d.uvec + yi * usum;
    const double* panel = d.panel + zs * d.sPanel;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double* P = panel + nd.panel_off;
    // Every entry of the panel this thread will need is requested FIRST (the panel does not depend on the vector): the lower triangle of L11 by wavefront 0
    // (two batches of 32 columns), the first slice of L21 by everybody; the assembly of the vector runs while they travel.
    const int cs = (c + 3) / 4;                                                // columns per slice of the product with L21 (c <= 64: at most 16)
    double pl0[32], pl1[32], pv0[16];
    if (wave == 0) {
#pragma unroll
        for (int q = 0; q < 32; ++q) { pl0[q] = (q < c && lane > q && lane < c) ? P[lane + (size_t)q * m] : 0.0; }
#pragma unroll
        for (int q = 0; q < 32; ++q) { const int k = 32 + q; pl1[q] = (k < c && lane > k && lane < c) ? P[lane + (size_t)k * m] : 0.0; }
    }
    {
        const int idx = tid, a = r > 0 ? idx % r : 0, q4 = r > 0 ? idx / r : 0, kbeg = q4 * cs, kend = min(c, kbeg + cs);
        const double* Pi = P + (c + a);
#pragma unroll
        for (int q = 0; q < 16; ++q) pv0[q] = (idx < 4 * r && kbeg + q < kend) ? Pi[(size_t)(kbeg + q) * m] : 0.0;
    }
    for (int i = tid; i < m; i += MF_THREADS) v[i] = i < c ? x[f + i] : 0.0;
    __syncthreads();
    MFS_STAMP(1);
    // the children's contributions, two children at a time: both records, then both children's values and places travel together (one memory round trip per PAIR
    // after the records instead of one per child); the sums stay in the children's order (a barrier between the two)
    for (int q = 0; q < nd.nch; q += 2) {
        const MfChild cr0 = d.crec[nd.chfirst + q];
        const MfChild cr1 = d.crec[nd.chfirst + min(q + 1, nd.nch - 1)];
        const int rc0 = cr0.rc, rc1 = q + 1 < nd.nch ? cr1.rc : 0;
        const double* u0 = ubase + cr0.u_off; const double* u1 = ubase + cr1.u_off;
        const int* rel0 = d.rel + cr0.rowptr; const int* rel1 = d.rel + cr1.rowptr;
        const int a0 = min(tid, max(rc0 - 1, 0)), a1 = min(tid, max(rc1 - 1, 0));
        const double w0 = u0[a0], w1 = rc1 ? u1[a1] : 0.0;
        const int p0 = rel0[a0], p1 = rc1 ? rel1[a1] : 0;
        if (tid < rc0) v[p0] += w0;
        for (int a = tid + MF_THREADS; a < rc0; a += MF_THREADS) v[rel0[a]] += u0[a];
        __syncthreads();
        if (rc1) {
            if (tid < rc1) v[p1] += w1;
            for (int a = tid + MF_THREADS; a < rc1; a += MF_THREADS) v[rel1[a]] += u1[a];
            __syncthreads();
        }
    }
    MFS_STAMP(2);
    if (wave == 0) {
        double vi = lane < c ? v[lane] : 0.0;
#pragma unroll
        for (int q = 0; q < 32; ++q) { if (q < c) vi = fma(-pl0[q], mf_readlane_d(vi, q), vi); }
#pragma unroll
        for (int q = 0; q < 32; ++q) { const int k = 32 + q; if (k < c) vi = fma(-pl1[q], mf_readlane_d(vi, k), vi); }
        if (lane < c) { v[lane] = vi; x[f + lane] = vi; }
    }
    __syncthreads();
    MFS_STAMP(3);
    for (int idx = tid; idx < 4 * r; idx += MF_THREADS) {
        const int a = idx % r, q4 = idx / r, kbeg = q4 * cs, kend = min(c, kbeg + cs);
        const double* Pi = P + (c + a);
        double acc = 0.0;
        if (idx == tid) {                                                       // the slice fetched at the top
#pragma unroll
            for (int q = 0; q < 16; ++q) acc += pv0[q] * (kbeg + q < kend ? v[kbeg + q] : 0.0);
        } else {
            double pv[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) pv[q] = kbeg + q < kend ? Pi[(size_t)(kbeg + q) * m] : 0.0;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc += pv[q] * (kbeg + q < kend ? v[kbeg + q] : 0.0);
        }
        part[idx] = acc;
    }
    __syncthreads();
    double* u = ubase + nd.u_off;
    for (int a = tid; a < r; a += MF_THREADS) u[a] = v[c + a] - ((part[a] + part[r + a]) + (part[2 * r + a] + part[3 * r + a]));
    MFS_STAMP(4);
}
template <int MF_THREADS, bool GP>
__global__ __launch_bounds__(MF_THREADS) void k_mf_forward(const MfDev d, const MfSlots sl, int first, int n, int nrhs, long long usum, double* __restrict__ X) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const MfNode nd = d.nrec[first + blockIdx.x];                              // blockIdx.y = instance * nrhs + right-hand side
#ifdef CALIPSO_LDL_TRACE
    MFS_BEGIN(0)
    mf_forward_node<MF_THREADS>(d, nd, (size_t)blockIdx.y, (size_t)(sl.use ? sl.slot[blockIdx.y / nrhs] : (int)(blockIdx.y / nrhs)), n, usum, X, sm, mfs_on, mfs_tr);
#else
    mf_forward_node<MF_THREADS>(d, nd, (size_t)blockIdx.y, (size_t)(sl.use ? sl.slot[blockIdx.y / nrhs] : (int)(blockIdx.y / nrhs)), n, usum, X, sm);
#endif
}
// backward: z_C = y_C / D_C - L21' x_R (x_R final: it belongs to ancestors);  x_C = L11^-T z_C.  One wavefront per column for the product with L21'
// (lanes along the rows, contiguous), then the c dependent steps in one wavefront (lane = column, x_i by v_readlane).
template <int MF_THREADS>
__device__ __forceinline__ void mf_backward_node(const MfDev& d, const MfNode& nd, const size_t yi, const size_t zs, int n, double* __restrict__ X, double* __restrict__ sm,
                                                 const bool mfs_on = false, const int mfs_tr = 0) {
    MFS_STAMP(0);
    const int f = nd.f, c = nd.c, r = nd.r, m = c + r;
    double* v = sm;
    double* x = X + yi * n;
    const double* panel = d.panel + zs * d.sPanel; const double* Dg = d.D + zs * d.sD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double* P = panel + nd.panel_off;
    const int* R = d.rows + nd.rowptr;
    constexpr int NW = MF_THREADS / 64, CPW = 64 / NW;                         // columns per wavefront: k = wave + NW q (c <= 64); their loads travel together
    // the panel entries of the first row chunk (and, wavefront 0, the rows of L11 it will walk) are requested before the vector is assembled
    double pvf[CPW], pl0[32], pl1[32];
#pragma unroll
    for (int q = 0; q < CPW; ++q) { const int k = wave + NW * q; pvf[q] = (k < c && lane < r) ? P[(c + lane) + (size_t)k * m] : 0.0; }
    if (wave == 0) {
#pragma unroll
        for (int q = 0; q < 32; ++q) { const int i = c - 1 - q; pl0[q] = (i >= 1 && lane < i) ? P[i + (size_t)lane * m] : 0.0; }
#pragma unroll
        for (int q = 0; q < 32; ++q) { const int i = c - 33 - q; pl1[q] = (i >= 1 && lane < i) ? P[i + (size_t)lane * m] : 0.0; }
    }
    for (int i = tid; i < m; i += MF_THREADS) v[i] = i < c ? x[f + i] / Dg[f + i] : x[R[i - c]];
    __syncthreads();
    MFS_STAMP(1);
    if (r > 0) {                                                               // (a root has no rows below its columns: nothing to subtract)
        double acc[CPW];
#pragma unroll
        for (int q = 0; q < CPW; ++q) acc[q] = 0.0;
        for (int a0 = 0; a0 < r; a0 += 64) {
            const int a = a0 + lane;
            const double va = a < r ? v[c + a] : 0.0;
            double pv[CPW];
            if (a0 == 0) {
#pragma unroll
                for (int q = 0; q < CPW; ++q) pv[q] = pvf[q];
            } else {
#pragma unroll
                for (int q = 0; q < CPW; ++q) { const int k = wave + NW * q; pv[q] = (k < c && a < r) ? P[(c + a) + (size_t)k * m] : 0.0; }
            }
#pragma unroll
            for (int q = 0; q < CPW; ++q) acc[q] += pv[q] * va;
        }
#pragma unroll
        for (int q = 0; q < CPW; ++q) {
            const int k = wave + NW * q;
            const double t = calipso::wave_sum_l63(acc[q]);     // (16 sums per wavefront: by __shfl_down they were 3.8 of a node's 8 us, bench/mf_solve_trace.py)
            if (lane == 63 && k < c) v[k] -= t;
        }
    }
    __syncthreads();
    MFS_STAMP(2);
    if (wave == 0) {
        double zk = lane < c ? v[lane] : 0.0;
#pragma unroll
        for (int q = 0; q < 32; ++q) { const int i = c - 1 - q; if (i >= 1) zk = fma(-pl0[q], mf_readlane_d(zk, i), zk); }
#pragma unroll
        for (int q = 0; q < 32; ++q) { const int i = c - 33 - q; if (i >= 1) zk = fma(-pl1[q], mf_readlane_d(zk, i), zk); }
        if (lane < c) x[f + lane] = zk;
    }
    MFS_STAMP(3); MFS_STAMP(4);
}
template <int MF_THREADS, bool GP>
__global__ __launch_bounds__(MF_THREADS) void k_mf_backward(const MfDev d, const MfSlots sl, int first, int n, int nrhs, double* __restrict__ X) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const MfNode nd = d.nrec[first + blockIdx.x];
#ifdef CALIPSO_LDL_TRACE
    MFS_BEGIN(1)
    mf_backward_node<MF_THREADS>(d, nd, (size_t)blockIdx.y, (size_t)(sl.use ? sl.slot[blockIdx.y / nrhs] : (int)(blockIdx.y / nrhs)), n, X, sm, mfs_on, mfs_tr);
#else
    mf_backward_node<MF_THREADS>(d, nd, (size_t)blockIdx.y, (size_t)(sl.use ? sl.slot[blockIdx.y / nrhs] : (int)(blockIdx.y / nrhs)), n, X, sm);
#endif
}

#include "sparse_wide.hpp"

int g_wide_fronts = 1;         // calipso_hip_debug_wide_fronts: 0 = the one-workgroup kernel for the global-memory fronts too (plans made afterwards)
static int g_mf_items = 1;      // extend-add items for the LDS fronts (calipso_hip_debug_mf_items)
struct MfSeg { int first, count; size_t lds_factor, lds_solve; int threads; bool global; int ypan; MfWide wide; };   // ypan: room for the 64 x MF_PY exchange rows of the in-register panels
// launch helpers: the thread count of a level is fixed by the analyse phase (MfSeg::threads)
#define MF_LAUNCH(KERNEL, G, GRID, LDS, STREAM, ...)                                                                              \
    do {                                                                                                                          \
        if ((G).global) hipLaunchKernelGGL((KERNEL<512, true>), GRID, dim3(512), LDS, STREAM, __VA_ARGS__);                       \
        else if ((G).threads == 512) hipLaunchKernelGGL((KERNEL<512, false>), GRID, dim3(512), LDS, STREAM, __VA_ARGS__);         \
        else hipLaunchKernelGGL((KERNEL<256, false>), GRID, dim3(256), LDS, STREAM, __VA_ARGS__);                                \
    } while (0)


// the sweeps of a solve: 256 threads whatever the front (their 175-184 registers allow two wavefronts per SIMD: TWO workgroups of 256 per compute unit instead of one
// of 512 — a level of a batch has more fronts than compute units, and the node's work is a few microseconds; the bits do not depend on the thread count)
#define MF_LAUNCH_SOLVE(KERNEL, G, GRID, LDS, STREAM, ...)                                                                        \
    do {                                                                                                                          \
        if ((G).global) hipLaunchKernelGGL((KERNEL<512, true>), GRID, dim3(512), LDS, STREAM, __VA_ARGS__);                       \
        else hipLaunchKernelGGL((KERNEL<256, false>), GRID, dim3(256), LDS, STREAM, __VA_ARGS__);                                \
    } while (0)

struct Segment { int first, count; bool chain; };

}  // namespace

struct calipso_hip_sparse {
    int n = 0, device = 0;
    i64 nnzA = 0, nnzU = 0, nnzL = 0, flops = 0;
    int levels = 0, widest = 0;
    bool lds_acc = true;
    std::vector<i64> perm;                       // 1-based, perm[k] = vertex eliminated k-th
    std::vector<int> hLp, hLi;                   // host copy of the pattern (get_factor)
    std::vector<Segment> plan;
    bool mf = false;                             // multifrontal numeric phase (nested-dissection orders whose fronts fit the LDS)
    MfDev md{};
    std::vector<MfSeg> mplan;
    std::vector<int> h_nfirst, h_ncols, h_nrows, h_rowptr, h_rows;
    std::vector<long long> h_panel_off;
    long long panel_total = 0, usum = 0;
    size_t cap_uvec = 0;
    int max_front = 0, nnodes = 0;
    int batch = 1, selected = 0;                 // matrices of this pattern factored together / the one get_factor reads
    long long upd_total = 0, pool_total = 0;
    int wide_count = 0;                          // most fronts of one level factored by many workgroups (their scratch: sparse_wide.hpp)
    int wide_blocks = 1;                         // most row blocks of such a front in the backward sweep
    double* wpart = nullptr;                     // partial products of the backward sweep: (column of the solve, node of the level, row block, 64)
    size_t cap_wpart = 0;
    std::vector<i64> inertia_all;                // batch x 3
    std::vector<void*> dev;                      // every device allocation
    SpDev d{};
    int* d_perm = nullptr;
    double *d_Aval = nullptr, *d_rhs = nullptr, *d_x = nullptr;
    size_t cap_rhs = 0;
    int work_slots = 0;
    hipStream_t stream = nullptr;
    bool owns_stream = true;      // false once a solver handle lent its own stream (sparse_borrow_stream): a stream per plan would only crowd the hardware queues
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipGraphExec_t graph_factor = nullptr;
    bool graph_tried = false, factored = false;
    i64 inertia[3] = {0, 0, 0};
    double ms_factor = 0.0, ms_solve = 0.0;
    std::string err;
};

static thread_local std::string g_sparse_err;
#define PK(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { (s ? s->err : g_sparse_err) = std::string(#call) + ": " + hipGetErrorString(e__); return CALIPSO_ERR_HIP; } } while (0)

namespace {

// (re)allocate everything that holds VALUES, for `batch` matrices of the analysed pattern: instance-major
int alloc_values(calipso_hip_sparse* s, int batch) {
    const size_t B = (size_t)batch;
    for (double** pp : {&s->d_Aval, &s->d.Lx, &s->d.D, &s->md.panel, &s->md.upd, &s->md.fpool, &s->md.wscr}) if (*pp) { (void)hipFree(*pp); *pp = nullptr; }
    PK(hipMalloc((void**)&s->d_Aval, sizeof(double) * B * std::max<size_t>((size_t)s->nnzA, 1)));
    PK(hipMemset(s->d_Aval, 0, sizeof(double) * B * std::max<size_t>((size_t)s->nnzA, 1)));      // (entries a structured handle never writes are structural zeros)
    PK(hipMalloc((void**)&s->d.D, sizeof(double) * B * (size_t)s->n));
    if (s->mf) {
        PK(hipMalloc((void**)&s->md.panel, sizeof(double) * B * std::max<size_t>((size_t)s->panel_total, 1)));
        PK(hipMalloc((void**)&s->md.upd, sizeof(double) * B * std::max<size_t>((size_t)s->upd_total, 1)));
        if (s->pool_total) PK(hipMalloc((void**)&s->md.fpool, sizeof(double) * B * (size_t)s->pool_total));
        if (s->wide_count) PK(hipMalloc((void**)&s->md.wscr, sizeof(double) * B * (size_t)s->wide_count * WF_SCR));
    } else {
        PK(hipMalloc((void**)&s->d.Lx, sizeof(double) * B * std::max<size_t>((size_t)s->nnzL, 1)));
    }
    s->d.Aval = s->d_Aval;
    s->md.Aval = s->d_Aval; s->md.D = s->d.D;
    s->md.sA = s->nnzA; s->md.sPanel = s->panel_total; s->md.sUpd = s->upd_total; s->md.sD = s->n; s->md.sPool = s->pool_total;
    s->batch = batch; s->selected = 0; s->factored = false;
    s->inertia_all.assign(3 * B, 0);
    return CALIPSO_OK;
}

template <typename T>
int upload(calipso_hip_sparse* s, const std::vector<T>& h, const T** out) {
    T* p = nullptr;
    PK(hipMalloc((void**)&p, sizeof(T) * std::max<size_t>(h.size(), 1)));
    s->dev.push_back(p);
    if (!h.empty()) PK(hipMemcpy(p, h.data(), sizeof(T) * h.size(), hipMemcpyHostToDevice));
    *out = p;
    return CALIPSO_OK;
}

// room for the partial products of the wide fronts' backward sweep, for `cols` columns per solve
int mf_reserve_wide_solve(calipso_hip_sparse* s, size_t cols) {
    if (!s->wide_count) return CALIPSO_OK;
    const size_t need = cols * (size_t)s->wide_count * (size_t)s->wide_blocks * 64;
    if (need > s->cap_wpart) {
        if (s->wpart) (void)hipFree(s->wpart);
        s->wpart = nullptr; s->cap_wpart = 0;
        PK(hipMalloc((void**)&s->wpart, sizeof(double) * need));
        s->cap_wpart = need;
    }
    return CALIPSO_OK;
}
// the numeric factorisation of nz matrices (storage slots sl) on stream st: a launch per level of the tree (fronts beyond the LDS: three, sparse_wide.hpp)
void mf_enqueue_factor(calipso_hip_sparse* s, hipStream_t st, const MfSlots& sl, unsigned nz) {
    for (const MfSeg& g : s->mplan) {
        if (g.wide.on) mf_wide_factor(st, s->md, sl, g.wide, g.first, g.count, nz);
        else MF_LAUNCH(k_mf_factor, g, dim3((unsigned)g.count, nz), g.lds_factor, st, s->md, sl, g.first, g.ypan);
    }
}
// both sweeps of a solve for ny = instances x nrhs columns of X (already permuted)
void mf_enqueue_solve(calipso_hip_sparse* s, hipStream_t st, const MfSlots& sl, unsigned ny, int nrhs, double* X) {
    const bool wide_ok = s->wpart && (size_t)ny * (size_t)s->wide_count * (size_t)s->wide_blocks * 64 <= s->cap_wpart;
    for (const MfSeg& g : s->mplan) {
        if (g.wide.solve && wide_ok) mf_wide_forward(st, s->md, sl, g.wide, g.first, g.count, ny, s->n, nrhs, s->usum, X);
        else MF_LAUNCH_SOLVE(k_mf_forward, g, dim3((unsigned)g.count, ny), g.lds_solve, st, s->md, sl, g.first, s->n, nrhs, s->usum, X);
    }
    for (auto g = s->mplan.rbegin(); g != s->mplan.rend(); ++g) {
        if (g->wide.solve && wide_ok) mf_wide_backward(st, s->md, sl, g->wide, g->first, g->count, ny, s->n, nrhs, X, s->wpart, s->wide_blocks);
        else MF_LAUNCH_SOLVE(k_mf_backward, (*g), dim3((unsigned)g->count, ny), g->lds_solve, st, s->md, sl, g->first, s->n, nrhs, X);
    }
}

void enqueue_factor(calipso_hip_sparse* s) {
    if (s->mf) { mf_enqueue_factor(s, s->stream, MfSlots{}, (unsigned)s->batch); return; }
    const size_t lds = s->lds_acc ? sizeof(double) * (size_t)s->n : 0;
    for (int z = 0; z < s->batch; ++z) {                    // the column method takes the matrices of a batch one after the other
        SpDev d = s->d;
        d.Aval += (size_t)z * (size_t)s->nnzA; d.Lx += (size_t)z * (size_t)s->nnzL; d.D += (size_t)z * (size_t)s->n;
        for (const Segment& g : s->plan) {
            const int grid = g.chain ? 1 : std::min(g.count, s->work_slots);
            if (s->lds_acc) hipLaunchKernelGGL(k_sp_factor<true>, dim3(grid), dim3(SP_THREADS), lds, s->stream, d, g.first, g.count);
            else hipLaunchKernelGGL(k_sp_factor<false>, dim3(grid), dim3(SP_THREADS), 0, s->stream, d, g.first, g.count);
        }
    }
}

}  // namespace


// ---- hooks for the Newton handle (ldl.hip): the Schur complement S of a stage-structured problem through the multifrontal path -----------------
// S lives densely (column-major, leading dimension NP) in every instance's slab; `src` holds, for each entry of the analysed pattern, its offset
// row + col * NP.  Everything is enqueued on the CALLER's stream; nothing synchronises.
namespace {
__global__ void k_gather_dense(calipso::Batch bt, const double* __restrict__ S, const long long* __restrict__ src, long long nnz, double* __restrict__ Aval) {
    calipso::inst_shift(bt, S);
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nnz) Aval[(size_t)bt.slot[blockIdx.z] * (size_t)nnz + (size_t)q] = S[src[q]];
}
__global__ void k_permute_in_slab(calipso::Batch bt, const double* __restrict__ x, const int* __restrict__ perm, int n, double* __restrict__ dx) {
    calipso::inst_shift(bt, x);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dx[(size_t)blockIdx.z * n + i] = x[perm[i]];
}
__global__ void k_permute_out_slab(calipso::Batch bt, const double* __restrict__ dx, const int* __restrict__ perm, int n, double* __restrict__ x) {
    calipso::inst_shift(bt, x);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[perm[i]] = dx[(size_t)blockIdx.z * n + i];
}
// the p columns of X (leading dimension ld) of ONE instance: blockIdx.y = column
__global__ void k_permute_in_cols(const double* __restrict__ X, long long ld, const int* __restrict__ perm, int n, double* __restrict__ dx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dx[(size_t)blockIdx.y * n + i] = X[(size_t)blockIdx.y * ld + perm[i]];
}
__global__ void k_permute_out_cols(const double* __restrict__ dx, const int* __restrict__ perm, int n, double* __restrict__ X, long long ld) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) X[(size_t)blockIdx.y * ld + perm[i]] = dx[(size_t)blockIdx.y * n + i];
}
// compute_inertia! (linear_solver.jl:33-44) of the S block: signs of its n pivots added to the handle's counters (icount[3..5] = positive, non-positive, zero)
__global__ __launch_bounds__(256) void k_count_signs(calipso::Batch bt, const double* __restrict__ D, int n, int* __restrict__ icount) {
    calipso::inst_shift_i(bt, icount);
    __shared__ int sh[3][4];
    const double* Dz = D + (size_t)bt.slot[blockIdx.z] * n;
    int pos = 0, nonpos = 0, zero = 0;
    for (int i = threadIdx.x; i < n; i += 256) { const double v = Dz[i]; pos += v > 0.0; nonpos += !(v > 0.0); zero += v == 0.0; }
    pos = calipso::wave_sum_i(pos); nonpos = calipso::wave_sum_i(nonpos); zero = calipso::wave_sum_i(zero);
    if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = pos; sh[1][threadIdx.x >> 6] = nonpos; sh[2][threadIdx.x >> 6] = zero; }
    __syncthreads();
    if (threadIdx.x == 0) {
        icount[3] += sh[0][0] + sh[0][1] + sh[0][2] + sh[0][3];
        icount[4] += sh[1][0] + sh[1][1] + sh[1][2] + sh[1][3];
        icount[5] += sh[2][0] + sh[2][1] + sh[2][2] + sh[2][3];
    }
}
}  // namespace

namespace calipso {
int sparse_reserve_solve(calipso_hip_sparse* s, int batch);
bool sparse_is_multifrontal(const calipso_hip_sparse* sp) { return sp && sp->mf; }
// a plan owned by a solver handle is only ever driven on that handle's stream: give the plan's own stream back
void sparse_borrow_stream(calipso_hip_sparse* sp, hipStream_t st) {
    if (!sp) return;
    if (sp->owns_stream && sp->stream) { (void)hipStreamSynchronize(sp->stream); (void)hipStreamDestroy(sp->stream); }
    sp->stream = st; sp->owns_stream = false;
}
int sparse_batch(const calipso_hip_sparse* sp) { return sp ? sp->batch : 0; }
// gather the pattern's entries from the dense S of every instance of `bt`, factor, add the pivot signs to the instances' counters
void sparse_values(calipso_hip_sparse* sp, double** values, long long* stride) { *values = sp ? sp->d_Aval : nullptr; *stride = sp ? (long long)sp->nnzA : 0; }
// values_in_place: the caller has written the pattern's entries into sparse_values() itself (structured handles: blocks.hip) — no gather
int sparse_factor_from_dense(calipso_hip_sparse* sp, hipStream_t st, const Batch& bt, const double* S, const long long* src, int* icount, bool values_in_place) {
    if (!sp || !sp->mf || bt.n > sp->batch) return CALIPSO_ERR_ARGUMENT;
    for (int k = 0; k < bt.n; ++k) if (bt.slot[k] >= sp->batch) return CALIPSO_ERR_ARGUMENT;
    const unsigned nz = (unsigned)bt.n;
    if (!values_in_place) hipLaunchKernelGGL(k_gather_dense, dim3((unsigned)((sp->nnzA + 255) / 256), 1, nz), dim3(256), 0, st, bt, S, src, (long long)sp->nnzA, sp->d_Aval);
    MfSlots sl{};
    sl.use = 1;
    for (int k = 0; k < bt.n; ++k) sl.slot[k] = bt.slot[k];
    mf_enqueue_factor(sp, st, sl, nz);
    hipLaunchKernelGGL(k_count_signs, dim3(1, 1, nz), dim3(256), 0, st, bt, sp->d.D, sp->n, icount);
    sp->factored = true;
    return CALIPSO_OK;
}
// x <- S^-1 x for the first n entries of every instance's x (a slab buffer)
int sparse_solve_inplace(calipso_hip_sparse* sp, hipStream_t st, const Batch& bt, double* x) {
    if (!sp || !sp->mf || bt.n > sp->batch) return CALIPSO_ERR_ARGUMENT;
    const unsigned nz = (unsigned)bt.n, gx = (unsigned)((sp->n + 255) / 256);
    MfSlots sl{};
    sl.use = 1;
    for (int k = 0; k < bt.n; ++k) { if (bt.slot[k] >= sp->batch) return CALIPSO_ERR_ARGUMENT; sl.slot[k] = bt.slot[k]; }
    hipLaunchKernelGGL(k_permute_in_slab, dim3(gx, 1, nz), dim3(256), 0, st, bt, x, sp->d_perm, sp->n, sp->d_x);
    mf_enqueue_solve(sp, st, sl, nz, 1, sp->d_x);
    hipLaunchKernelGGL(k_permute_out_slab, dim3(gx, 1, nz), dim3(256), 0, st, bt, sp->d_x, sp->d_perm, sp->n, x);
    return CALIPSO_OK;
}
// X (ld x p, column-major, in the slab of the single instance `slot`) <- S^-1 X: all p right-hand sides through the tree in the same launches
int sparse_solve_inplace_multi(calipso_hip_sparse* s, hipStream_t st, int slot, double* X, long long ld, int p) {
    if (!s || !s->mf || slot < 0 || slot >= s->batch || p < 1 || p > 65535) return CALIPSO_ERR_ARGUMENT;
    const int rc = sparse_reserve_solve(s, p);
    if (rc != CALIPSO_OK) return rc;
    MfSlots sl{};
    sl.use = 1;
    // every column of the launch belongs to the same factor: with nrhs = p the kernels take slot[blockIdx.y / p] = slot[0]
    sl.slot[0] = slot;
    const unsigned gx = (unsigned)((s->n + 255) / 256), ny = (unsigned)p;
    hipLaunchKernelGGL(k_permute_in_cols, dim3(gx, ny), dim3(256), 0, st, X, ld, s->d_perm, s->n, s->d_x);
    mf_enqueue_solve(s, st, sl, ny, p, s->d_x);
    hipLaunchKernelGGL(k_permute_out_cols, dim3(gx, ny), dim3(256), 0, st, s->d_x, s->d_perm, s->n, X, ld);
    return CALIPSO_OK;
}
// buffers for in-place solves of `batch` instances with one right-hand side each (sparse_solve_inplace allocates nothing)
int sparse_reserve_solve(calipso_hip_sparse* s, int batch) {
    PK(hipSetDevice(s->device));
    const size_t need = (size_t)s->n * (size_t)batch;
    if (need > s->cap_rhs) {
        if (s->d_rhs) (void)hipFree(s->d_rhs);
        if (s->d_x) (void)hipFree(s->d_x);
        s->d_rhs = s->d_x = nullptr; s->cap_rhs = 0;
        PK(hipMalloc((void**)&s->d_rhs, sizeof(double) * need)); PK(hipMalloc((void**)&s->d_x, sizeof(double) * need));
        s->cap_rhs = need;
    }
    const size_t need_u = (size_t)std::max<long long>(s->usum, 1) * (size_t)batch;
    if (need_u > s->cap_uvec) {
        if (s->md.uvec) (void)hipFree(s->md.uvec);
        s->md.uvec = nullptr; s->cap_uvec = 0;
        PK(hipMalloc((void**)&s->md.uvec, sizeof(double) * need_u));
        s->cap_uvec = need_u;
    }
    return mf_reserve_wide_solve(s, (size_t)batch);
}
void sparse_work(const calipso_hip_sparse* sp, double out[3]) { out[0] = (double)sp->flops; out[1] = (double)sp->nnzL; out[2] = (double)sp->n; }
void sparse_describe(const calipso_hip_sparse* sp, int64_t out[4]) { out[0] = sp->levels; out[1] = sp->max_front; out[2] = sp->nnzU; out[3] = sp->mf ? 2 : (sp->lds_acc ? 1 : 0); }
}  // namespace calipso

// (tests / A-B timing) how plans made AFTERWARDS treat fronts beyond the LDS: 1 = many workgroups per front (default), 0 = one; returns the old value
// (tests / A-B timing) how plans made AFTERWARDS treat fronts beyond the LDS: 1 = many workgroups per front (default), 0 = one; returns the old value
// (tests / A-B timing) whether plans made AFTERWARDS carry the extend-add items of the LDS fronts (1, default) or assemble child by child, row by row (0); returns the old value
extern "C" int32_t calipso_hip_debug_mf_items(int32_t on) { const int was = g_mf_items; if (on >= 0) g_mf_items = on != 0; return was; }
extern "C" int32_t calipso_hip_debug_wide_fronts(int32_t on) { const int was = g_wide_fronts; if (on >= 0) g_wide_fronts = on != 0; return was; }
#ifdef CALIPSO_LDL_TRACE
extern "C" int32_t calipso_hip_debug_mf_trace(long long* out, int32_t reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mf_trace), sizeof(long long) * 64 * 12) != hipSuccess) return -1;
    int n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_mf_trace_n), sizeof(int)) != hipSuccess) return -1;
    if (reset) { const int z = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_mf_trace_n), &z, sizeof(int)); }
    return n;
}
extern "C" int32_t calipso_hip_debug_mfs_trace(long long* out, int32_t reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mfs_trace), sizeof(long long) * 128 * 8) != hipSuccess) return -1;
    int n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_mfs_trace_n), sizeof(int)) != hipSuccess) return -1;
    if (reset) { const int z = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_mfs_trace_n), &z, sizeof(int)); }
    return n;
}
#endif

extern "C" {

const char* calipso_hip_sparse_last_error(calipso_hip_sparse* s) { return s ? s->err.c_str() : g_sparse_err.c_str(); }

int32_t calipso_hip_sparse_destroy(calipso_hip_sparse* s) {
    if (!s) return CALIPSO_OK;
    (void)hipSetDevice(s->device);
    if (s->stream && s->owns_stream) (void)hipStreamSynchronize(s->stream);
    if (s->graph_factor) (void)hipGraphExecDestroy(s->graph_factor);
    for (void* p : s->dev) if (p) (void)hipFree(p);
    for (double* p : {s->d_Aval, s->d.Lx, s->d.D, s->md.panel, s->md.upd, s->md.fpool, s->md.wscr}) if (p) (void)hipFree(p);
    if (s->d_rhs) (void)hipFree(s->d_rhs);
    if (s->d_x) (void)hipFree(s->d_x);
    if (s->md.uvec) (void)hipFree(s->md.uvec);
    if (s->wpart) (void)hipFree(s->wpart);
    if (s->e0) (void)hipEventDestroy(s->e0);
    if (s->e1) (void)hipEventDestroy(s->e1);
    if (s->stream && s->owns_stream) (void)hipStreamDestroy(s->stream);
    delete s;
    return CALIPSO_OK;
}

// QDLDL(A; perm) analyse phase (qdldl.jl:134-188): order, P A P', etree, pattern of L — plus the level schedule of the device factorisation.
// colptr / rowval: Julia SparseMatrixCSC pattern (1-based); only the upper triangle is read (triu!, linear_solver.jl:23).
// method: 0 natural, 1 RCM, 2 minimum degree, 4 nested dissection, 3 = `perm` (1-based, perm[k] = vertex eliminated k-th).
static int32_t sparse_create_impl(int64_t n, const int64_t* colptr, const int64_t* rowval, int32_t method, const int64_t* perm, int32_t device,
                                  calipso_hip_sparse** out);
int32_t calipso_hip_sparse_create(int64_t n, const int64_t* colptr, const int64_t* rowval, int32_t method, const int64_t* perm, int32_t device,
                                  calipso_hip_sparse** out) {
    if (!out) return CALIPSO_ERR_ARGUMENT;
    *out = nullptr;
    const int32_t rc = sparse_create_impl(n, colptr, rowval, method, perm, device, out);
    if (rc != CALIPSO_OK && *out) {             // a failure after the handle was made (stream, uploads, allocations): no half-built handle leaves here
        g_sparse_err = (*out)->err;
        (void)calipso_hip_sparse_destroy(*out);
        *out = nullptr;
    }
    return rc;
}
static int32_t sparse_create_impl(int64_t n, const int64_t* colptr, const int64_t* rowval, int32_t method, const int64_t* perm, int32_t device,
                                  calipso_hip_sparse** out) {
    calipso_hip_sparse* s = nullptr;
    if (n < 1 || n > 0x3fffffff || !colptr || method < 0 || method > 5 || (method == 3 && !perm)) {
        g_sparse_err = "calipso_hip_sparse_create: bad arguments"; return CALIPSO_ERR_ARGUMENT;
    }
    if (!calipso::csc_pattern_ok(n, colptr, rowval)) { g_sparse_err = "colptr must be 1-based (Julia SparseMatrixCSC) and non-decreasing, rowval in 1..n, nnz < 2^31"; return CALIPSO_ERR_ARGUMENT; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_sparse_err = "no HIP device available (libcalipso_hip has no CPU path)"; return CALIPSO_ERR_HIP; }
    if (device < 0 || device >= ndev) { g_sparse_err = "device ordinal out of range"; return CALIPSO_ERR_ARGUMENT; }
    // ---- order
    std::vector<i64> p((size_t)n);
    std::vector<std::pair<int, int>> pieces;        // nested dissection: (first position, count) of every leaf piece / separator
    if (method == 3) {
        std::vector<char> seen((size_t)n, 0);
        for (i64 k = 0; k < n; ++k) { const i64 v = perm[k]; if (v < 1 || v > n || seen[(size_t)v - 1]) { g_sparse_err = "perm is not a permutation of 1:n"; return CALIPSO_ERR_ARGUMENT; } seen[(size_t)v - 1] = 1; p[(size_t)k] = v; }
    } else if (method >= 4) {
        const int rc = calipso::nested_dissection_pieces(n, colptr, rowval, p.data(), pieces);
        if (rc < 0) { g_sparse_err = "nested dissection failed"; return rc; }
    } else {
        const int rc = calipso_hip_ordering(n, colptr, rowval, method, p.data());
        if (rc < 0) { g_sparse_err = "calipso_hip_ordering failed"; return rc; }
    }
    std::vector<int> ip((size_t)n);
    for (i64 k = 0; k < n; ++k) ip[(size_t)p[(size_t)k] - 1] = (int)k;
    // ---- P A P' as a lower CSC (rows ascending), with the gather map from the caller's nzval
    const i64 nnzA = colptr[n] - 1;
    std::vector<std::vector<std::pair<int, int>>> col((size_t)n);       // (row, source index)
    i64 nnzU = 0;
    for (i64 c = 0; c < n; ++c)
        for (i64 q = colptr[c] - 1; q < colptr[c + 1] - 1; ++q) {
            const i64 r = rowval[q] - 1;
            if (r < 0 || r >= n) { g_sparse_err = "row index out of range"; return CALIPSO_ERR_ARGUMENT; }
            if (r > c) continue;
            const int pr = ip[(size_t)r], pc = ip[(size_t)c];
            col[(size_t)std::min(pr, pc)].push_back({std::max(pr, pc), (int)q});
            ++nnzU;
        }
    std::vector<int> Alp((size_t)n + 1, 0), Ali, Asrc;
    Ali.reserve((size_t)nnzU); Asrc.reserve((size_t)nnzU);
    std::vector<std::vector<int>> lowrow((size_t)n);                    // row view of the strictly lower part: columns i < j of row j
    for (int j = 0; j < (int)n; ++j) {
        auto& cj = col[(size_t)j];
        std::sort(cj.begin(), cj.end());
        for (size_t a = 1; a < cj.size(); ++a) if (cj[a].first == cj[a - 1].first) { g_sparse_err = "duplicate entry in the CSC pattern"; return CALIPSO_ERR_ARGUMENT; }
        for (auto& e : cj) { Ali.push_back(e.first); Asrc.push_back(e.second); if (e.first > j) lowrow[(size_t)e.first].push_back(j); }
        Alp[(size_t)j + 1] = (int)Ali.size();
    }
    // QDLDL_etree! refuses a matrix with an empty column of the upper triangle (qdldl.jl:366-371): column j of triu(P A P') = row j of the lower part + diagonal
    for (int j = 0; j < (int)n; ++j) {
        const bool diag = Alp[(size_t)j] < Alp[(size_t)j + 1] && Ali[(size_t)Alp[(size_t)j]] == j;
        if (!diag && lowrow[(size_t)j].empty()) { g_sparse_err = "empty column in triu(A): QDLDL_etree! fails on this matrix (qdldl.jl:366-371)"; return CALIPSO_ERR_ARGUMENT; }
    }
    // ---- elimination tree (Liu, path compression)
    std::vector<int> parent((size_t)n, -1), anc((size_t)n, -1);
    for (int j = 0; j < (int)n; ++j)
        for (int i0 : lowrow[(size_t)j]) {
            int i = i0;
            while (i != -1 && i < j) { const int nxt = anc[(size_t)i]; anc[(size_t)i] = j; if (nxt == -1) parent[(size_t)i] = j; i = nxt; }
        }
    // ---- pattern of L: struct(L_j) = struct(A_j below the diagonal) U (struct(L_c) \ {j}) over the children c of j
    std::vector<std::vector<int>> children((size_t)n);
    for (int j = 0; j < (int)n; ++j) if (parent[(size_t)j] >= 0) children[(size_t)parent[(size_t)j]].push_back(j);
    std::vector<std::vector<int>> Lcol((size_t)n);
    std::vector<int> mark((size_t)n, -1);
    i64 nnzL = 0, flops = 0;
    for (int j = 0; j < (int)n; ++j) {
        std::vector<int>& lj = Lcol[(size_t)j];
        mark[(size_t)j] = j;
        for (int q = Alp[(size_t)j]; q < Alp[(size_t)j + 1]; ++q) { const int r = Ali[(size_t)q]; if (r > j && mark[(size_t)r] != j) { mark[(size_t)r] = j; lj.push_back(r); } }
        for (int c : children[(size_t)j]) for (int r : Lcol[(size_t)c]) if (r != j && mark[(size_t)r] != j) { mark[(size_t)r] = j; lj.push_back(r); }
        std::sort(lj.begin(), lj.end());
        nnzL += (i64)lj.size();
        if (nnzL > 0x7fffffff) { g_sparse_err = "nnz(L) exceeds 2^31 - 1"; return CALIPSO_ERR_ARGUMENT; }
    }
    std::vector<int> Lp((size_t)n + 1, 0), Li; Li.reserve((size_t)nnzL);
    for (int j = 0; j < (int)n; ++j) { Li.insert(Li.end(), Lcol[(size_t)j].begin(), Lcol[(size_t)j].end()); Lp[(size_t)j + 1] = (int)Li.size(); }
    // ---- row view
    std::vector<int> Rp((size_t)n + 1, 0);
    for (int v : Li) Rp[(size_t)v + 1] += 1;
    for (int j = 0; j < (int)n; ++j) Rp[(size_t)j + 1] += Rp[(size_t)j];
    std::vector<int> Rk((size_t)nnzL), Rpos((size_t)nnzL), Rend((size_t)nnzL), nextr(Rp.begin(), Rp.end() - 1);
    for (int k = 0; k < (int)n; ++k)
        for (int q = Lp[(size_t)k]; q < Lp[(size_t)k + 1]; ++q) {
            const int at = nextr[(size_t)Li[(size_t)q]]++;
            Rk[(size_t)at] = k; Rpos[(size_t)at] = q; Rend[(size_t)at] = Lp[(size_t)k + 1];
            flops += (i64)(Lp[(size_t)k + 1] - q);
        }
    // ---- levels and the launch plan
    std::vector<int> level((size_t)n, 0);
    int height = 0;
    for (int j = 0; j < (int)n; ++j) {          // children have smaller indices: one ascending pass
        if (parent[(size_t)j] >= 0) level[(size_t)parent[(size_t)j]] = std::max(level[(size_t)parent[(size_t)j]], level[(size_t)j] + 1);
        height = std::max(height, level[(size_t)j] + 1);
    }
    std::vector<int> order((size_t)n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return level[(size_t)a] < level[(size_t)b]; });
    std::vector<Segment> plan;
    int widest = 0;
    for (int a = 0; a < (int)n;) {
        int b = a;
        while (b < (int)n && level[(size_t)order[(size_t)b]] == level[(size_t)order[(size_t)a]]) ++b;
        const int cnt = b - a;
        widest = std::max(widest, cnt);
        if (cnt == 1 && !plan.empty() && plan.back().chain) plan.back().count += 1;        // extend the chain of single-column levels
        else plan.push_back({a, cnt, cnt == 1});
        a = b;
    }
    // ---- multifrontal symbolic phase (method 4): the dissection pieces as supernodes
    // A piece the level structure could not split (a clique-like stage block) or a wide separator is cut into a chain of chunks of <= w columns;
    // w is lowered until every front (chunk + its rows below) fits one CU's LDS — e.g. 56 for stages of 56 variables, where chunks that straddle
    // two stages would carry both stages' neighbours.
    bool use_mf = false;
    const std::vector<std::pair<int, int>> base_pieces = pieces;
    int NN = 0;
    std::vector<int> m_first, m_cols, m_rows, m_parent, m_rowptr, m_rowsv, m_rel, m_childptr, m_children, m_order, m_Aloc;
    std::vector<long long> m_panel_off, m_upd_off, m_u_off;
    std::vector<MfSeg> mplan;
    long long panel_total = 0, upd_total = 0, usum = 0;
    int max_front = 0, mf_levels = 0, mf_widest = 0;
    // last resort: chunks of 64 columns with the oversized fronts in global memory (MF_MAX_FRONT_GLOBAL)
    std::vector<long long> m_foff;
    long long pool_total = 0;
    std::vector<int> m_wbase, w_ptrE, w_ptrC, w_Ecol, w_Esrc;     // fronts factored by many workgroups (sparse_wide.hpp)
    std::vector<MfRowItem> w_C;
    int wide_count = 0;
    for (int attempt = 0; attempt < 8; ++attempt) {
        const int width = attempt < 7 ? (int[]){64, 56, 48, 40, 32, 24, 16}[attempt] : 64;
        const int front_limit = attempt < 7 ? MF_MAX_FRONT : (g_wide_fronts ? MF_MAX_FRONT_WIDE : MF_MAX_FRONT_GLOBAL);
        if (method != 4 || base_pieces.empty() || use_mf) break;
        pieces.clear();
        for (const auto& bp : base_pieces) for (int o = 0; o < bp.second; o += width) pieces.push_back({bp.first + o, std::min(width, bp.second - o)});
        NN = (int)pieces.size();
        use_mf = true;
        max_front = 0;
        m_rowsv.clear(); m_children.clear(); mplan.clear(); panel_total = upd_total = usum = 0; mf_widest = 0;
        {
        std::vector<int> node_of((size_t)n, -1);
        for (int t = 0; t < NN; ++t) for (int q = 0; q < pieces[(size_t)t].second; ++q) node_of[(size_t)(pieces[(size_t)t].first + q)] = t;
        for (int j = 0; j < (int)n; ++j) if (node_of[(size_t)j] < 0) use_mf = false;
        std::vector<std::vector<int>> R((size_t)NN);
        m_parent.assign((size_t)NN, -1);
        std::vector<int> stamp((size_t)n, -1);
        for (int t = 0; t < NN && use_mf; ++t) {
            // rows below the node reached from any of its columns, plus what the children passed up (already in R[t])
            const int f = pieces[(size_t)t].first, c = pieces[(size_t)t].second, last = f + c - 1;
            std::vector<int>& rt = R[(size_t)t];
            for (int v : rt) stamp[(size_t)v] = t;
            for (int j = f; j <= last; ++j) for (int i : Lcol[(size_t)j]) if (i > last && stamp[(size_t)i] != t) { stamp[(size_t)i] = t; rt.push_back(i); }
            std::sort(rt.begin(), rt.end());
            max_front = std::max(max_front, c + (int)rt.size());
            if (!rt.empty()) {
                const int par = node_of[(size_t)rt[0]];
                m_parent[(size_t)t] = par;
                const int pl = pieces[(size_t)par].first + pieces[(size_t)par].second - 1;
                std::vector<int>& rp = R[(size_t)par];                  // closure: what the parent does not own it must carry on
                for (int v : rt) if (v > pl) rp.push_back(v);
                std::sort(rp.begin(), rp.end()); rp.erase(std::unique(rp.begin(), rp.end()), rp.end());
            }
        }
        if (max_front > front_limit) use_mf = false;                    // a front that does not fit: next chunk width / global fronts / column method
        if (use_mf) {
            m_first.resize((size_t)NN); m_cols.resize((size_t)NN); m_rows.resize((size_t)NN); m_rowptr.assign((size_t)NN + 1, 0);
            m_panel_off.resize((size_t)NN); m_upd_off.resize((size_t)NN); m_u_off.resize((size_t)NN);
            for (int t = 0; t < NN; ++t) {
                const int c = pieces[(size_t)t].second, r = (int)R[(size_t)t].size();
                m_first[(size_t)t] = pieces[(size_t)t].first; m_cols[(size_t)t] = c; m_rows[(size_t)t] = r;
                m_rowptr[(size_t)t + 1] = m_rowptr[(size_t)t] + r;
                m_panel_off[(size_t)t] = panel_total; panel_total += (long long)(c + r) * c;
                m_upd_off[(size_t)t] = upd_total; upd_total += (long long)r * r;
                m_u_off[(size_t)t] = usum; usum += r;
                m_rowsv.insert(m_rowsv.end(), R[(size_t)t].begin(), R[(size_t)t].end());
            }
            if (upd_total + panel_total > (1ll << 31)) { use_mf = false; continue; }      // (16 GiB of update matrices + panels per matrix: the column method instead)
            // relative indices into the parent's front, children lists, levels
            m_rel.assign(m_rowsv.size(), 0);
            std::vector<std::vector<int>> kids((size_t)NN);
            std::vector<int> lev((size_t)NN, 0);
            for (int t = 0; t < NN; ++t) {
                const int par = m_parent[(size_t)t];
                if (par < 0) continue;
                kids[(size_t)par].push_back(t);
                lev[(size_t)par] = std::max(lev[(size_t)par], lev[(size_t)t] + 1);      // children precede parents in position order
                const int pf = m_first[(size_t)par], pc = m_cols[(size_t)par];
                const std::vector<int>& rp = R[(size_t)par];
                for (int k = 0; k < m_rows[(size_t)t]; ++k) {
                    const int v = R[(size_t)t][(size_t)k];
                    m_rel[(size_t)m_rowptr[(size_t)t] + (size_t)k] = v < pf + pc ? v - pf : pc + (int)(std::lower_bound(rp.begin(), rp.end(), v) - rp.begin());
                }
            }
            m_childptr.assign((size_t)NN + 1, 0);
            for (int t = 0; t < NN; ++t) { m_childptr[(size_t)t + 1] = m_childptr[(size_t)t] + (int)kids[(size_t)t].size(); m_children.insert(m_children.end(), kids[(size_t)t].begin(), kids[(size_t)t].end()); }
            // where each entry of the permuted lower CSC lands in its node's front
            m_Aloc.assign(Ali.size(), 0);
            for (int j = 0; j < (int)n; ++j) {
                const int t = node_of[(size_t)j], f = m_first[(size_t)t], c = m_cols[(size_t)t];
                const std::vector<int>& rt = R[(size_t)t];
                for (int q = Alp[(size_t)j]; q < Alp[(size_t)j + 1]; ++q) {
                    const int i = Ali[(size_t)q];
                    const int lr = i < f + c ? i - f : c + (int)(std::lower_bound(rt.begin(), rt.end(), i) - rt.begin());
                    m_Aloc[(size_t)q] = lr * (lr + 1) / 2 + (j - f);          // packed lower triangle (k_mf_factor: tri)
                }
            }
            m_order.resize((size_t)NN);
            m_foff.assign((size_t)NN, 0); pool_total = 0;
            m_wbase.assign((size_t)NN, -1); w_ptrE.clear(); w_ptrC.clear(); w_Ecol.clear(); w_Esrc.clear(); w_C.clear(); wide_count = 0;
            std::iota(m_order.begin(), m_order.end(), 0);
            std::stable_sort(m_order.begin(), m_order.end(), [&](int a, int b) { return lev[(size_t)a] < lev[(size_t)b]; });
            for (int a = 0; a < NN;) {
                int b = a; size_t lf = 0, ls = 0, mmax = 0;
                while (b < NN && lev[(size_t)m_order[(size_t)b]] == lev[(size_t)m_order[(size_t)a]]) {
                    const int t = m_order[(size_t)b]; const size_t m = (size_t)(m_cols[(size_t)t] + m_rows[(size_t)t]);
                    lf = std::max(lf, sizeof(double) * (m * (m + 1) / 2 + 2 * m)); ls = std::max(ls, sizeof(double) * 5 * m);   // the vector + four slices of partial sums per row
                    mmax = std::max(mmax, m);
                    ++b;
                }
                const bool glob = mmax > (size_t)MF_MAX_FRONT;
                if (glob) {                                               // the level's fronts in the (reused) global pool; the solves keep only v in LDS
                    long long off = 0;
                    for (int q = a; q < b; ++q) { const int t = m_order[(size_t)q]; const long long mm = m_cols[(size_t)t] + m_rows[(size_t)t]; m_foff[(size_t)t] = off; off += mm * (mm + 1) / 2; }
                    pool_total = std::max(pool_total, off);
                    lf = 0;
                }
                int ypan = 0;                                             // full 16-column panels of fronts with >= 32 rows go through registers (pivot16.hpp) when the LDS has room for the exchange rows
                if (!glob && mmax >= 32 && lf + sizeof(double) * 64 * MF_PY <= (size_t)(160 * 1024 - 2048)) { ypan = 1; lf += sizeof(double) * 64 * MF_PY; }
                MfWide wide;                                              // fronts beyond the LDS: many workgroups per front (sparse_wide.hpp)
                if (glob && g_wide_fronts) {
                    wide.on = 1;
                    wide_count = std::max(wide_count, b - a);
                    for (int q = a; q < b; ++q) {
                        const int t = m_order[(size_t)q], f = m_first[(size_t)t], c = m_cols[(size_t)t], r = m_rows[(size_t)t], mm = c + r;
                        wide.m = std::max(wide.m, mm); wide.r = std::max(wide.r, r); wide.nch = std::max(wide.nch, (int)kids[(size_t)t].size());
                        // per row of the front: its entries of A (any order: distinct targets) and the child rows that land in it, children ascending
                        const int base = (int)w_ptrE.size();
                        m_wbase[(size_t)t] = base;
                        std::vector<int> cntE((size_t)mm + 1, 0), cntC((size_t)mm + 1, 0);
                        const std::vector<int>& rt = R[(size_t)t];
                        auto local = [&](int i) { return i < f + c ? i - f : c + (int)(std::lower_bound(rt.begin(), rt.end(), i) - rt.begin()); };
                        for (int j = f; j < f + c; ++j) for (int e = Alp[(size_t)j]; e < Alp[(size_t)j + 1]; ++e) ++cntE[(size_t)local(Ali[(size_t)e]) + 1];
                        for (int ch : kids[(size_t)t]) for (int k = 0; k < m_rows[(size_t)ch]; ++k) ++cntC[(size_t)m_rel[(size_t)m_rowptr[(size_t)ch] + (size_t)k] + 1];
                        const int e0 = (int)w_Ecol.size(), c0 = (int)w_C.size();
                        for (int i = 0; i < mm; ++i) { cntE[(size_t)i + 1] += cntE[(size_t)i]; cntC[(size_t)i + 1] += cntC[(size_t)i]; }
                        for (int i = 0; i <= mm; ++i) { w_ptrE.push_back(e0 + cntE[(size_t)i]); w_ptrC.push_back(c0 + cntC[(size_t)i]); }
                        w_Ecol.resize((size_t)e0 + (size_t)cntE[(size_t)mm]); w_Esrc.resize(w_Ecol.size()); w_C.resize((size_t)c0 + (size_t)cntC[(size_t)mm]);
                        std::vector<int> atE(cntE.begin(), cntE.end() - 1), atC(cntC.begin(), cntC.end() - 1);
                        for (int j = f; j < f + c; ++j) for (int e = Alp[(size_t)j]; e < Alp[(size_t)j + 1]; ++e) {
                            const int at = e0 + atE[(size_t)local(Ali[(size_t)e])]++;
                            w_Ecol[(size_t)at] = j - f; w_Esrc[(size_t)at] = Asrc[(size_t)e];
                        }
                        for (int ch : kids[(size_t)t]) for (int k = 0; k < m_rows[(size_t)ch]; ++k) {       // kids are ascending: so is every row's item list
                            const int at = c0 + atC[(size_t)m_rel[(size_t)m_rowptr[(size_t)ch] + (size_t)k]]++;
                            w_C[(size_t)at] = {m_upd_off[(size_t)ch] + (long long)k * m_rows[(size_t)ch], m_rowptr[(size_t)ch], k, (int)(m_u_off[(size_t)ch] + k), 0};
                        }
                    }
                }
                wide.solve = wide.on && (wide.m > 2048 || wide.nch > 4);
                mplan.push_back({a, b - a, lf, ls, mmax > (size_t)MF_BIG ? 512 : 256, glob, ypan, wide});
                mf_widest = std::max(mf_widest, b - a);
                a = b;
            }
            mf_levels = (int)mplan.size();
        }
    }
    }
    // ---- the handle and its device side
    s = new calipso_hip_sparse();
    *out = s;
    s->n = (int)n; s->device = device; s->nnzA = nnzA; s->nnzU = nnzU; s->nnzL = nnzL; s->flops = flops;
    s->levels = height; s->widest = widest; s->perm = p; s->plan = plan; s->hLp = Lp; s->hLi = Li;
    s->lds_acc = n <= SP_LDS_MAX_N;
    PK(hipSetDevice(device));
    PK(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    PK(hipEventCreate(&s->e0)); PK(hipEventCreate(&s->e1));
    s->d.n = (int)n;
    int rc;
    std::vector<int> hperm((size_t)n);
    for (i64 k = 0; k < n; ++k) hperm[(size_t)k] = (int)(p[(size_t)k] - 1);
    const int* cperm = nullptr;
    if ((rc = upload(s, order, &s->d.order)) || (rc = upload(s, Alp, &s->d.Alp)) || (rc = upload(s, Ali, &s->d.Ali)) || (rc = upload(s, Asrc, &s->d.Asrc)) ||
        (rc = upload(s, Lp, &s->d.Lp)) || (rc = upload(s, Li, &s->d.Li)) || (rc = upload(s, Rp, &s->d.Rp)) || (rc = upload(s, Rk, &s->d.Rk)) ||
        (rc = upload(s, Rpos, &s->d.Rpos)) || (rc = upload(s, Rend, &s->d.Rend)) || (rc = upload(s, hperm, &cperm))) return rc;
    s->d_perm = const_cast<int*>(cperm);
    s->upd_total = use_mf ? upd_total : 0; s->panel_total = use_mf ? panel_total : 0;
    s->mf = use_mf;
    if ((rc = alloc_values(s, 1))) return rc;
    if (use_mf) {
        s->mplan = mplan; s->nnodes = NN; s->max_front = max_front; s->panel_total = panel_total; s->usum = usum;
        s->h_nfirst = m_first; s->h_ncols = m_cols; s->h_nrows = m_rows; s->h_rowptr = m_rowptr; s->h_rows = m_rowsv; s->h_panel_off = m_panel_off;
        s->levels = mf_levels; s->widest = mf_widest;
        MfDev& md = s->md;
        md.nnodes = NN;
        if ((rc = upload(s, m_order, &md.order)) || (rc = upload(s, m_first, &md.nfirst)) || (rc = upload(s, m_cols, &md.ncols)) || (rc = upload(s, m_rows, &md.nrows)) ||
            (rc = upload(s, m_rowptr, &md.rowptr)) || (rc = upload(s, m_rowsv, &md.rows)) || (rc = upload(s, m_rel, &md.rel)) || (rc = upload(s, m_childptr, &md.childptr)) ||
            (rc = upload(s, m_children, &md.children)) || (rc = upload(s, m_panel_off, &md.panel_off)) || (rc = upload(s, m_upd_off, &md.upd_off)) ||
            (rc = upload(s, m_u_off, &md.u_off)) || (rc = upload(s, m_Aloc, &md.Aloc)) || (rc = upload(s, m_foff, &md.foff))) return rc;
        {
            std::vector<MfNode> nrec((size_t)NN);
            std::vector<MfChild> crec;
            std::vector<int> pos_of((size_t)NN, 0), ypan_of((size_t)NN, 0);
            for (int pos = 0; pos < NN; ++pos) pos_of[(size_t)m_order[(size_t)pos]] = pos;
            for (const MfSeg& g : mplan) for (int q = 0; q < g.count; ++q) ypan_of[(size_t)(g.first + q)] = g.ypan;
            for (int pos = 0; pos < NN; ++pos) {
                const int t = m_order[(size_t)pos];
                MfNode& nd = nrec[(size_t)pos];
                nd.s = t; nd.f = m_first[(size_t)t]; nd.c = m_cols[(size_t)t]; nd.r = m_rows[(size_t)t];
                nd.rowptr = m_rowptr[(size_t)t]; nd.chfirst = (int)crec.size(); nd.nch = m_childptr[(size_t)t + 1] - m_childptr[(size_t)t];
                nd.alp0 = (int)Alp[(size_t)nd.f]; nd.alp1 = (int)Alp[(size_t)(nd.f + nd.c)]; nd.pad0 = m_wbase.empty() ? -1 : m_wbase[(size_t)t];
                nd.pad1 = m_parent[(size_t)t] >= 0 ? pos_of[(size_t)m_parent[(size_t)t]] : -1; nd.pad2 = ypan_of[(size_t)pos];
                nd.panel_off = m_panel_off[(size_t)t]; nd.upd_off = m_upd_off[(size_t)t]; nd.u_off = m_u_off[(size_t)t]; nd.foff = m_foff.empty() ? 0 : m_foff[(size_t)t];
                for (int q = m_childptr[(size_t)t]; q < m_childptr[(size_t)t + 1]; ++q) {
                    const int ch = m_children[(size_t)q];
                    crec.push_back({m_rows[(size_t)ch], m_rowptr[(size_t)ch], m_upd_off[(size_t)ch], m_u_off[(size_t)ch], pos_of[(size_t)ch], 0});
                }
            }
            // extend-add items of the fronts that live in LDS (k_mf_factor's assembly): per node the entries of its children's update matrices, children ascending, a
            // child's lower triangle row by row; not built when a node has more than 65535 children or the table would exceed 512 MiB (the row-wise loop then)
            std::vector<MfXItem> xit;
            {
                std::vector<char> lds_front((size_t)NN, 0);
                for (const MfSeg& g : mplan) for (int q = 0; q < g.count; ++q) lds_front[(size_t)(g.first + q)] = !g.global;
                size_t total = 0; bool ok = true;
                for (int pos = 0; pos < NN && ok; ++pos) {
                    if (!lds_front[(size_t)pos]) continue;
                    const int t = m_order[(size_t)pos];
                    if (m_childptr[(size_t)t + 1] - m_childptr[(size_t)t] > 65535) ok = false;
                    for (int q = m_childptr[(size_t)t]; q < m_childptr[(size_t)t + 1]; ++q) { const size_t rch = (size_t)m_rows[(size_t)m_children[(size_t)q]]; total += rch * (rch + 1) / 2; }
                }
                if (ok && g_mf_items && total <= ((size_t)512 << 20) / sizeof(MfXItem)) {
                    xit.reserve(total + 1);
                    for (int pos = 0; pos < NN; ++pos) {
                        MfNode& nd = nrec[(size_t)pos];
                        nd.xa0 = nd.xa1 = -1;
                        if (!lds_front[(size_t)pos]) continue;
                        const int t = m_order[(size_t)pos];
                        nd.xa0 = (int)xit.size();
                        for (int q = m_childptr[(size_t)t]; q < m_childptr[(size_t)t + 1]; ++q) {
                            const int ch = m_children[(size_t)q], rch = m_rows[(size_t)ch];
                            const int* rel = m_rel.data() + m_rowptr[(size_t)ch];
                            const unsigned ord = (unsigned)(q - m_childptr[(size_t)t]);
                            for (int a = 0; a < rch; ++a) for (int b = 0; b <= a; ++b)
                                xit.push_back({(unsigned)(m_upd_off[(size_t)ch] + (long long)a * rch + b), (unsigned)(rel[a] * (rel[a] + 1) / 2 + rel[b]) | (ord << 16)});
                        }
                        nd.xa1 = (int)xit.size();
                    }
                    if (xit.empty()) xit.push_back({0u, 0u});
                } else for (MfNode& nd : nrec) nd.xa0 = nd.xa1 = -1;
            }
            md.xit = nullptr;
            if ((rc = upload(s, nrec, &md.nrec)) || (rc = upload(s, crec, &md.crec)) || (!xit.empty() && (rc = upload(s, xit, &md.xit)))) return rc;
        }
        s->pool_total = pool_total; s->wide_count = wide_count;
        for (const MfSeg& g : mplan) if (g.wide.solve) s->wide_blocks = std::max(s->wide_blocks, (g.wide.r + WF_SOLVE_ROWS - 1) / WF_SOLVE_ROWS);
        if (wide_count && ((rc = upload(s, w_ptrE, &md.wptrE)) || (rc = upload(s, w_ptrC, &md.wptrC)) || (rc = upload(s, w_Ecol, &md.wEcol)) ||
                           (rc = upload(s, w_Esrc, &md.wEsrc)) || (rc = upload(s, w_C, &md.wC)))) return rc;
        if ((rc = alloc_values(s, 1))) return rc;      // (again: now with the pool of the global-memory fronts)
        for (const MfSeg& g : s->mplan) if (g.wide.on && !mf_wide_prepare(&s->err)) return CALIPSO_ERR_HIP;
        md.Alp = s->d.Alp; md.Asrc = s->d.Asrc;
        for (const void* fn : {(const void*)k_mf_factor<256, false>, (const void*)k_mf_factor<512, false>, (const void*)k_mf_forward<256, false>,
                               (const void*)k_mf_forward<512, false>, (const void*)k_mf_backward<256, false>, (const void*)k_mf_backward<512, false>})
            PK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048));     // (k_mf_factor also holds 1 KiB of static LDS)
    }
    s->work_slots = std::min(widest, 2048);
    if (s->lds_acc) {
        PK(hipFuncSetAttribute((const void*)k_sp_factor<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * SP_LDS_MAX_N)));
    } else {
        s->work_slots = std::min(s->work_slots, 512);
        PK(hipMalloc((void**)&s->d.work, sizeof(double) * (size_t)s->work_slots * (size_t)n)); s->dev.push_back(s->d.work);
    }
    return CALIPSO_OK;
}

// info = [n, nnz(triu A), nnz(L), levels of the elimination tree, launches per factorisation, widest level, multiply-adds per factorisation,
//         1 if the column accumulator lives in LDS]
int32_t calipso_hip_sparse_info(calipso_hip_sparse* s, int64_t info[8]) {
    if (!s || !info) return CALIPSO_ERR_ARGUMENT;
    info[0] = s->n; info[1] = s->nnzU; info[2] = s->nnzL; info[3] = s->levels; info[4] = s->mf ? (i64)s->mplan.size() : (i64)s->plan.size(); info[5] = s->widest; info[6] = s->flops; info[7] = s->mf ? 2 : (s->lds_acc ? 1 : 0);
    return CALIPSO_OK;
}

// Number of matrices (same pattern, different values) the following calls treat together: nzval = batch x nnz, inertia = batch x 3,
// b / x = batch x (n x nrhs).  The multifrontal path factors the whole batch in the same launches (one workgroup per front and matrix);
// the column method takes them one after the other.  Invalidates the current factorisation.
int32_t calipso_hip_sparse_set_batch(calipso_hip_sparse* s, int64_t batch) {
    if (!s || batch < 1 || batch > 65535) return CALIPSO_ERR_ARGUMENT;
    PK(hipSetDevice(s->device));
    PK(hipStreamSynchronize(s->stream));
    if (s->graph_factor) { (void)hipGraphExecDestroy(s->graph_factor); s->graph_factor = nullptr; }
    s->graph_tried = false;
    return alloc_values(s, (int)batch);
}
// which matrix of the batch calipso_hip_sparse_get_factor reads (default 0)
int32_t calipso_hip_sparse_select(calipso_hip_sparse* s, int64_t instance) {
    if (!s || instance < 0 || instance >= s->batch) return CALIPSO_ERR_ARGUMENT;
    s->selected = (int)instance;
    return CALIPSO_OK;
}

// QDLDL_factor! + compute_inertia! (qdldl.jl:400-589, linear_solver.jl:19-44) for new values on the analysed pattern.
// nzval: batch x nnz(A) values in the caller's CSC order (host).  inertia: batch x 3 (may be NULL).  Returns CALIPSO_WARN_ZERO_PIVOT if any
// matrix met an exact zero pivot (its inertia[0] = -1).
static int32_t sparse_factorize(calipso_hip_sparse* s, const double* nzval, int64_t* inertia, hipMemcpyKind kind) {
    if (!s || (!nzval && s->nnzA > 0)) return CALIPSO_ERR_ARGUMENT;
    PK(hipSetDevice(s->device));
    const size_t B = (size_t)s->batch;
    if (s->nnzA) PK(hipMemcpyAsync(s->d_Aval, nzval, sizeof(double) * B * (size_t)s->nnzA, kind, s->stream));
    PK(hipEventRecord(s->e0, s->stream));
    if (!s->graph_tried) {          // the level schedule is a fixed launch sequence with fixed arguments: captured once, replayed afterwards
        s->graph_tried = true;
        hipGraph_t g = nullptr;
        if (hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            enqueue_factor(s);
            if (hipStreamEndCapture(s->stream, &g) == hipSuccess && g) {
                if (hipGraphInstantiate(&s->graph_factor, g, nullptr, nullptr, 0) != hipSuccess) s->graph_factor = nullptr;
                (void)hipGraphDestroy(g);
            }
        }
        (void)hipGetLastError();
    }
    if (s->graph_factor) PK(hipGraphLaunch(s->graph_factor, s->stream));
    else enqueue_factor(s);
    PK(hipEventRecord(s->e1, s->stream));
    std::vector<double> D(B * (size_t)s->n);
    PK(hipMemcpyAsync(D.data(), s->d.D, sizeof(double) * D.size(), hipMemcpyDeviceToHost, s->stream));
    PK(hipStreamSynchronize(s->stream));
    PK(hipGetLastError());
    float ms = 0.f; PK(hipEventElapsedTime(&ms, s->e0, s->e1)); s->ms_factor = ms;
    // compute_inertia! as the reference sees it: the up-looking factorisation stops at the first exact zero pivot in elimination order and the
    // rest of D stays at the zeros it was reset to (qdldl.jl:444,456,579)
    int rc = CALIPSO_OK;
    for (size_t z = 0; z < B; ++z) {
        const double* Dz = D.data() + z * (size_t)s->n;
        i64 pos = 0, nonpos = 0, zero = 0; int k = 0;
        for (; k < s->n; ++k) { const double dk = Dz[k]; if (dk == 0.0) break; pos += dk > 0.0; nonpos += dk <= 0.0; }
        if (k < s->n) { zero = s->n - k; nonpos += s->n - k; pos = -1; rc = CALIPSO_WARN_ZERO_PIVOT; }
        s->inertia_all[3 * z] = pos; s->inertia_all[3 * z + 1] = nonpos; s->inertia_all[3 * z + 2] = zero;
    }
    s->factored = true;
    if (inertia) std::copy(s->inertia_all.begin(), s->inertia_all.end(), inertia);
    return rc;
}
int32_t calipso_hip_sparse_factorize(calipso_hip_sparse* s, const double* nzval, int64_t* inertia) { return sparse_factorize(s, nzval, inertia, hipMemcpyHostToDevice); }
// the same with the values already resident on the handle's device (d_nzval: device pointer, batch x nnz)
int32_t calipso_hip_sparse_factorize_device(calipso_hip_sparse* s, const double* d_nzval, int64_t* inertia) { return sparse_factorize(s, d_nzval, inertia, hipMemcpyDeviceToDevice); }

// solve!(F, b) (qdldl.jl:330-351) for nrhs right-hand sides per matrix: b, x = batch x (column-major n x nrhs) host arrays (may alias)
static int32_t sparse_solve(calipso_hip_sparse* s, int64_t nrhs, const double* b, double* x, hipMemcpyKind kin, hipMemcpyKind kout) {
    if (!s || nrhs < 0 || (nrhs > 0 && (!b || !x)) || nrhs * (int64_t)(s ? s->batch : 1) > 65535) return CALIPSO_ERR_ARGUMENT;
    if (!s->factored) { s->err = "calipso_hip_sparse_solve: factorize first"; return CALIPSO_ERR_ARGUMENT; }
    if (nrhs == 0) return CALIPSO_OK;
    PK(hipSetDevice(s->device));
    const size_t cols = (size_t)nrhs * (size_t)s->batch;
    const size_t need = (size_t)s->n * cols;
    if (need > s->cap_rhs) {
        if (s->d_rhs) (void)hipFree(s->d_rhs);
        if (s->d_x) (void)hipFree(s->d_x);
        s->d_rhs = s->d_x = nullptr; s->cap_rhs = 0;
        PK(hipMalloc((void**)&s->d_rhs, sizeof(double) * need)); PK(hipMalloc((void**)&s->d_x, sizeof(double) * need));
        s->cap_rhs = need;
    }
    PK(hipMemcpyAsync(s->d_rhs, b, sizeof(double) * need, kin, s->stream));
    PK(hipEventRecord(s->e0, s->stream));
    const unsigned gx = (unsigned)((s->n + 255) / 256), ny = (unsigned)cols;
    hipLaunchKernelGGL(k_sp_permute_in, dim3(gx, ny), dim3(256), 0, s->stream, s->d_rhs, s->d_perm, s->n, s->d_x);
    if (s->mf) {
        const size_t need_u = (size_t)std::max<long long>(s->usum, 1) * cols;
        if (need_u > s->cap_uvec) {
            if (s->md.uvec) (void)hipFree(s->md.uvec);
            s->md.uvec = nullptr; s->cap_uvec = 0;
            PK(hipMalloc((void**)&s->md.uvec, sizeof(double) * need_u));
            s->cap_uvec = need_u;
        }
        { const int wrc = mf_reserve_wide_solve(s, cols); if (wrc) return wrc; }
        mf_enqueue_solve(s, s->stream, MfSlots{}, ny, (int)nrhs, s->d_x);
    } else {
        for (int z = 0; z < s->batch; ++z) {
            SpDev d = s->d;
            d.Lx += (size_t)z * (size_t)s->nnzL; d.D += (size_t)z * (size_t)s->n;
            double* xz = s->d_x + (size_t)z * (size_t)nrhs * (size_t)s->n;
            for (const Segment& g : s->plan)
                hipLaunchKernelGGL(k_sp_forward, dim3(g.chain ? 1 : (unsigned)std::min(g.count, 4096), (unsigned)nrhs), dim3(64), 0, s->stream, d, g.first, g.count, xz);
            for (auto g = s->plan.rbegin(); g != s->plan.rend(); ++g)
                hipLaunchKernelGGL(k_sp_backward, dim3(g->chain ? 1 : (unsigned)std::min(g->count, 4096), (unsigned)nrhs), dim3(64), 0, s->stream, d, g->first, g->count, xz);
        }
    }
    hipLaunchKernelGGL(k_sp_permute_out, dim3(gx, ny), dim3(256), 0, s->stream, s->d_x, s->d_perm, s->n, s->d_rhs);
    PK(hipEventRecord(s->e1, s->stream));
    PK(hipMemcpyAsync(x, s->d_rhs, sizeof(double) * need, kout, s->stream));
    PK(hipStreamSynchronize(s->stream));
    PK(hipGetLastError());
    float ms = 0.f; PK(hipEventElapsedTime(&ms, s->e0, s->e1)); s->ms_solve = ms;
    return CALIPSO_OK;
}
int32_t calipso_hip_sparse_solve(calipso_hip_sparse* s, int64_t nrhs, const double* b, double* x) { return sparse_solve(s, nrhs, b, x, hipMemcpyHostToDevice, hipMemcpyDeviceToHost); }
// the same with right-hand sides and solutions resident on the handle's device (device pointers; may alias)
int32_t calipso_hip_sparse_solve_device(calipso_hip_sparse* s, int64_t nrhs, const double* d_b, double* d_x) { return sparse_solve(s, nrhs, d_b, d_x, hipMemcpyDeviceToDevice, hipMemcpyDeviceToDevice); }

// the factor for inspection: perm[n] (1-based), Lp[n+1], Li[nnz(L)] (1-based, strictly lower, rows ascending — F.L of qdldl.jl:160-166 without the
// unit diagonal), Lx[nnz(L)], D[n]; any output may be NULL
int32_t calipso_hip_sparse_get_factor(calipso_hip_sparse* s, int64_t* perm, int64_t* Lp, int64_t* Li, double* Lx, double* D) {
    if (!s) return CALIPSO_ERR_ARGUMENT;
    if ((Lx || D) && !s->factored) { s->err = "calipso_hip_sparse_get_factor: factorize first"; return CALIPSO_ERR_ARGUMENT; }
    PK(hipSetDevice(s->device));
    if (perm) std::copy(s->perm.begin(), s->perm.end(), perm);
    if (Lp) for (size_t k = 0; k < s->hLp.size(); ++k) Lp[k] = (i64)s->hLp[k] + 1;
    if (Li) for (size_t k = 0; k < s->hLi.size(); ++k) Li[k] = (i64)s->hLi[k] + 1;
    if (Lx && s->nnzL && s->mf) {
        // multifrontal storage: dense m x c panels per node; pick the entries of the exact pattern out of them
        std::vector<double> panel((size_t)s->panel_total);
        PK(hipMemcpyAsync(panel.data(), s->md.panel + (size_t)s->selected * (size_t)s->panel_total, sizeof(double) * panel.size(), hipMemcpyDeviceToHost, s->stream));
        PK(hipStreamSynchronize(s->stream));
        for (int t = 0; t < s->nnodes; ++t) {
            const int f = s->h_nfirst[(size_t)t], c = s->h_ncols[(size_t)t], m = c + s->h_nrows[(size_t)t];
            const int* Rb = s->h_rows.data() + s->h_rowptr[(size_t)t]; const int* Re = Rb + s->h_nrows[(size_t)t];
            const double* Pn = panel.data() + s->h_panel_off[(size_t)t];
            for (int j = f; j < f + c; ++j)
                for (int q = s->hLp[(size_t)j]; q < s->hLp[(size_t)j + 1]; ++q) {
                    const int i = s->hLi[(size_t)q];
                    const int lr = i < f + c ? i - f : c + (int)(std::lower_bound(Rb, Re, i) - Rb);
                    Lx[q] = Pn[(size_t)lr + (size_t)(j - f) * m];
                }
        }
    } else
    if (Lx && s->nnzL) PK(hipMemcpyAsync(Lx, s->d.Lx + (size_t)s->selected * (size_t)s->nnzL, sizeof(double) * (size_t)s->nnzL, hipMemcpyDeviceToHost, s->stream));
    if (D) PK(hipMemcpyAsync(D, s->d.D + (size_t)s->selected * (size_t)s->n, sizeof(double) * (size_t)s->n, hipMemcpyDeviceToHost, s->stream));
    PK(hipStreamSynchronize(s->stream));
    return CALIPSO_OK;
}

// ms[0] = device time of the last factorisation (all level launches), ms[1] = of the last solve call; HIP events on the handle's stream
int32_t calipso_hip_sparse_timing(calipso_hip_sparse* s, double ms[2]) {
    if (!s || !ms) return CALIPSO_ERR_ARGUMENT;
    ms[0] = s->ms_factor; ms[1] = s->ms_solve;
    return CALIPSO_OK;
}

}  // extern "C"
