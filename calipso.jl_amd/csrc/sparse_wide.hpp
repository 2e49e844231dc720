// Fronts of the multifrontal factorisation that exceed one compute unit's LDS, factored by MANY workgroups (included by sparse.hip behind its MfDev /
// MfNode / MfSlots definitions, inside its anonymous namespace: one translation unit).
//
// A node of the dissection tree eliminates c <= 64 columns of a front of m = c + r rows (sparse.hip: a wide separator is a chain of such nodes).  With
// the front in the global-memory pool (packed lower triangle, row i at i (i + 1) / 2) k_mf_factor<512, true> gives the whole node to ONE workgroup: a
// 1500-row front is ~2 ms of one compute unit while 255 idle.  Here a LEVEL of such nodes is three launches, grid = (work, node of the level, matrix
// of the batch):
//   k_wf_assemble   a wavefront per ROW of the front, the row in LDS: zero, the entries of A in that row, then the rows of the children's update
//                   matrices that land in it, children in ascending order (the one-workgroup kernel's summation order: the assembled front has the
//                   same bits), one coalesced store of the finished row.  The host lists, per row, its entries and its (child, child row) items once
//                   per pattern: no search, no ordering between workgroups (a row has one owner).
//   k_wf_diag       one workgroup: the c x c diagonal block by the dense path's diag_block (ldl_device.hpp: 16 columns per wavefront in registers,
//                   X = L11^-1 and M = X' D^-1 X on the matrix cores; identity padding to 64) -> D, the rows < c of the panel, X and M for the next launch.
//   k_wf_update     a workgroup per 64 x 64 tile of the update matrix,  U = F22 - (F21 M) F21'  from the RAW rows F21 (the dense path's form: nothing
//                   between the pivots and the trailing update but M), written straight to the node's slot of the update pool; and, in the same
//                   launch, a workgroup per 64 rows of the panel,  L21 = F21 X' D^-1.
// The storage formats (panel column-major m x c, U row-major r x r, D) are k_mf_factor's: the solves and calipso_hip_sparse_get_factor do not change.
// Rounding differs from the one-workgroup kernel (inverse-based block solves instead of substitution; one K = 64 sum per update entry instead of four
// K = 16 sums): both meet the oracle's QDLDL to 1e-10 (tests/test_gpu_sparse.py).
#pragma once
// (ldl_device.hpp is included at the top of sparse.hip: this file sits inside a namespace)

constexpr int WF_LDT = 66;                                           // row stride of the operand tiles in LDS (k fastest: conflict-free fragment reads)
constexpr int WF_UPDATE_LDS = 3 * 64 * WF_LDT * (int)sizeof(double);
constexpr int WF_DIAG_LDS = calipso::DIAG_LDS_DOUBLES * (int)sizeof(double);
constexpr int WF_ROWS = 4;                                           // rows of a front per k_wf_assemble workgroup (a wavefront each); fewer where 4 rows exceed the LDS
constexpr int WF_ASSEMBLE_LDS = 160 * 1024 - 2048;
constexpr int WF_SOLVE_ROWS = 1024;                                  // rows of L21 per workgroup of the backward sweep's partial products
// scratch of one (node of the level, matrix): X | M | L11 (64 x 64 column-major each) | D (64) | counters
constexpr int WF_SCR_X = 0, WF_SCR_M = 4096, WF_SCR_L = 8192, WF_SCR_D = 12288, WF_SCR_I = 12352, WF_SCR = 12416;

__device__ __forceinline__ int wf_tri(int i) { return (i * (i + 1)) >> 1; }                 // m <= MF_MAX_FRONT_WIDE: fits 31 bits
__device__ __forceinline__ size_t wf_slot(const MfSlots& sl) { return sl.use ? (size_t)sl.slot[blockIdx.z] : (size_t)blockIdx.z; }
__device__ __forceinline__ double* wf_scratch(const MfDev& d) { return d.wscr + ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * WF_SCR; }

__global__ __launch_bounds__(64 * WF_ROWS) void k_wf_assemble(const MfDev d, const MfSlots sl, int first, int stride) {
    extern __shared__ __attribute__((aligned(16))) double wf_rows_lds[];
    const MfNode nd = d.nrec[first + blockIdx.y];
    const int m = nd.c + nd.r, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = (int)(blockDim.x >> 6) * (int)blockIdx.x + wave;
    if (row >= m) return;                                             // (no workgroup barrier below: a wavefront works alone on its row)
    const size_t z = wf_slot(sl);
    double* acc = wf_rows_lds + (size_t)wave * stride;
    const double* Aval = d.Aval + z * d.sA;
    const double* upd = d.upd + z * d.sUpd;
    const int base = nd.pad0 + row;
    const int e0 = d.wptrE[base], e1 = d.wptrE[base + 1], c0 = d.wptrC[base], c1 = d.wptrC[base + 1];
    MfRowItem it = c0 < c1 ? d.wC[c0] : MfRowItem{0, 0, 0, 0, 0};
    for (int e = lane; e <= row; e += 64) acc[e] = 0.0;
    for (int p = e0 + lane; p < e1; p += 64) acc[d.wEcol[p]] = Aval[d.wEsrc[p]];      // (the LDS queue of a wavefront is in order)
    for (int q = c0; q < c1; ++q) {
        const MfRowItem cur = it;
        if (q + 1 < c1) it = d.wC[q + 1];                             // the next item travels while this one is added
        const double* U = upd + cur.uoff;
        const int* rel = d.rel + cur.relptr;
        for (int b = lane; b <= cur.a; b += 64) acc[rel[b]] += U[b];  // rel is increasing: distinct targets, all at or left of the diagonal
    }
    double* Fr = d.fpool + z * d.sPool + nd.foff + wf_tri(row);
    for (int e = lane; e <= row; e += 64) Fr[e] = acc[e];
}

__global__ __launch_bounds__(calipso::DIAG_THREADS) void k_wf_diag(const MfDev d, const MfSlots sl, int first) {
    extern __shared__ __attribute__((aligned(16))) double wf_diag_lds[];
    const MfNode nd = d.nrec[first + blockIdx.y];
    const int f = nd.f, c = nd.c, m = c + nd.r, tid = threadIdx.x;
    const size_t z = wf_slot(sl);
    const double* F = d.fpool + z * d.sPool + nd.foff;
    double* scr = wf_scratch(d);
    const int lane = tid & 63, w = tid >> 6, R = w >> 2, Cc = w & 3, fr = lane & 15, fk = lane >> 4;
    calipso::v4d acc = (calipso::v4d){0.0, 0.0, 0.0, 0.0};
    if (R >= Cc) {                                                     // tile (R, C) of the block, symmetric inside the diagonal tiles, identity beyond c
        const int i = 16 * R + fr;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = 16 * Cc + fk + 4 * q;
            const int hi = i > k ? i : k, lo = i > k ? k : i;
            const double v = F[wf_tri(hi < c ? hi : c - 1) + (lo < c ? lo : 0)];
            acc[q] = hi < c ? v : (i == k ? 1.0 : 0.0);
        }
    }
    calipso::diag_block(wf_diag_lds, acc, 64, c, 0, 64, scr + WF_SCR_L, scr + WF_SCR_D, scr + WF_SCR_X, scr + WF_SCR_M, (int*)(scr + WF_SCR_I));
    __syncthreads();                                                   // (the block's own global stores, read back by the same workgroup)
    double* P = d.panel + z * d.sPanel + nd.panel_off;
    if (tid < c) (d.D + z * d.sD)[f + tid] = scr[WF_SCR_D + tid];
    for (int e = tid; e < c * 64; e += calipso::DIAG_THREADS) {
        const int k = e >> 6, i = e & 63;                              // lanes along the rows: contiguous stores
        if (i < c) P[i + (size_t)k * m] = i > k ? scr[WF_SCR_L + i + 64 * k] : 0.0;
    }
}

__global__ __launch_bounds__(256) void k_wf_update(const MfDev d, const MfSlots sl, int first) {
    extern __shared__ __attribute__((aligned(16))) double wf_lds[];
    double* As = wf_lds;                                               // As[row][k] = F21(64 bi + row, k): raw
    double* Bs = wf_lds + 64 * WF_LDT;                                 // Bs[row][k] = F21(64 bj + row, k)
    double* Ms = wf_lds + 2 * 64 * WF_LDT;                             // M (symmetric), then Z = As M;  the panel blocks: X
    const MfNode nd = d.nrec[first + blockIdx.y];
    const int c = nd.c, r = nd.r, m = c + r, tid = threadIdx.x;
    const int nb = (r + 63) >> 6, ntile = (nb * (nb + 1)) >> 1;
    const int t = (int)blockIdx.x;
    if (t >= ntile + nb) return;
    const bool panel = t >= ntile;
    int bi, bj;
    if (panel) { bi = bj = t - ntile; }
    else {
        bi = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
        while (((bi + 1) * (bi + 2)) >> 1 <= t) ++bi;
        while ((bi * (bi + 1)) >> 1 > t) --bi;
        bj = t - ((bi * (bi + 1)) >> 1);
    }
    const size_t z = wf_slot(sl);
    const double* F = d.fpool + z * d.sPool + nd.foff;
    const double* scr = wf_scratch(d);
    const int lane = tid & 63, wave = tid >> 6, fr = lane & 15, fk = lane >> 4;
    {   // operand tiles: a front row's first c entries are contiguous — lanes along k
        const int k = tid & 63, kc = k < c ? k : 0;
        double va[16], vb[16], vm[16];
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int rw = (tid >> 6) + 4 * it, ga = c + 64 * bi + rw, gb = c + 64 * bj + rw;
            va[it] = F[wf_tri(ga < m ? ga : m - 1) + kc];
            vb[it] = F[wf_tri(gb < m ? gb : m - 1) + kc];
            vm[it] = scr[(panel ? WF_SCR_X : WF_SCR_M) + tid + 256 * it];
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int rw = (tid >> 6) + 4 * it, ga = c + 64 * bi + rw, gb = c + 64 * bj + rw;
            As[rw * WF_LDT + k] = (ga < m && k < c) ? va[it] : 0.0;
            Bs[rw * WF_LDT + k] = (gb < m && k < c) ? vb[it] : 0.0;
            // M(k, rw) = M(rw, k);  X(k, rw) -> Xs[k][rw] (row k of X, its column index fastest)
            if (panel) Ms[k * WF_LDT + rw] = vm[it]; else Ms[rw * WF_LDT + k] = vm[it];
        }
    }
    if (panel) {
        // L21 = F21 X' D^-1: first operand = rows of X (panel columns), second = rows of F21 — the 16-lane index of the result runs along the contiguous rows of the panel
        double* P = d.panel + z * d.sPanel + nd.panel_off;
        __syncthreads();
        calipso::v4d acc[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[cb] = (calipso::v4d){0.0, 0.0, 0.0, 0.0};
        const double* Br = As + (16 * wave + fr) * WF_LDT + fk;
        const double* Xr = Ms + fr * WF_LDT + fk;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const double b = Br[4 * kk];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(Xr[16 * cb * WF_LDT + 4 * kk], b, acc[cb], 0, 0, 0);
        }
        const int row = c + 64 * bi + 16 * wave + fr;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int k = 16 * cb + fk + 4 * rr;
                if (row < m && k < c) P[row + (size_t)k * m] = acc[cb][rr] * (1.0 / scr[WF_SCR_D + k]);
            }
        return;
    }
    // the entries of F22 this lane will update, in flight over the products
    double old[4][4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int a = 64 * bi + 16 * wave + fk + 4 * rr;
        const int ac = a < r ? a : r - 1;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const int b = 64 * bj + 16 * cb + fr;
            old[cb][rr] = F[wf_tri(c + ac) + c + (b < ac ? b : ac)];
        }
    }
    __syncthreads();
    calipso::v4d acc[4];
    const double* Ar = As + (16 * wave + fr) * WF_LDT + fk;
    {   // Z = As M (rows 16 wave .. of the tile row)
        const double* Mr = Ms + fr * WF_LDT + fk;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[cb] = (calipso::v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const double a = Ar[4 * kk];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Mr[16 * cb * WF_LDT + 4 * kk], acc[cb], 0, 0, 0);
        }
    }
    __syncthreads();                                                   // every read of M is done: Z takes its place
#pragma unroll
    for (int cb = 0; cb < 4; ++cb)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) Ms[(16 * wave + fk + 4 * rr) * WF_LDT + 16 * cb + fr] = acc[cb][rr];
    __syncthreads();
    {
        const double* Zr = Ms + (16 * wave + fr) * WF_LDT + fk;
        const double* Br = Bs + fr * WF_LDT + fk;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[cb] = (calipso::v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const double a = Zr[4 * kk];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Br[16 * cb * WF_LDT + 4 * kk], acc[cb], 0, 0, 0);
        }
    }
    double* U = d.upd + z * d.sUpd + nd.upd_off;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int a = 64 * bi + 16 * wave + fk + 4 * rr;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const int b = 64 * bj + 16 * cb + fr;
            if (a < r && b <= a) U[(size_t)a * r + b] = old[cb][rr] - acc[cb][rr];
        }
    }
}

// what a level's launches must cover (maxima over its nodes)
struct MfWide { int on = 0; int m = 0, r = 0, nch = 0; int solve = 0; };   // solve: the sweeps of this level by many workgroups too (measured cross-over: ~2200 rows, or many children per node)

inline bool mf_wide_prepare(std::string* err) {
    if (!calipso::lds_attribute((const void*)k_wf_update, WF_UPDATE_LDS) || !calipso::lds_attribute((const void*)k_wf_diag, WF_DIAG_LDS) ||
        !calipso::lds_attribute((const void*)k_wf_assemble, WF_ASSEMBLE_LDS)) {
        if (err) *err = "the wide-front kernels: the LDS attribute was refused";
        return false;
    }
    return true;
}

inline void mf_wide_factor(hipStream_t st, const MfDev& md, const MfSlots& sl, const MfWide& w, int first, int count, unsigned nz) {
    const int stride = (w.m + 1) & ~1;
    int rows = WF_ROWS;
    while (rows > 1 && sizeof(double) * (size_t)stride * rows > (size_t)WF_ASSEMBLE_LDS) rows >>= 1;
    hipLaunchKernelGGL(k_wf_assemble, dim3((unsigned)((w.m + rows - 1) / rows), (unsigned)count, nz), dim3(64 * rows),
                       sizeof(double) * (size_t)stride * rows, st, md, sl, first, stride);
    hipLaunchKernelGGL(k_wf_diag, dim3(1, (unsigned)count, nz), dim3(calipso::DIAG_THREADS), WF_DIAG_LDS, st, md, sl, first);
    if (w.r > 0) {
        const int nb = (w.r + 63) / 64;
        hipLaunchKernelGGL(k_wf_update, dim3((unsigned)(nb * (nb + 1) / 2 + nb), (unsigned)count, nz), dim3(256), WF_UPDATE_LDS, st, md, sl, first);
    }
}

// ---- the sweeps of a solve through such fronts ------------------------------------------------------------------------------------------------
// One workgroup per node (k_mf_forward / k_mf_backward) keeps the front's vector in LDS (5 m doubles: m <= 4044) and reads the m x c panel alone (13 us at
// 1500 rows, 27 at 4000).  Here a level is two launches per sweep, grid = (row block, node, instance x right-hand side):
//   forward   k_wfs_head: v_C = b_C + the children's rows that land in the node's own columns (the per-row items of the assembly), y_C = L11^-1 v_C in one
//             wavefront;  k_wfs_tail: u_R = (children's rows) - L21 y_C, a thread per row, four k-slices combined in k_mf_forward's order;
//   backward  k_wfs_dot: partial products L21' x_R per block of WF_SOLVE_ROWS rows (a wavefront per 16 columns, lanes along the rows);
//             k_wfs_back: z_C = y_C / D - the partial products in ascending block order, x_C = L11^-T z_C in one wavefront.
// yi = blockIdx.z = instance * nrhs + right-hand side; the partial products of a level live in wpart[(yi * nodes of the level + node) * blocks * 64].
__global__ __launch_bounds__(64) void k_wfs_head(const MfDev d, const MfSlots sl, int first, int n, int nrhs, long long usum, double* __restrict__ X) {
    const MfNode nd = d.nrec[first + blockIdx.y];
    const int f = nd.f, c = nd.c, m = c + nd.r, lane = threadIdx.x;
    const size_t yi = blockIdx.z;
    double* x = X + yi * n;
    const double* ubase = d.uvec + yi * usum;
    const double* P = d.panel + (size_t)(sl.use ? sl.slot[yi / nrhs] : (int)(yi / nrhs)) * d.sPanel + nd.panel_off;
    double pl0[32], pl1[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) pl0[q] = (q < c && lane > q && lane < c) ? P[lane + (size_t)q * m] : 0.0;
#pragma unroll
    for (int q = 0; q < 32; ++q) { const int k = 32 + q; pl1[q] = (k < c && lane > k && lane < c) ? P[lane + (size_t)k * m] : 0.0; }
    double vi = 0.0;
    if (lane < c) {
        vi = x[f + lane];
        const int base = nd.pad0 + lane;
        for (int q = d.wptrC[base]; q < d.wptrC[base + 1]; ++q) vi += ubase[d.wC[q].uo];      // children in ascending order
    }
#pragma unroll
    for (int q = 0; q < 32; ++q) { if (q < c) vi = fma(-pl0[q], mf_readlane_d(vi, q), vi); }
#pragma unroll
    for (int q = 0; q < 32; ++q) { const int k = 32 + q; if (k < c) vi = fma(-pl1[q], mf_readlane_d(vi, k), vi); }
    if (lane < c) x[f + lane] = vi;
}

__global__ __launch_bounds__(256) void k_wfs_tail(const MfDev d, const MfSlots sl, int first, int n, int nrhs, long long usum, const double* __restrict__ X) {
    __shared__ double y[64];
    const MfNode nd = d.nrec[first + blockIdx.y];
    const int f = nd.f, c = nd.c, r = nd.r, m = c + r, tid = threadIdx.x;
    if ((int)blockIdx.x * 256 >= r) return;
    const size_t yi = blockIdx.z;
    const double* x = X + yi * n;
    double* ubase = d.uvec + yi * usum;
    const double* P = d.panel + (size_t)(sl.use ? sl.slot[yi / nrhs] : (int)(yi / nrhs)) * d.sPanel + nd.panel_off;
    if (tid < 64) y[tid] = tid < c ? x[f + tid] : 0.0;
    __syncthreads();
    const int a = (int)blockIdx.x * 256 + tid;
    if (a >= r) return;
    double v = 0.0;
    {
        const int base = nd.pad0 + c + a;
        for (int q = d.wptrC[base]; q < d.wptrC[base + 1]; ++q) v += ubase[d.wC[q].uo];
    }
    const int cs = (c + 3) / 4;
    const double* Pi = P + (c + a);
    double part[4];
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        const int kbeg = q4 * cs, kend = min(c, kbeg + cs);
        double pv[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) pv[q] = kbeg + q < kend ? Pi[(size_t)(kbeg + q) * m] : 0.0;
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc += pv[q] * (kbeg + q < kend ? y[kbeg + q] : 0.0);
        part[q4] = acc;
    }
    ubase[nd.u_off + a] = v - ((part[0] + part[1]) + (part[2] + part[3]));
}

__global__ __launch_bounds__(256) void k_wfs_dot(const MfDev d, const MfSlots sl, int first, int n, int nrhs, int nblk, const double* __restrict__ X, double* __restrict__ wpart) {
    const MfNode nd = d.nrec[first + blockIdx.y];
    const int c = nd.c, r = nd.r, m = c + r, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int a_beg = (int)blockIdx.x * WF_SOLVE_ROWS, a_end = min(r, a_beg + WF_SOLVE_ROWS);
    const size_t yi = blockIdx.z;
    double* out = wpart + ((yi * gridDim.y + blockIdx.y) * (size_t)nblk + blockIdx.x) * 64;
    if (a_beg >= r) { if (tid < 64) out[tid] = 0.0; return; }
    const double* x = X + yi * n;
    const double* P = d.panel + (size_t)(sl.use ? sl.slot[yi / nrhs] : (int)(yi / nrhs)) * d.sPanel + nd.panel_off;
    const int* R = d.rows + nd.rowptr;
    double acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.0;
    for (int a0 = a_beg; a0 < a_end; a0 += 64) {
        const int a = a0 + lane;
        const bool in = a < a_end;
        const double va = in ? x[R[in ? a : a_beg]] : 0.0;
        double pv[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) { const int k = wave + 4 * q; pv[q] = (k < c && in) ? P[(c + a) + (size_t)k * m] : 0.0; }
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] += pv[q] * va;
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const double t = calipso::wave_sum(acc[q]);
        if (lane == 0) out[wave + 4 * q] = t;
    }
}

__global__ __launch_bounds__(64) void k_wfs_back(const MfDev d, const MfSlots sl, int first, int n, int nrhs, int nblk, double* __restrict__ X, const double* __restrict__ wpart) {
    const MfNode nd = d.nrec[first + blockIdx.y];
    const int f = nd.f, c = nd.c, r = nd.r, m = c + r, lane = threadIdx.x;
    const size_t yi = blockIdx.z;
    double* x = X + yi * n;
    const size_t zs = (size_t)(sl.use ? sl.slot[yi / nrhs] : (int)(yi / nrhs));
    const double* P = d.panel + zs * d.sPanel + nd.panel_off;
    const double* Dg = d.D + zs * d.sD;
    double pl0[32], pl1[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) { const int i = c - 1 - q; pl0[q] = (i >= 1 && lane < i) ? P[i + (size_t)lane * m] : 0.0; }
#pragma unroll
    for (int q = 0; q < 32; ++q) { const int i = c - 33 - q; pl1[q] = (i >= 1 && lane < i) ? P[i + (size_t)lane * m] : 0.0; }
    double zk = 0.0;
    if (lane < c) {
        zk = x[f + lane] / Dg[f + lane];
        const double* in = wpart + (yi * gridDim.y + blockIdx.y) * (size_t)nblk * 64 + lane;
        double t = 0.0;
        const int used = (r + WF_SOLVE_ROWS - 1) / WF_SOLVE_ROWS;
        for (int b = 0; b < used; ++b) t += in[(size_t)b * 64];
        zk -= t;
    }
#pragma unroll
    for (int q = 0; q < 32; ++q) { const int i = c - 1 - q; if (i >= 1) zk = fma(-pl0[q], mf_readlane_d(zk, i), zk); }
#pragma unroll
    for (int q = 0; q < 32; ++q) { const int i = c - 33 - q; if (i >= 1) zk = fma(-pl1[q], mf_readlane_d(zk, i), zk); }
    if (lane < c) x[f + lane] = zk;
}

inline void mf_wide_forward(hipStream_t st, const MfDev& md, const MfSlots& sl, const MfWide& w, int first, int count, unsigned ny, int n, int nrhs, long long usum, double* X) {
    hipLaunchKernelGGL(k_wfs_head, dim3(1, (unsigned)count, ny), dim3(64), 0, st, md, sl, first, n, nrhs, usum, X);
    if (w.r > 0) hipLaunchKernelGGL(k_wfs_tail, dim3((unsigned)((w.r + 255) / 256), (unsigned)count, ny), dim3(256), 0, st, md, sl, first, n, nrhs, usum, X);
}
inline void mf_wide_backward(hipStream_t st, const MfDev& md, const MfSlots& sl, const MfWide& w, int first, int count, unsigned ny, int n, int nrhs, double* X, double* wpart, int nblk) {
    if (w.r > 0) hipLaunchKernelGGL(k_wfs_dot, dim3((unsigned)((w.r + WF_SOLVE_ROWS - 1) / WF_SOLVE_ROWS), (unsigned)count, ny), dim3(256), 0, st, md, sl, first, n, nrhs, nblk, X, wpart);
    hipLaunchKernelGGL(k_wfs_back, dim3(1, (unsigned)count, ny), dim3(64), 0, st, md, sl, first, n, nrhs, nblk, X, wpart);
}
