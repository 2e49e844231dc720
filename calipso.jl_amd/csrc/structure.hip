// structure.hip — stage-banded structure of the condensed system.
// Trajectory-optimisation problems (the reference's src/trajectory_optimization layer: indices.jl:41-180, sparsity.jl:28-129) order
// their variables stage by stage; the Lagrangian Hessian is then block diagonal, the dynamics / stage constraints touch two
// consecutive stages, and the Schur complement  S = Lxx + eps*I + omega*gx'gx + hx'(Omega hx)  of schur.hip is BANDED: S[i][j] = 0
// for |i - j| > hb.  Without pivoting the factor L of S keeps that band, so everything outside it can be skipped:
//   k_schur          only the tiles that intersect the band, and per tile only the constraint rows that touch both its row and
//                    its column range (a constraint row couples the variables between its first and last non-zero column)
//   LDL^T of S       panel rows and trailing tiles within the band of the panel
//   triangular solve off-diagonal updates within the band of the block
// The structure is taken from the non-zero pattern of the blocks currently held by the handle (calipso_hip_analyze_structure):
// the reference gets the same information from its sparsity pattern + AMD ordering (qdldl.jl:134-188).  Default = dense.
#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "internal.hpp"

using namespace calipso;

// ---- re-validation of uploads against an analysed structure ---------------------------------------------------------------------
// flag[0] != 0  <=>  the block holds a non-zero entry where the structure promises a zero
__global__ void k_check_band(int nx, int hb, const double* __restrict__ L, int* __restrict__ flag) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)nx * nx) return;
    const int i = (int)(idx % nx), j = (int)(idx / nx);
    const int dist = i > j ? i - j : j - i;
    if (dist > hb && L[idx] != 0.0) atomicOr(flag, 1);
}
__global__ void k_check_rows(int rows, int row0, int m, int nx, const double* __restrict__ Z, const int* __restrict__ zrow, int* __restrict__ flag) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)rows * nx) return;
    const int k = row0 + (int)(idx % rows), j = (int)(idx / rows);
    if ((j < zrow[2 * k] || j >= zrow[2 * k + 1]) && Z[k + (size_t)j * m] != 0.0) atomicOr(flag, 1);
}

// stage-parallel mode factors S only inside its skyline: a Hessian entry (i, j) must satisfy max(i, j) <= reach[min(i, j)]
__global__ void k_check_reach(int nx, const int* __restrict__ reach, const double* __restrict__ L, int* __restrict__ flag) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)nx * nx) return;
    const int i = (int)(idx % nx), j = (int)(idx / nx);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    if (hi > reach[lo] && L[idx] != 0.0) atomicOr(flag, 1);
}

// stage blocks: entry (i, j) of the Hessian must lie inside the diagonal block of column j
__global__ void k_check_colrange(int nx, const int* __restrict__ range, const double* __restrict__ L, int* __restrict__ flag) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)nx * nx) return;
    const int i = (int)(idx % nx), j = (int)(idx / nx);
    if ((i < range[2 * j] || i >= range[2 * j + 1]) && L[idx] != 0.0) atomicOr(flag, 1);
}

static int structure_clear(calipso_hip_solver* s) {
    s->band64 = 0; s->half_bandwidth = 0;
    s->stage_parallel = false; s->h_reach.clear();
    if (s->blocks.on) { blocks_release(s); s->hessian_dirty = true; }      // (the dense Schur kernel needs Lsym, which the blocks had borrowed, rebuilt)
    s->h_zrow.clear(); s->h_lreach.clear();
    const Dims& d = s->d;
    const size_t G = ((size_t)d.nx + 15) / 16;
    std::vector<int> kr(4 * G);
    for (size_t g = 0; g < G; ++g) { kr[4 * g] = 0; kr[4 * g + 1] = d.ne; kr[4 * g + 2] = 0; kr[4 * g + 3] = d.nc; }
    CK(hipSetDevice(s->device));
    CK(hipStreamSynchronize(s->stream));
    CK(hipMemcpy(s->krange, kr.data(), sizeof(int) * kr.size(), hipMemcpyHostToDevice));
    ldl_drop_graphs(s);      // the launch sequences change with the band
    return CALIPSO_OK;
}

namespace calipso {
int structure_validate(calipso_hip_solver* s, int which) {
    if (s->compact) return CALIPSO_OK;      // uploads of a structured handle are packed against the declared structure on the host
    const Dims& d = s->d;
    int* flag = s->icount + 60;
    CK(hipMemsetAsync(flag, 0, sizeof(int), s->stream));
    if (which == 0) {
        const size_t n = (size_t)d.nx * d.nx;
        if (s->band64 > 0) hipLaunchKernelGGL(k_check_band, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, d.nx, s->half_bandwidth, s->Lxx, flag);
        // the multifrontal factorisation gathers S through the skyline only (finer than the band, and in force even when the band covers everything)
        if (s->blocks.on && s->blocks.d_colrange) hipLaunchKernelGGL(k_check_colrange, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, d.nx, s->blocks.d_colrange, s->Lxx, flag);
        if (s->stage_parallel && s->d_reach) hipLaunchKernelGGL(k_check_reach, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, d.nx, s->d_reach, s->Lxx, flag);
    } else {
        const int rows = which == 1 ? d.ne : d.nc, row0 = which == 1 ? 0 : d.ne;
        const size_t n = (size_t)rows * d.nx;
        if (n) hipLaunchKernelGGL(k_check_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, rows, row0, d.m, d.nx, s->Z, s->zrow, flag);
    }
    CK(hipMemcpyAsync(s->hicount + 60, flag, sizeof(int), hipMemcpyDeviceToHost, s->stream));
    CK(hipStreamSynchronize(s->stream));
    if (s->hicount[60] != 0) {
        s->structure_resets += 1;
        s->err = "uploaded block has non-zeros outside the analysed structure: the handle is back to the dense treatment";
        return structure_clear(s);
    }
    return CALIPSO_OK;
}
}  // namespace calipso

extern "C" {

// out[0] = half bandwidth hb of S, out[1] = 64-wide blocks per panel inside the band (0 = treated as dense),
// out[2], out[3] = average number of equality / cone rows a 16-column group has to visit (of ne / nc)
int32_t calipso_hip_analyze_structure(calipso_hip_solver* s, int64_t out[4]) {
    if (!s) return CALIPSO_ERR_ARGUMENT;
    if (s->compact) { s->err = "calipso_hip_analyze_structure: the structure of a structured handle is fixed at creation"; return CALIPSO_ERR_ARGUMENT; }
    const Dims& d = s->d;
    const int nx = d.nx, ne = d.ne, nc = d.nc, m = d.m;
    CK(hipSetDevice(s->device));
    CK(hipStreamSynchronize(s->stream));
    s->stage_parallel = false;             // a new analysis: the multifrontal plan of an earlier pattern no longer applies (calipso_hip_set_stage_parallel again)
    if (s->blocks.on) { blocks_release(s); s->hessian_dirty = true; }      // ... nor do the stage blocks (calipso_hip_set_stage_blocks again)
    std::vector<double> L((size_t)nx * nx), Z((size_t)std::max(1, m) * nx);
    CK(hipMemcpy(L.data(), s->Lxx, sizeof(double) * L.size(), hipMemcpyDeviceToHost));
    if (m) CK(hipMemcpy(Z.data(), s->Z, sizeof(double) * (size_t)m * nx, hipMemcpyDeviceToHost));
    long hb = 0;
    for (int j = 0; j < nx; ++j)
        for (int i = 0; i < nx; ++i)
            if (L[i + (size_t)j * nx] != 0.0) hb = std::max<long>(hb, std::abs(i - j));
    // first / last non-zero column of every constraint row of the stacked Jacobian
    std::vector<int> cmin(std::max(1, m), nx), cmax(std::max(1, m), -1);
    for (int j = 0; j < nx; ++j)
        for (int k = 0; k < m; ++k)
            if (Z[k + (size_t)j * m] != 0.0) { cmin[k] = std::min(cmin[k], j); cmax[k] = std::max(cmax[k], j); }
    // the rows of a second-order cone are coupled through its weight block: they share the union of their column ranges
    for (int j = 0; j < d.n_soc; ++j) {
        const int st = ne + s->h_soc_start[j], dim = s->h_soc_dim[j];
        int lo = nx, hi = -1;
        for (int k = st; k < st + dim; ++k) { lo = std::min(lo, cmin[k]); hi = std::max(hi, cmax[k]); }
        for (int k = st; k < st + dim; ++k) { cmin[k] = lo; cmax[k] = hi; }
    }
    for (int k = 0; k < m; ++k) if (cmax[k] >= cmin[k]) hb = std::max<long>(hb, cmax[k] - cmin[k]);
    // skyline of S (column j of the lower triangle reaches down to row reach[j]): S = Lxx + sum over constraint rows of an outer product over the
    // row's column range, so j is coupled to everything up to the furthest end of a range that contains it (and to its Lxx neighbours)
    {
        std::vector<int>& reach = s->h_reach;
        reach.assign((size_t)nx, 0);
        for (int j = 0; j < nx; ++j) reach[(size_t)j] = j;
        for (int j = 0; j < nx; ++j)
            for (int i = 0; i < nx; ++i)
                if (L[i + (size_t)j * nx] != 0.0) { const int lo = std::min(i, j), hi = std::max(i, j); reach[(size_t)lo] = std::max(reach[(size_t)lo], hi); }
        s->h_lreach = reach;                                   // the Hessian alone: its diagonal blocks (calipso_hip_set_stage_blocks)
        std::vector<int> ext((size_t)nx, -1);                 // furthest range end among the ranges STARTING at a column
        for (int k = 0; k < m; ++k) if (cmax[k] >= cmin[k]) ext[(size_t)cmin[k]] = std::max(ext[(size_t)cmin[k]], cmax[k]);
        int run = -1;                                         // furthest end among the ranges that started at or before j
        for (int j = 0; j < nx; ++j) { run = std::max(run, ext[(size_t)j]); if (run >= j) reach[(size_t)j] = std::max(reach[(size_t)j], run); }
    }
    // per 16-column group: the range of equality rows / cone rows whose column range overlaps the group
    const int G = (nx + 15) / 16;
    std::vector<int> kr(4 * (size_t)G);
    for (int g = 0; g < G; ++g) { kr[4 * g] = ne; kr[4 * g + 1] = 0; kr[4 * g + 2] = nc; kr[4 * g + 3] = 0; }
    for (int k = 0; k < m; ++k) {
        if (cmax[k] < cmin[k]) continue;
        const bool eq = k < ne;
        const int r = eq ? k : k - ne;
        for (int g = cmin[k] / 16; g <= cmax[k] / 16; ++g) {
            int* e = &kr[4 * g + (eq ? 0 : 2)];
            e[0] = std::min(e[0], r); e[1] = std::max(e[1], r + 1);
        }
    }
    double ve = 0.0, vc = 0.0;
    for (int g = 0; g < G; ++g) { ve += std::max(0, kr[4 * g + 1] - kr[4 * g]); vc += std::max(0, kr[4 * g + 3] - kr[4 * g + 2]); }
    const int band64 = (int)((hb + 63) / 64);                  // 64-row blocks below a diagonal block that can be non-zero
    const int nblk = d.NP / NB;
    CK(hipMemcpy(s->krange, kr.data(), sizeof(int) * kr.size(), hipMemcpyHostToDevice));
    if (m) {
        std::vector<int> zr(2 * (size_t)m);
        for (int k = 0; k < m; ++k) { zr[2 * k] = cmax[k] >= cmin[k] ? cmin[k] : 0; zr[2 * k + 1] = cmax[k] >= cmin[k] ? cmax[k] + 1 : 0; }
        CK(hipMemcpy(s->zrow, zr.data(), sizeof(int) * zr.size(), hipMemcpyHostToDevice));
        s->h_zrow = zr;
    } else s->h_zrow.clear();
    s->half_bandwidth = (int)hb;
    s->band64 = band64 >= nblk - 1 ? 0 : std::max(1, band64);  // 0: nothing to skip
    // entries of S outside the band are never written in banded mode and must read as zero (the block inverses span whole
    // diagonal blocks): clear what an earlier dense factorisation may have left there
    CK(hipMemsetAsync(s->S, 0, sizeof(double) * (size_t)d.NP * d.NP, s->stream));
    CK(hipMemsetAsync(s->Tinv, 0, sizeof(double) * tinv_doubles(d.NP), s->stream));
    CK(hipStreamSynchronize(s->stream));
    ldl_drop_graphs(s);      // the launch sequences change with the band
    if (out) { out[0] = hb; out[1] = s->band64; out[2] = G ? (int64_t)(ve / G) : 0; out[3] = G ? (int64_t)(vc / G) : 0; }
    return CALIPSO_OK;
}

// back to the dense treatment (e.g. before uploading blocks with a different pattern)
int32_t calipso_hip_clear_structure(calipso_hip_solver* s) {
    if (!s) return CALIPSO_ERR_ARGUMENT;
    if (s->compact) { s->err = "calipso_hip_clear_structure: a structured handle has no dense treatment to go back to"; return CALIPSO_ERR_ARGUMENT; }
    return structure_clear(s);
}


// Stage-parallel factorisation of the Schur complement.  After calipso_hip_analyze_structure: S is treated as the sparse matrix
// its skyline describes, ordered by nested dissection — for a trajectory problem the separators are single stages — and factored by the
// multifrontal LDL^T of sparse.hip: log2(stages) launches instead of the chain of nx pivots of the blocked factorisation; the triangular solves
// run over the same tree.  `batch` >= the largest group this handle will lead (1 for a handle stepped alone).  Pivot signs (inertia) are those of
// the blocked factorisation (Sylvester); values agree to rounding, not bitwise (another elimination order).  on = 0 switches back.
// info (may be NULL) = [tree levels, rows of the largest front, nnz of the pattern of S (upper triangle), numeric phase (2 = multifrontal)].
int32_t calipso_hip_set_stage_parallel(calipso_hip_solver* s, int32_t on, int32_t batch, int64_t info[4]) {
    if (!s || batch < 1 || batch > MAX_BATCH) return CALIPSO_ERR_ARGUMENT;
    CK(hipSetDevice(s->device));
    CK(hipStreamSynchronize(s->stream));
    if (s->spS) { (void)calipso_hip_sparse_destroy(s->spS); s->spS = nullptr; }
    if (s->spS_src) { (void)hipFree(s->spS_src); s->spS_src = nullptr; }
    if (s->spS_inv) { (void)hipFree(s->spS_inv); s->spS_inv = nullptr; }
    if (s->d_reach) { (void)hipFree(s->d_reach); s->d_reach = nullptr; }
    s->stage_parallel = false;
    if (!on) { if (s->compact) { s->err = "calipso_hip_set_stage_parallel: a structured handle always factors through the multifrontal path"; return CALIPSO_ERR_ARGUMENT; } return CALIPSO_OK; }
    const int nx = s->d.nx, NP = s->d.NP;
    // where entry (row c >= column r) of the lower triangle of S lives: the dense S, or — structured handle — the tile of its segment pair (what
    // the skyline holds between the tiles is structurally zero: it points at the spare zero cell behind the last tile)
    std::map<std::pair<int, int>, const calipso::SegPair*> pair_of;
    if (s->compact) for (const calipso::SegPair& p : s->blocks.h_pairs) pair_of[{p.a, p.b}] = &p;
    auto s_offset = [&](int c, int r) -> long long {
        if (!s->compact) return (long long)c + (long long)r * NP;
        const int a = s->blocks.h_seg_of_col[(size_t)c], b = s->blocks.h_seg_of_col[(size_t)r];
        auto it = pair_of.find({a, b});
        if (it == pair_of.end()) return (long long)s->blocks_zero_cell;
        const calipso::Segment& sa = s->blocks.h_seg[(size_t)a]; const calipso::Segment& sb = s->blocks.h_seg[(size_t)b];
        return it->second->soff + (c - sa.c0) + (long long)(r - sb.c0) * sa.nc;
    };
    if ((int)s->h_reach.size() != nx) { s->err = "calipso_hip_set_stage_parallel: call calipso_hip_analyze_structure first"; return CALIPSO_ERR_ARGUMENT; }
    // upper-triangular CSC pattern (1-based) of the skyline: entry (r, c), r <= c, iff reach[r] >= c
    std::vector<int64_t> colptr((size_t)nx + 1, 0), rowval;
    for (int r = 0; r < nx; ++r) for (int c = r; c <= s->h_reach[(size_t)r]; ++c) colptr[(size_t)c + 1] += 1;
    colptr[0] = 1;
    for (int c = 0; c < nx; ++c) colptr[(size_t)c + 1] += colptr[(size_t)c];
    rowval.resize((size_t)(colptr[(size_t)nx] - 1));
    std::vector<long long> src(rowval.size());
    {
        std::vector<int64_t> next(colptr.begin(), colptr.end() - 1);
        for (int r = 0; r < nx; ++r)                                   // ascending r => rows sorted inside every column
            for (int c = r; c <= s->h_reach[(size_t)r]; ++c) {
                const int64_t at = next[(size_t)c]++ - 1;
                rowval[(size_t)at] = r + 1;
                src[(size_t)at] = s_offset(c, r);                     // S holds its lower triangle: entry (row c, column r)
            }
    }
    calipso_hip_sparse* sp = nullptr;
    int rc = calipso_hip_sparse_create(nx, colptr.data(), rowval.data(), 4, nullptr, s->device, &sp);
    int64_t desc[4] = {0, 0, 0, 0};
    if (rc == CALIPSO_OK && sp) sparse_describe(sp, desc);
    const int64_t max_front = 196;
    if (rc != CALIPSO_OK || !sparse_is_multifrontal(sp) || desc[1] > max_front) {    // (default: LDS-resident fronts only)
        s->err = rc != CALIPSO_OK ? std::string("calipso_hip_set_stage_parallel: ") + calipso_hip_sparse_last_error(sp)
                                  : std::string("calipso_hip_set_stage_parallel: a front of the dissection of S exceeds one CU's LDS (196 rows): the blocked factorisation stays");
        if (sp) (void)calipso_hip_sparse_destroy(sp);
        return rc != CALIPSO_OK ? rc : CALIPSO_ERR_ARGUMENT;
    }
    if (batch > 1 && (rc = calipso_hip_sparse_set_batch(sp, batch)) != CALIPSO_OK) { (void)calipso_hip_sparse_destroy(sp); return rc; }
    if ((rc = sparse_reserve_solve(sp, batch)) != CALIPSO_OK) { (void)calipso_hip_sparse_destroy(sp); return rc; }
    {
        hipError_t e = hipMalloc((void**)&s->spS_src, sizeof(long long) * std::max<size_t>(src.size(), 1));
        if (e == hipSuccess) e = hipMemcpy(s->spS_src, src.data(), sizeof(long long) * src.size(), hipMemcpyHostToDevice);
        if (e == hipSuccess && s->compact) {     // the inverse map: which pattern entry a cell of the packed S is (cells above the diagonal of a diagonal tile: none)
            std::vector<int> inv(s->blocks_zero_cell + 1, -1);       // (the packed S ends with the zero cell: api.hip)
            for (size_t q = 0; q < src.size(); ++q) if (src[q] != (long long)s->blocks_zero_cell) inv[(size_t)src[q]] = (int)q;
            e = hipMalloc((void**)&s->spS_inv, sizeof(int) * inv.size());
            if (e == hipSuccess) e = hipMemcpy(s->spS_inv, inv.data(), sizeof(int) * inv.size(), hipMemcpyHostToDevice);
        }
        if (e == hipSuccess) e = hipMalloc((void**)&s->d_reach, sizeof(int) * (size_t)std::max(nx, 1));
        if (e == hipSuccess) e = hipMemcpy(s->d_reach, s->h_reach.data(), sizeof(int) * (size_t)nx, hipMemcpyHostToDevice);
        if (e != hipSuccess) {                                          // nothing half-built stays behind
            (void)calipso_hip_sparse_destroy(sp);
            if (s->spS_src) { (void)hipFree(s->spS_src); s->spS_src = nullptr; }
            if (s->spS_inv) { (void)hipFree(s->spS_inv); s->spS_inv = nullptr; }
            if (s->d_reach) { (void)hipFree(s->d_reach); s->d_reach = nullptr; }
            return calipso::check(s, e, "calipso_hip_set_stage_parallel");
        }
    }
    sparse_borrow_stream(sp, s->stream);
    s->spS = sp; s->stage_parallel = true;
    if (info) sparse_describe(sp, info);
    return CALIPSO_OK;
}

}  // extern "C"
