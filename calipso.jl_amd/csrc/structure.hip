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

static int structure_clear(calipso_hip_solver* s) {
    s->band64 = 0; s->half_bandwidth = 0;
    const Dims& d = s->d;
    const size_t G = ((size_t)d.nx + 15) / 16;
    std::vector<int> kr(4 * G);
    for (size_t g = 0; g < G; ++g) { kr[4 * g] = 0; kr[4 * g + 1] = d.ne; kr[4 * g + 2] = 0; kr[4 * g + 3] = d.nc; }
    CK(hipSetDevice(s->device));
    CK(hipStreamSynchronize(s->stream));
    CK(hipMemcpy(s->krange, kr.data(), sizeof(int) * kr.size(), hipMemcpyHostToDevice));
    if (s->graph_ldl) { (void)hipGraphExecDestroy(s->graph_ldl); s->graph_ldl = nullptr; }      // the launch sequences change with the band
    if (s->graph_trsv) { (void)hipGraphExecDestroy(s->graph_trsv); s->graph_trsv = nullptr; }
    s->graph_ldl_tried = false; s->graph_trsv_tried = false;
    return CALIPSO_OK;
}

namespace calipso {
int structure_validate(calipso_hip_solver* s, int which) {
    const Dims& d = s->d;
    int* flag = s->icount + 60;
    CK(hipMemsetAsync(flag, 0, sizeof(int), s->stream));
    if (which == 0) {
        const size_t n = (size_t)d.nx * d.nx;
        hipLaunchKernelGGL(k_check_band, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s->stream, d.nx, s->half_bandwidth, s->Lxx, flag);
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
    const Dims& d = s->d;
    const int nx = d.nx, ne = d.ne, nc = d.nc, m = d.m;
    CK(hipSetDevice(s->device));
    CK(hipStreamSynchronize(s->stream));
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
    }
    s->half_bandwidth = (int)hb;
    s->band64 = band64 >= nblk - 1 ? 0 : std::max(1, band64);  // 0: nothing to skip
    // entries of S outside the band are never written in banded mode and must read as zero (the block inverses span whole
    // 512 x 512 diagonal blocks): clear what an earlier dense factorisation may have left there
    CK(hipMemsetAsync(s->S, 0, sizeof(double) * (size_t)d.NP * d.NP, s->stream));
    CK(hipMemsetAsync(s->Tinv, 0, sizeof(double) * (d.NP < 512 ? (size_t)d.NP * d.NP : (size_t)(d.NP / 512) * 512 * 512), s->stream));
    CK(hipStreamSynchronize(s->stream));
    if (s->graph_ldl) { (void)hipGraphExecDestroy(s->graph_ldl); s->graph_ldl = nullptr; }      // the launch sequences change with the band
    if (s->graph_trsv) { (void)hipGraphExecDestroy(s->graph_trsv); s->graph_trsv = nullptr; }
    s->graph_ldl_tried = false; s->graph_trsv_tried = false;
    if (out) { out[0] = hb; out[1] = s->band64; out[2] = G ? (int64_t)(ve / G) : 0; out[3] = G ? (int64_t)(vc / G) : 0; }
    return CALIPSO_OK;
}

// back to the dense treatment (e.g. before uploading blocks with a different pattern)
int32_t calipso_hip_clear_structure(calipso_hip_solver* s) {
    if (!s) return CALIPSO_ERR_ARGUMENT;
    return structure_clear(s);
}

}  // extern "C"
