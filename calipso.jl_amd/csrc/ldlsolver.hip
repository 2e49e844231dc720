// ldlsolver.hip — the LinearSolver seam of the reference (src/solver/linear_solver.jl:1-60) as a stand-alone device solver:
//   ldl_solver(A) / factorize!(s, A; update) / compute_inertia!(s) / linear_solve!(s, x, A, b; fact, update)
// for ANY sparse symmetric quasi-definite matrix the caller assembled itself (the reference hands its condensed K,
// `solver.data.jacobian_variables_symmetric`, to these functions from search_direction.jl:34, iterative_refinement.jl:25,
// differentiate.jl:19,45 and inertia.jl:23).  With this seam the reference's own unmodified search_direction! /
// iterative_refinement! / differentiate! run on top of the GPU factorisation: nothing but A's CSC arrays and the right-hand
// sides cross the boundary.
//
// As the reference's QDLDL (qdldl.jl:134-188): only triu(A) is read; no pivoting; D = diagonal of the LDL^T; the inertia is
// (#D > 0, #D <= 0, #D == 0); an exact zero pivot makes `positive` = -1.  The elimination order is the natural one (the x-block
// first): for a quasi-definite matrix every symmetric permutation has an LDL^T, and the dense fill does not depend on the order.
// The matrix is factored densely by the blocked LDL^T of ldl.hip (fp64 matrix cores) — a handle created for nx = n, ne = nc = 0
// carries exactly the buffers that needs (S, Dx, panel scratch, the 512-wide inverse blocks of the triangular solves).
#include <algorithm>
#include <vector>

#include "internal.hpp"
#include "device_utils.hpp"
#include "host_logic.hpp"

using namespace calipso;

// one workgroup per column c (1-based CSC as Julia's SparseMatrixCSC): entries (r, c) with r <= c go to the lower triangle of the
// column-major S at (c, r); entries below the diagonal are ignored (triu!, linear_solver.jl:23)
// With an elimination order installed (iperm != NULL, 0-based positions) the entry lands at the permuted position: P K P'.
__global__ __launch_bounds__(256) void k_scatter_csc_upper(int n, int NP, const long long* __restrict__ colptr, const long long* __restrict__ rowval,
                                                           const double* __restrict__ nzval, const int* __restrict__ iperm, double* __restrict__ S) {
    const int c = blockIdx.x;
    const long long p0 = colptr[c] - 1, p1 = colptr[c + 1] - 1;
    for (long long p = p0 + threadIdx.x; p < p1; p += blockDim.x) {
        const long long r = rowval[p] - 1;
        if (r <= c && r >= 0 && r < n) {
            const int pr = iperm ? iperm[r] : (int)r, pc = iperm ? iperm[c] : c;
            const int hi = pr > pc ? pr : pc, lo = pr > pc ? pc : pr;
            S[(size_t)hi + (size_t)lo * NP] = nzval[p];
        }
    }
}
// identity in the padding [n, NP)
__global__ void k_pad_diag(int n, int NP, double* __restrict__ S) {
    const int i = n + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < NP) S[(size_t)i + (size_t)i * NP] = 1.0;
}
// x = [b[perm]; 0]  (permute!, qdldl.jl:333) and back  out[perm] = x  (ipermute!, :349); perm == NULL: natural order
__global__ void k_pad_vec(const double* __restrict__ b, const int* __restrict__ perm, int n, int NP, double* __restrict__ x) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < NP) x[i] = i < n ? b[perm ? perm[i] : i] : 0.0;
}
__global__ void k_unpermute(const double* __restrict__ x, const int* __restrict__ perm, int n, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[perm ? perm[i] : i] = x[i];
}

struct LdlAux {   // device staging of the caller's CSC arrays (grown on demand), kept in the handle's side table
    long long* colptr = nullptr; long long* rowval = nullptr; double* nzval = nullptr; double* rhs = nullptr;
    size_t cap_nz = 0, cap_rhs = 0;
    int64_t inertia[3] = {0, 0, 0};
    bool factored = false;
    int *perm = nullptr, *iperm = nullptr;      // device, 0-based; NULL = natural order
    double* tmp = nullptr;                      // n doubles: un-permuted solution column
};
static LdlAux* aux_of(calipso_hip_solver* s, bool create) {
    if (!s->ldl_aux && create) s->ldl_aux = new LdlAux();
    return static_cast<LdlAux*>(s->ldl_aux);
}

namespace calipso {
void ldlsolver_release(calipso_hip_solver* s) {
    LdlAux* a = aux_of(s, false);
    if (!a) return;
    if (a->colptr) (void)hipFree(a->colptr);
    if (a->rowval) (void)hipFree(a->rowval);
    if (a->nzval) (void)hipFree(a->nzval);
    if (a->rhs) (void)hipFree(a->rhs);
    if (a->perm) (void)hipFree(a->perm);
    if (a->iperm) (void)hipFree(a->iperm);
    if (a->tmp) (void)hipFree(a->tmp);
    delete a;
    s->ldl_aux = nullptr;
}
}  // namespace calipso

extern "C" {

// ldl_solver(A)  linear_solver.jl:46-50 — a solver for n x n matrices on `device`
int32_t calipso_hip_ldl_create(int64_t n, int32_t device, calipso_hip_solver** out) {
    const int64_t none = 0;
    const int64_t ptr0[1] = {0};
    return calipso_hip_create(n, 0, 0, 0, 0, &none, 0, ptr0, &none, device, out);
}

// factorize!(s, A; update) + compute_inertia!(s)  linear_solver.jl:19-44
int32_t calipso_hip_ldl_factorize_csc(calipso_hip_solver* s, int64_t n, const int64_t* colptr, const int64_t* rowval, const double* nzval,
                                      int64_t inertia[3]) {
    if (!s || !colptr || (!rowval && colptr[n] > 1) || (!nzval && colptr[n] > 1)) return CALIPSO_ERR_ARGUMENT;
    const Dims& d = s->d;
    if (n != d.nx || d.ne != 0 || d.nc != 0 || s->compact) { s->err = "calipso_hip_ldl_factorize_csc: the handle was not created by calipso_hip_ldl_create for this n"; return CALIPSO_ERR_ARGUMENT; }
    if (!csc_pattern_ok(n, colptr, rowval)) { s->err = "calipso_hip_ldl_factorize_csc: colptr must be 1-based (Julia SparseMatrixCSC) and non-decreasing, rowval in 1..n"; return CALIPSO_ERR_ARGUMENT; }
    CK(hipSetDevice(s->device));
    LdlAux& a = *aux_of(s, true);
    const size_t nnz = (size_t)(colptr[n] - 1);
    if (nnz > a.cap_nz || !a.colptr) {
        // pointers are cleared as soon as they are freed and the capacity is raised only once both buffers exist: an allocation failure
        // in between leaves the staging empty, not dangling (ldlsolver_release frees whatever is non-null)
        if (a.rowval) { (void)hipFree(a.rowval); a.rowval = nullptr; }
        if (a.nzval) { (void)hipFree(a.nzval); a.nzval = nullptr; }
        a.cap_nz = 0;
        if (!a.colptr) CK(hipMalloc((void**)&a.colptr, sizeof(long long) * (size_t)(n + 1)));
        const size_t cap = std::max<size_t>(nnz, 1);
        CK(hipMalloc((void**)&a.rowval, sizeof(long long) * cap));
        CK(hipMalloc((void**)&a.nzval, sizeof(double) * cap));
        a.cap_nz = cap;
    }
    CK(hipMemcpyAsync(a.colptr, colptr, sizeof(long long) * (size_t)(n + 1), hipMemcpyHostToDevice, s->stream));
    if (nnz) {
        CK(hipMemcpyAsync(a.rowval, rowval, sizeof(long long) * nnz, hipMemcpyHostToDevice, s->stream));
        CK(hipMemcpyAsync(a.nzval, nzval, sizeof(double) * nnz, hipMemcpyHostToDevice, s->stream));
    }
    CK(hipMemsetAsync(s->S, 0, sizeof(double) * (size_t)d.NP * d.NP, s->stream));
    hipLaunchKernelGGL(k_scatter_csc_upper, dim3((unsigned)n), dim3(256), 0, s->stream, (int)n, d.NP, a.colptr, a.rowval, a.nzval, a.iperm, s->S);
    if (d.NP > n) hipLaunchKernelGGL(k_pad_diag, dim3((d.NP - (int)n + 255) / 256), dim3(256), 0, s->stream, (int)n, d.NP, s->S);
    fill_i(s, s->icount, 6, 0);       // (the device-side sign counters of ldl.hip are not used here)
    launch_ldl(s);
    // compute_inertia! on the host from D, exactly as linear_solver.jl:33-44 sees it: QDLDL_factor! stops at the first exact zero
    // pivot (qdldl.jl:456,579: positive_inertia = -1) and leaves the rest of D at the zeros it was reset to (qdldl.jl:444)
    s->hstage.resize((size_t)n);
    CK(hipMemcpyAsync(s->hstage.data(), s->Dx, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, s->stream));
    SYNC();
    s->stats.factorizations += 1;
    int rc = CALIPSO_OK;
    {
        int64_t pos = 0, nonpos = 0, zero = 0, k = 0;
        for (; k < n; ++k) {
            const double dk = s->hstage[(size_t)k];
            if (dk == 0.0) break;
            pos += dk > 0.0; nonpos += dk <= 0.0;
        }
        if (k < n) { zero = n - k; nonpos += n - k; pos = -1; rc = CALIPSO_WARN_ZERO_PIVOT; }
        a.inertia[0] = pos; a.inertia[1] = nonpos; a.inertia[2] = zero;
    }
    a.factored = true;
    if (inertia) { inertia[0] = a.inertia[0]; inertia[1] = a.inertia[1]; inertia[2] = a.inertia[2]; }
    return rc;
}

// Choose the elimination order of the following factorisations (qdldl.jl:134-143: perm = amd(A), iperm = invperm(perm)).
//   method 0 natural, 1 reverse Cuthill-McKee, 2 minimum degree, 4 nested dissection (ordering.hip), 3 the caller's `perm` (1-based, perm[k] = vertex eliminated k-th).
// perm (may be NULL for methods 0-2) receives / supplies the order.  info (may be NULL): [0] half bandwidth of P A P', [1] 64-row blocks per
// panel the device factorisation visits (0 = all: dense treatment), [2] nnz(L) of the sparse symbolic factor (-1: QDLDL_etree! failure),
// [3] nnz(triu A).  When the permuted matrix is banded the blocked LDL^T and the triangular solves skip everything outside the band.
int32_t calipso_hip_ldl_analyze_csc(calipso_hip_solver* s, int64_t n, const int64_t* colptr, const int64_t* rowval, int32_t method, int64_t* perm,
                                    int64_t info[4]) {
    if (!s || !colptr || method < 0 || method > 4 || (method == 3 && !perm)) return CALIPSO_ERR_ARGUMENT;
    if (!csc_pattern_ok(n, colptr, rowval)) { s->err = "calipso_hip_ldl_analyze_csc: colptr must start at 1 and be non-decreasing, rowval in 1..n"; return CALIPSO_ERR_ARGUMENT; }
    const Dims& d = s->d;
    if (n != d.nx || d.ne != 0 || d.nc != 0 || s->compact) { s->err = "calipso_hip_ldl_analyze_csc: the handle was not created by calipso_hip_ldl_create for this n"; return CALIPSO_ERR_ARGUMENT; }
    std::vector<int64_t> p((size_t)n);
    if (method == 3) std::copy(perm, perm + n, p.begin());
    else { const int rc = calipso_hip_ordering(n, colptr, rowval, method, p.data()); if (rc < 0) return rc; }
    int64_t sinfo[2] = {0, 0};
    const int64_t lnz = calipso_hip_symbolic(n, colptr, rowval, p.data(), nullptr, nullptr, nullptr, nullptr, nullptr, sinfo);
    if (lnz < -1) { s->err = "calipso_hip_ldl_analyze_csc: perm is not a permutation of 1:n"; return (int32_t)lnz; }
    CK(hipSetDevice(s->device));
    CK(hipStreamSynchronize(s->stream));
    LdlAux& a = *aux_of(s, true);
    std::vector<int> hp((size_t)n), hip_((size_t)n);
    bool natural = true;
    for (int64_t k = 0; k < n; ++k) { hp[(size_t)k] = (int)(p[(size_t)k] - 1); hip_[(size_t)(p[(size_t)k] - 1)] = (int)k; natural = natural && p[(size_t)k] == k + 1; }
    if (natural) {
        if (a.perm) { (void)hipFree(a.perm); a.perm = nullptr; }
        if (a.iperm) { (void)hipFree(a.iperm); a.iperm = nullptr; }
    } else {
        if (!a.perm) { CK(hipMalloc((void**)&a.perm, sizeof(int) * (size_t)n)); CK(hipMalloc((void**)&a.iperm, sizeof(int) * (size_t)n)); }
        CK(hipMemcpy(a.perm, hp.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
        CK(hipMemcpy(a.iperm, hip_.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice));
    }
    // band of the device factorisation (as structure.hip: 64-row blocks below a diagonal block that can be non-zero)
    const int nblk = d.NP / NB;
    const int hb = (int)sinfo[0];
    const int band64 = (hb + 63) / 64;
    const int new_band = band64 >= nblk - 1 ? 0 : std::max(1, band64);
    if (new_band != s->band64 || (new_band > 0 && hb != s->half_bandwidth)) {     // the launch sequences change with the band
        ldl_drop_graphs(s);
    }
    s->band64 = new_band; s->half_bandwidth = new_band > 0 ? hb : 0;
    // the block inverses of the triangular solves span whole diagonal blocks: what lies outside the band must read as zero
    CK(hipMemsetAsync(s->Tinv, 0, sizeof(double) * tinv_doubles(d.NP), s->stream));
    CK(hipStreamSynchronize(s->stream));
    a.factored = false;
    if (perm) std::copy(p.begin(), p.end(), perm);
    if (info) { info[0] = sinfo[0]; info[1] = s->band64; info[2] = lnz; info[3] = sinfo[1]; }
    return CALIPSO_OK;
}

// compute_inertia!(s)  linear_solver.jl:33-44 (of the last factorisation)
int32_t calipso_hip_ldl_inertia(calipso_hip_solver* s, int64_t inertia[3]) {
    if (!s || !inertia) return CALIPSO_ERR_ARGUMENT;
    const LdlAux* a = aux_of(s, false);
    if (!a || !a->factored) { s->err = "no factorisation yet"; return CALIPSO_ERR_ARGUMENT; }
    for (int k = 0; k < 3; ++k) inertia[k] = a->inertia[k];
    return CALIPSO_OK;
}

// linear_solve!(s, x, A, b; fact = false)  linear_solver.jl:52-60 / solve!(F, x) qdldl.jl:330-351 for nrhs right-hand sides
// (column-major n x nrhs; the Matrix method of linear_solver.jl:82-99 loops over the columns the same way)
int32_t calipso_hip_ldl_solve(calipso_hip_solver* s, int64_t n, int64_t nrhs, const double* b, double* x) {
    if (!s || !b || !x || nrhs < 0) return CALIPSO_ERR_ARGUMENT;
    const Dims& d = s->d;
    LdlAux* ap = aux_of(s, false);
    if (n != d.nx || !ap || !ap->factored) { s->err = "calipso_hip_ldl_solve: factorize first"; return CALIPSO_ERR_ARGUMENT; }
    LdlAux& a = *ap;
    CK(hipSetDevice(s->device));
    const size_t need = (size_t)n * (size_t)std::max<int64_t>(nrhs, 1);
    if (need > a.cap_rhs) {
        if (a.rhs) { (void)hipFree(a.rhs); a.rhs = nullptr; }
        a.cap_rhs = 0;
        CK(hipMalloc((void**)&a.rhs, sizeof(double) * need));
        a.cap_rhs = need;                       // only after the allocation succeeded
    }
    if (nrhs == 0) return CALIPSO_OK;
    CK(hipMemcpyAsync(a.rhs, b, sizeof(double) * (size_t)n * nrhs, hipMemcpyHostToDevice, s->stream));
    for (int64_t j = 0; j < nrhs; ++j) {
        hipLaunchKernelGGL(k_pad_vec, dim3((d.NP + 255) / 256), dim3(256), 0, s->stream, a.rhs + (size_t)j * n, a.perm, (int)n, d.NP, s->xbuf);
        launch_trsv(s, s->xbuf);
        hipLaunchKernelGGL(k_unpermute, dim3(((int)n + 255) / 256), dim3(256), 0, s->stream, s->xbuf, a.perm, (int)n, a.rhs + (size_t)j * n);
    }
    CK(hipMemcpyAsync(x, a.rhs, sizeof(double) * (size_t)n * nrhs, hipMemcpyDeviceToHost, s->stream));
    SYNC();
    return CALIPSO_OK;
}

}  // extern "C"
