// scatter.hip — the tail of evaluate! on the device (src/solver/evaluate.jl:37-121).
//
// The reference's generated functions fill a CACHE of non-zero values; `methods.<field>_sparsity` lists their (row, col) positions and
// evaluate! scatters them into the dense ProblemData matrices by plain assignment,
//        for (i, idx) in enumerate(sparsity)   problem.<field>[idx...] = cache[i]   end
// in list order.  The trajectory layer emits the same (row, col) more than once (consecutive stages share the (X_t+1, X_t+1) block,
// src/trajectory_optimization/dynamics.jl:245-260, methods.jl:24-27), so the LAST writer of an entry wins.
// Here the index lists are registered once (calipso_hip_set_sparsity: the winners are determined on the host, the destination offsets
// live on the device) and every evaluation uploads only the value caches — O(nnz) over PCIe instead of the dense nx^2 / m nx blocks —
// and scatters them with the reference's semantics:
//   Jacobians     equality_jacobian_variables / cone_jacobian_variables: assignment into the stacked Jacobian (entries outside the
//                 list keep their value: zero since creation, as in the reference's zeros(...) matrices)
//   Hessian       the three matrices objective_jacobian_variables_variables, equality_dual_..._variables_variables and
//                 cone_dual_..._variables_variables are each assigned from their own cache and then SUMMED into the one Lagrangian
//                 Hessian the device holds (residual_jacobian_variables.jl:10-16; the tensor terms iff options.constraint_tensor)
#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "internal.hpp"
#include "host_logic.hpp"

using namespace calipso;

namespace {

struct Pattern {
    int64_t count = 0;               // length of the caller's cache
    int64_t winners = 0;             // entries that survive the assignment order
    int* src = nullptr;              // device: index into the cache of winner w
    long long* dst = nullptr;        // device: offset (doubles) into the destination matrix of winner w
    long long* dst2 = nullptr;       // structured handles: the same entry in the row-major copy of its block (dst: the column-major copy)
    double* vals = nullptr;          // device staging of the cache (kept: the last values of a part are re-used when a later call does not pass it)
    bool loaded = false;             // vals holds values of an earlier scatter
};
struct ScatterAux { std::map<std::string, Pattern> pat; };

ScatterAux* aux_of(calipso_hip_solver* s, bool create) {
    if (!s->scatter_aux && create) s->scatter_aux = new ScatterAux();
    return static_cast<ScatterAux*>(s->scatter_aux);
}

__global__ void k_scatter_assign(int64_t n, const int* __restrict__ src, const long long* __restrict__ dst, const double* __restrict__ vals, double* __restrict__ M) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w < n) M[dst[w]] = vals[src[w]];
}
__global__ void k_scatter_add(int64_t n, const int* __restrict__ src, const long long* __restrict__ dst, const double* __restrict__ vals, double* __restrict__ M) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w < n) M[dst[w]] += vals[src[w]];        // the winners of ONE list are distinct entries: no two threads touch the same address
}

const char* const HESSIAN_PARTS[3] = {"objective_jacobian_variables_variables", "equality_dual_jacobian_variables_variables",
                                      "cone_dual_jacobian_variables_variables"};

void free_pattern(Pattern& p) {
    if (p.src) (void)hipFree(p.src);
    if (p.dst) (void)hipFree(p.dst);
    if (p.dst2) (void)hipFree(p.dst2);
    if (p.vals) (void)hipFree(p.vals);
    p = Pattern();
}

}  // namespace

namespace calipso {
void scatter_release(calipso_hip_solver* s) {
    ScatterAux* a = aux_of(s, false);
    if (!a) return;
    for (auto& kv : a->pat) free_pattern(kv.second);
    delete a;
    s->scatter_aux = nullptr;
}
}  // namespace calipso

extern "C" {

// methods.<field>_sparsity: `count` (row, col) pairs, 1-based, in the order of the value cache; duplicates allowed (the last one wins).
// field: one of the three Hessian parts above, "equality_jacobian_variables", "cone_jacobian_variables".  count = 0 removes the pattern.
int32_t calipso_hip_set_sparsity(calipso_hip_solver* s, const char* field, int64_t count, const int64_t* rows, const int64_t* cols) {
    if (!s || !field || count < 0 || (count > 0 && (!rows || !cols))) return CALIPSO_ERR_ARGUMENT;
    const Dims& d = s->d;
    const std::string f = field;
    int64_t nr = 0, ld = 0, off = 0;
    if (f == HESSIAN_PARTS[0] || f == HESSIAN_PARTS[1] || f == HESSIAN_PARTS[2]) { nr = d.nx; ld = d.nx; }
    else if (f == "equality_jacobian_variables") { nr = d.ne; ld = d.m; }
    else if (f == "cone_jacobian_variables") { nr = d.nc; ld = d.m; off = d.ne; }
    else { s->err = std::string("calipso_hip_set_sparsity: no sparsity for field ") + field; return CALIPSO_ERR_ARGUMENT; }
    for (int64_t p = 0; p < count; ++p)
        if (rows[p] < 1 || rows[p] > nr || cols[p] < 1 || cols[p] > d.nx) { s->err = "calipso_hip_set_sparsity: index out of range"; return CALIPSO_ERR_ARGUMENT; }
    CK(hipSetDevice(s->device));
    CK(hipStreamSynchronize(s->stream));
    ScatterAux* a = aux_of(s, true);
    Pattern& pat = a->pat[f];
    free_pattern(pat);
    if (count == 0) { a->pat.erase(f); return CALIPSO_OK; }
    // winners: for every distinct (row, col) the LAST position in the list (assignment order of evaluate.jl:40-42 etc.)
    std::map<std::pair<int64_t, int64_t>, int64_t> last;
    for (int64_t p = 0; p < count; ++p) last[{rows[p], cols[p]}] = p;
    std::vector<int> src; std::vector<long long> dst, dst2;
    src.reserve(last.size()); dst.reserve(last.size());
    for (const auto& kv : last) {
        src.push_back((int)kv.second);
        if (s->compact) {       // structured handle: straight into the two packed copies of the entry's block
            long long oc = 0, orr = 0;
            const int which = f == "equality_jacobian_variables" ? 1 : (f == "cone_jacobian_variables" ? 2 : 0);
            if (!blocks_entry_offsets(s, which, (int)(off + kv.first.first - 1), (int)(kv.first.second - 1), &oc, &orr)) {
                a->pat.erase(f);
                s->err = "calipso_hip_set_sparsity: an entry lies outside the structure declared at calipso_hip_create_structured";
                return CALIPSO_ERR_ARGUMENT;
            }
            dst.push_back(oc); dst2.push_back(orr);
        } else dst.push_back((long long)(off + kv.first.first - 1) + (long long)(kv.first.second - 1) * ld);
    }
    pat.count = count; pat.winners = (int64_t)src.size();
    if (s->compact) {
        CK(hipMalloc((void**)&pat.dst2, sizeof(long long) * dst2.size()));
        CK(hipMemcpy(pat.dst2, dst2.data(), sizeof(long long) * dst2.size(), hipMemcpyHostToDevice));
    }
    CK(hipMalloc((void**)&pat.src, sizeof(int) * src.size()));
    CK(hipMalloc((void**)&pat.dst, sizeof(long long) * dst.size()));
    CK(hipMalloc((void**)&pat.vals, sizeof(double) * (size_t)count));
    CK(hipMemcpy(pat.src, src.data(), sizeof(int) * src.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(pat.dst, dst.data(), sizeof(long long) * dst.size(), hipMemcpyHostToDevice));
    return CALIPSO_OK;
}

// problem.<field>[idx...] = cache[i] for a Jacobian field (values: the cache, `count` as registered)
int32_t calipso_hip_scatter_field(calipso_hip_solver* s, const char* field, const double* values, int64_t count) {
    if (!s || !field || !values) return CALIPSO_ERR_ARGUMENT;
    const std::string f = field;
    if (f != "equality_jacobian_variables" && f != "cone_jacobian_variables") { s->err = "calipso_hip_scatter_field: Jacobian fields only (the Hessian parts go through calipso_hip_scatter_hessian)"; return CALIPSO_ERR_ARGUMENT; }
    ScatterAux* a = aux_of(s, false);
    if (!a || !a->pat.count(f) || a->pat[f].count != count) { s->err = std::string("calipso_hip_scatter_field: no sparsity of that length registered for ") + field; return CALIPSO_ERR_ARGUMENT; }
    Pattern& pat = a->pat[f];
    CK(hipSetDevice(s->device));
    CK(hipMemcpyAsync(pat.vals, values, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, s->stream));
    if (s->compact) {
        hipLaunchKernelGGL(k_scatter_assign, dim3((unsigned)((pat.winners + 255) / 256)), dim3(256), 0, s->stream, pat.winners, pat.src, pat.dst, pat.vals, s->Lsym);
        hipLaunchKernelGGL(k_scatter_assign, dim3((unsigned)((pat.winners + 255) / 256)), dim3(256), 0, s->stream, pat.winners, pat.src, pat.dst2, pat.vals, s->Lsym);
    } else
    hipLaunchKernelGGL(k_scatter_assign, dim3((unsigned)((pat.winners + 255) / 256)), dim3(256), 0, s->stream, pat.winners, pat.src, pat.dst, pat.vals, s->Z);
    if (structure_active(s)) { const int rc = structure_validate(s, f == "equality_jacobian_variables" ? 1 : 2); if (rc < 0) return rc; }
    blocks_pack(s, true, false);
    SYNC();      // `values` may be reused by the caller
    return CALIPSO_OK;
}

// Lagrangian Hessian = assign(objective part) + assign(equality-dual part) + assign(cone-dual part).  A NULL cache means "not part of this
// evaluate! call": as in the reference, where each of the three stored matrices keeps its previous values when its flag is not set
// (evaluate.jl:37-42) and residual_jacobian_variables.jl:10-16 always sums all three, the part's LAST uploaded values are added again from the
// device-resident cache; a part that has never been scattered (or has no registered sparsity) contributes zeros.  counts as registered.
int32_t calipso_hip_scatter_hessian(calipso_hip_solver* s, const double* objective_values, int64_t n_objective, const double* equality_dual_values,
                                    int64_t n_equality_dual, const double* cone_dual_values, int64_t n_cone_dual) {
    if (!s) return CALIPSO_ERR_ARGUMENT;
    const double* vals[3] = {objective_values, equality_dual_values, cone_dual_values};
    const int64_t cnt[3] = {n_objective, n_equality_dual, n_cone_dual};
    ScatterAux* a = aux_of(s, false);
    for (int k = 0; k < 3; ++k)
        if (vals[k] && (!a || !a->pat.count(HESSIAN_PARTS[k]) || a->pat[HESSIAN_PARTS[k]].count != cnt[k])) {
            s->err = std::string("calipso_hip_scatter_hessian: no sparsity of that length registered for ") + HESSIAN_PARTS[k];
            return CALIPSO_ERR_ARGUMENT;
        }
    const Dims& d = s->d;
    CK(hipSetDevice(s->device));
    if (s->compact) {           // the Hessian blocks (contiguous behind the Z blocks in the packed region)
        const StageBlocks& B = s->blocks;
        const long long lo = B.h_lblk.front().off_c, hi = B.h_lblk.back().off_r + (long long)B.h_lblk.back().n * B.h_lblk.back().n;
        CK(hipMemsetAsync(s->Lsym + lo, 0, sizeof(double) * (size_t)(hi - lo), s->stream));
    } else
    CK(hipMemsetAsync(s->Lxx, 0, sizeof(double) * (size_t)d.nx * d.nx, s->stream));     // the three dense matrices of the reference are zero outside their lists
    for (int k = 0; k < 3; ++k) {
        if (!a || !a->pat.count(HESSIAN_PARTS[k])) continue;
        Pattern& pat = a->pat[HESSIAN_PARTS[k]];
        if (vals[k]) { CK(hipMemcpyAsync(pat.vals, vals[k], sizeof(double) * (size_t)cnt[k], hipMemcpyHostToDevice, s->stream)); pat.loaded = true; }
        if (!pat.loaded || pat.winners == 0) continue;
        if (s->compact) {
            hipLaunchKernelGGL(k_scatter_add, dim3((unsigned)((pat.winners + 255) / 256)), dim3(256), 0, s->stream, pat.winners, pat.src, pat.dst, pat.vals, s->Lsym);
            hipLaunchKernelGGL(k_scatter_add, dim3((unsigned)((pat.winners + 255) / 256)), dim3(256), 0, s->stream, pat.winners, pat.src, pat.dst2, pat.vals, s->Lsym);
        } else
        hipLaunchKernelGGL(k_scatter_add, dim3((unsigned)((pat.winners + 255) / 256)), dim3(256), 0, s->stream, pat.winners, pat.src, pat.dst, pat.vals, s->Lxx);
    }
    s->hessian_dirty = true;
    if (structure_active(s)) { const int rc = structure_validate(s, 0); if (rc < 0) return rc; }
    blocks_pack(s, false, true);
    SYNC();
    return CALIPSO_OK;
}

}  // extern "C"
