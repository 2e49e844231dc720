// api.hip — the C ABI of include/calipso_hip.h and the host-side driver of solve! (src/solver/solve.jl:8-377),
// inertia_correction! (inertia.jl:30-80), iterative_refinement! (iterative_refinement.jl:1-52) and the filter
// (filter.jl:1-89) over the HIP kernels of this directory.  The host only takes the scalar decisions the reference
// takes (convergence tests, line-search acceptance, regularisation updates); all array arithmetic is on the device.
#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "internal.hpp"
#include "device_utils.hpp"
#include "host_logic.hpp"

using namespace calipso;

namespace calipso {
int check(H* s, hipError_t e, const char* what) {
    char buf[512];
    snprintf(buf, sizeof buf, "%s: %s", what, hipGetErrorString(e));
    if (s) s->err = buf;
    return CALIPSO_ERR_HIP;
}
bool lds_attribute(const void* kernel, int bytes) {
    static std::mutex mu;
    static std::map<std::pair<const void*, int>, std::pair<int, int>> done;      // (kernel, device) -> (largest size granted, smallest size refused; 0 / INT_MAX: none yet)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return false;
    std::lock_guard<std::mutex> lock(mu);
    auto& e = done.emplace(std::make_pair(kernel, dev), std::make_pair(0, 0x7fffffff)).first->second;
    if (bytes <= e.first) return true;           // a size this large was granted before
    if (bytes >= e.second) return false;         // a size this small was refused before
    const bool ok = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
    if (ok) e.first = bytes; else { e.second = bytes; (void)hipGetLastError(); }
    return ok;
}
// CALIPSO_HIP_FAULT_INJECT=launch (tests/test_gpu_robustness.py): a launch the runtime must refuse (1 MB of dynamic LDS) in the middle of the factorisation's launches —
// what a kernel whose configuration the device cannot serve looks like to the host: nothing is returned, the error sits in hipGetLastError (host_logic.hpp: launch_errors)
__global__ void k_fault_probe(int* p) { if (p) *p = 1; }
void inject_refused_launch(hipStream_t stream) {
    static const bool inject = [] { const char* f = getenv("CALIPSO_HIP_FAULT_INJECT"); return f && strstr(f, "launch") != nullptr; }();
    if (inject) hipLaunchKernelGGL(k_fault_probe, dim3(1), dim3(64), 1u << 20, stream, (int*)nullptr);
}
}  // namespace calipso

static std::string g_create_err;


template <typename T>
static int dalloc(H* s, T** p, size_t count) {
    if (count == 0) count = 1;
    CK(hipMalloc((void**)p, count * sizeof(T)));
    // zero-fill ON THE HANDLE'S STREAM: the streams are non-blocking, so a hipMemset on the null stream would not be ordered with
    // the uploads / kernels that follow on s->stream and could land after them
    CK(hipMemsetAsync(*p, 0, count * sizeof(T), s->stream));
    return 0;
}

static int fail_arg(H* s, const std::string& msg) { s->err = msg; return CALIPSO_ERR_ARGUMENT; }

extern "C" {

const char* calipso_hip_version(void) { return "calipso-hip 0.1 (gfx950)"; }

int32_t calipso_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char* calipso_hip_last_error(H* s) { return s ? s->err.c_str() : g_create_err.c_str(); }

// the declared structure of a structured handle (calipso_hip_create_structured): 1-based, as the caller's sparsity lists are
struct StructureSpec { const int64_t* row_first; const int64_t* row_last; int64_t n_blocks; const int64_t* block_start; };

static int32_t create_impl(int64_t nx, int64_t np, int64_t ne, int64_t nc, int64_t n_nonneg, const int64_t* nonneg_idx, int64_t n_soc,
                           const int64_t* soc_ptr, const int64_t* soc_idx, int32_t device, const StructureSpec* spec, H** out) {
    if (!out) return CALIPSO_ERR_ARGUMENT;
    *out = nullptr;
    if (nx < 1 || np < 0 || ne < 0 || nc < 0 || n_nonneg < 0 || n_soc < 0) { g_create_err = "negative or empty dimension"; return CALIPSO_ERR_ARGUMENT; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_create_err = "no HIP device available (libcalipso_hip has no CPU path)"; return CALIPSO_ERR_HIP; }
    if (device < 0 || device >= ndev) { g_create_err = "device ordinal out of range"; return CALIPSO_ERR_ARGUMENT; }
    // cone layout contract: nonnegative = 1..q, then contiguous second-order blocks in order, covering 1..nc
    // (the only layout for which cones/cone.jl:27-59's vcat order agrees with residual.jl:46-48 etc.)
    for (int64_t i = 0; i < n_nonneg; ++i)
        if (nonneg_idx[i] != i + 1) { g_create_err = "nonnegative_indices must be 1:q"; return CALIPSO_ERR_LAYOUT; }
    H* s = new H();
    int64_t next = n_nonneg + 1;
    int woff = 0, maxd = 0;
    for (int64_t j = 0; j < n_soc; ++j) {
        const int64_t len = soc_ptr[j + 1] - soc_ptr[j];
        if (len == 0) continue;   // Indices allows empty vectors (indices.jl:22)
        for (int64_t p = 0; p < len; ++p)
            if (soc_idx[soc_ptr[j] + p] != next + p) { delete s; g_create_err = "second_order_indices must be contiguous blocks following the nonnegative entries"; return CALIPSO_ERR_LAYOUT; }
        if (len > MAX_SOC_DIM) { delete s; g_create_err = "second-order cone dimension above the supported maximum (1024)"; return CALIPSO_ERR_ARGUMENT; }
        s->h_soc_start.push_back((int)(next - 1));
        s->h_soc_dim.push_back((int)len);
        s->h_soc_woff.push_back(woff);
        woff += (int)(len * len);
        maxd = std::max<int>(maxd, (int)len);
        next += len;
    }
    if (next - 1 != nc) { delete s; g_create_err = "cone index sets do not cover 1:num_cone"; return CALIPSO_ERR_LAYOUT; }
    s->h_nonneg.assign(nonneg_idx, nonneg_idx + n_nonneg);
    s->h_soc_ptr.assign(soc_ptr, soc_ptr + n_soc + 1);
    s->h_soc_idx.assign(soc_idx, soc_idx + (n_soc ? soc_ptr[n_soc] : 0));
    Dims& d = s->d;
    d.nx = (int)nx; d.np = (int)np; d.ne = (int)ne; d.nc = (int)nc;
    d.n = d.nx + d.ne + d.nc;                 // dimensions.jl:35
    d.N = d.nx + 2 * d.ne + 3 * d.nc;         // dimensions.jl:22-23
    d.m = d.ne + d.nc;
    d.q = (int)n_nonneg; d.n_soc = (int)s->h_soc_start.size(); d.max_dim = maxd;
    d.n_wide = 0;
    for (int dim_j : s->h_soc_dim) d.n_wide += dim_j > 4;
    // nx padded: a power-of-two multiple of 64 up to 512 (small systems: one solve block), multiples of 512 above
    if (d.nx <= 512) { d.NP = 64; while (d.NP < d.nx) d.NP *= 2; } else d.NP = ((d.nx + 511) / 512) * 512;   // multiple of the triangular-solve block (and of TILE, NB)
    s->device = device;
    *out = s;   // from here on errors are reported through the handle
    CK(hipSetDevice(device));
    {
        // Instances are meant to run concurrently (one stream each).  Streams of equal priority may be multiplexed onto one
        // hardware queue, which serialises them; alternating the priority class spreads consecutive handles over queues.
        static std::atomic<int> n_handles{0};
        int least = 0, greatest = 0;
        CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        const int k = n_handles.fetch_add(1);
        int span = least - greatest;   // numerically lower = higher priority
        const int prio = span > 0 ? greatest + (k % (span + 1)) : least;
        CK(hipStreamCreateWithPriority(&s->stream, hipStreamNonBlocking, prio));
    }
    for (auto& e : s->ev) CK(hipEventCreate(&e));
    const size_t NX = d.nx, NE = d.ne, NC = d.nc, N = d.N, NPd = d.NP, M = d.m, n = d.n;
    int rc = 0;
    // Structured handle: the block tables come first (host work only), because the slab is sized from them — no dense Lxx / [gx; hx] / S / Tinv exist
    calipso::BlockPlan plan;
    if (spec) {
        std::vector<int> zr(2 * (size_t)d.m, 0), lr((size_t)d.nx, 0);
        for (int k = 0; k < d.m; ++k) {
            const int64_t f = spec->row_first[k], l = spec->row_last[k];
            if (l >= f && (f < 1 || l > nx)) { s->err = "calipso_hip_create_structured: column range of a constraint row outside 1..nx"; return CALIPSO_ERR_ARGUMENT; }
            zr[2 * (size_t)k] = l >= f ? (int)(f - 1) : 0; zr[2 * (size_t)k + 1] = l >= f ? (int)l : 0;
        }
        for (int j = 0; j < d.n_soc; ++j) {          // the rows of a second-order cone are coupled through its weight block: they share the union of their ranges
            const int st = d.ne + s->h_soc_start[j], dim = s->h_soc_dim[j];
            int lo = d.nx, hi = 0;
            for (int k = st; k < st + dim; ++k) if (zr[2 * (size_t)k + 1] > zr[2 * (size_t)k]) { lo = std::min(lo, zr[2 * (size_t)k]); hi = std::max(hi, zr[2 * (size_t)k + 1]); }
            for (int k = st; k < st + dim; ++k) { zr[2 * (size_t)k] = hi > lo ? lo : 0; zr[2 * (size_t)k + 1] = hi > lo ? hi : 0; }
        }
        if (spec->n_blocks < 1 || spec->block_start[0] != 1) { s->err = "calipso_hip_create_structured: hessian_block_start must begin with 1"; return CALIPSO_ERR_ARGUMENT; }
        for (int64_t b = 0; b < spec->n_blocks; ++b) {
            const int64_t c0 = spec->block_start[b], c1 = b + 1 < spec->n_blocks ? spec->block_start[b + 1] : nx + 1;
            if (c1 <= c0 || c1 > nx + 1) { s->err = "calipso_hip_create_structured: hessian_block_start must be increasing within 1..nx"; return CALIPSO_ERR_ARGUMENT; }
            for (int64_t c = c0; c < c1; ++c) lr[(size_t)c - 1] = (int)c1 - 2;
        }
        s->h_zrow = zr; s->h_lreach = lr;
        // skyline of S (as calipso_hip_analyze_structure derives it from a numeric pattern)
        std::vector<int>& reach = s->h_reach;
        reach = lr;
        std::vector<int> ext((size_t)d.nx, -1);
        for (int k = 0; k < d.m; ++k) if (zr[2 * (size_t)k + 1] > zr[2 * (size_t)k]) ext[(size_t)zr[2 * (size_t)k]] = std::max(ext[(size_t)zr[2 * (size_t)k]], zr[2 * (size_t)k + 1] - 1);
        int run = -1;
        for (int j = 0; j < d.nx; ++j) { run = std::max(run, ext[(size_t)j]); if (run >= j) reach[(size_t)j] = std::max(reach[(size_t)j], run); }
        if (!calipso::blocks_plan(d, zr, lr, plan, s->err)) return CALIPSO_ERR_ARGUMENT;
        s->compact = true;
        s->blocks_zero_cell = plan.spacked - 1;
    }
    const bool cp = s->compact;
    // Every per-instance buffer is carved out of ONE slab (256-byte aligned pieces, same order for every handle of a shape), so
    // that two handles of the same shape differ by a single pointer offset — what lets a group step them through the same
    // launches (internal.hpp: Batch).  Shape-only data (tile list, cone index arrays) is outside the slab.
    std::vector<std::pair<double**, size_t>> carve;
    auto SL = [&](double** pp, size_t count) { carve.push_back({pp, count ? count : 1}); };
    double* icount_d = nullptr;
    SL(&s->Lxx, cp ? 1 : NX * NX); SL(&s->Lsym, cp ? plan.packed : NX * NX); SL(&s->Z, cp ? 1 : M * NX);     // structured: Lsym is the region of the packed blocks
    SL(&s->fx, NX); SL(&s->gyx, NX); SL(&s->hzx, NX); SL(&s->gh, M);
    SL(&s->cone_product, NC); SL(&s->cone_target, NC); SL(&s->barrier_gradient, NC);
    SL(&s->dscal, 64); SL(&s->refpart, std::max((NE + NC + 255) / 256 + 1, M + 2) + (size_t)d.n_wide);
    SL(&s->solution, N); SL(&s->candidate, N); SL(&s->lambda, NE); SL(&s->parameters, (size_t)d.np);
    SL(&s->residual, N); SL(&s->residual_error, N); SL(&s->step, N); SL(&s->step_correction, N);
    SL(&s->saved_point, N); SL(&s->saved_g, NE); SL(&s->saved_h, NC);
    SL(&s->residual_symmetric, n); SL(&s->step_symmetric, n); SL(&s->merit_gradient, n);
    SL(&s->S, cp ? plan.spacked : NPd * NPd); SL(&s->Dx, NPd); SL(&s->Ypanel, cp ? 1 : NPd * NB);                     // structured: S = the tiles of the segment pairs, contiguous
    SL(&s->Tinv, cp ? 1 : calipso::tinv_doubles(d.NP)); SL(&s->Ttmp, cp ? 1 : NPd * 1024); SL(&s->zf2, NPd); SL(&s->WH, cp ? 1 : NC * NX);
    SL(&s->Wfac, (cp || !calipso::wform_layout_ok(d.NP, calipso::trsv_block(d.NP, 512))) ? 1 : NPd * NPd / 2);     // W-form blocks of the solves: < NP^2 / 2 doubles for every solve-block width
    SL(&s->wz, NC); SL(&s->kzz, NC);
    SL(&s->Wsoc, (size_t)woff); SL(&s->Bsoc, (size_t)woff); SL(&s->socwork, (size_t)2 * woff);
    SL(&icount_d, 32);                                        // 64 ints
    double* krange_d = nullptr;
    const size_t KG = (NX + 15) / 16;                         // 16-column groups (structure.hip)
    SL(&krange_d, 2 * KG);                                    // 4 ints per group
    double* zrow_d = nullptr;
    SL(&zrow_d, M);                                           // 2 ints per row of [gx; hx]
    const size_t maxdim = std::max(std::max(NX, M), NPd);   // rows of the largest mat-vec (the stacked Jacobian has m = ne + nc rows)
    SL(&s->gemv_partial, cp ? 1 : std::max(64 * maxdim, ((NX + 15) / 16) * M + ((M + 1023) / 1024) * NX));   // gemv_n chunks / gemv_both partials (the block mat-vecs need none)
    SL(&s->vtmp, 4 * std::max(N, NPd));
    SL(&s->xbuf, NPd); SL(&s->zf, NPd); SL(&s->t1, M); SL(&s->t2, M);
    SL(&s->zsx, M); SL(&s->w1, NX); SL(&s->w2, NX); SL(&s->lxv, NX);
    SL(&s->lgp, NX * d.np); SL(&s->gp, NE * d.np); SL(&s->hp, NC * d.np);
    SL(&s->jacobian_parameters, N * d.np); SL(&s->solution_sensitivity, N * d.np);
    SL(&s->qp.q, NX); SL(&s->qp.bh, M);
    size_t total = 0;
    for (auto& c : carve) total += (c.second + 31) & ~(size_t)31;
    CK(hipMalloc((void**)&s->slab, total * sizeof(double)));
    CK(hipMemsetAsync(s->slab, 0, total * sizeof(double), s->stream));   // (on the handle's stream, see dalloc)
    s->slab_doubles = total;
    {
        size_t off = 0;
        for (auto& c : carve) { *c.first = s->slab + off; off += (c.second + 31) & ~(size_t)31; }
    }
    s->icount = reinterpret_cast<int*>(icount_d);
    s->krange = reinterpret_cast<int*>(krange_d);
    s->zrow = reinterpret_cast<int*>(zrow_d);
    s->gx = s->Z; s->hx = s->Z + NE; s->g = s->gh; s->hc = s->gh + NE;
    s->Lf = s->S;
    schur_plan(s);
    rc |= dalloc(s, &s->cone.soc_start, (size_t)d.n_soc); rc |= dalloc(s, &s->cone.soc_dim, (size_t)d.n_soc);
    rc |= dalloc(s, &s->cone.soc_woff, (size_t)d.n_soc); rc |= dalloc(s, &s->cone.entry_soc, NC);
    rc |= dalloc(s, &s->cone.wide, (size_t)d.n_wide);
    std::vector<int> zgrp;
    calipso::solve_tail_plan(d, s->h_soc_start, s->h_soc_dim, zgrp);
    if (zgrp.size() >= 2) { rc |= dalloc(s, &s->zgrp, zgrp.size()); s->n_zgrp = (int)zgrp.size() - 1; }
    rc |= dalloc(s, &s->gate, (size_t)4);
    if (rc) return CALIPSO_ERR_HIP;
    CK(hipHostMalloc((void**)&s->hscal, 64 * sizeof(double), hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostMalloc((void**)&s->hicount, 64 * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostMalloc((void**)&s->hseq, 64, hipHostMallocMapped | hipHostMallocCoherent));
    CK(hipHostGetDevicePointer((void**)&s->hscal_dev, s->hscal, 0));
    CK(hipHostGetDevicePointer((void**)&s->hicount_dev, s->hicount, 0));
    CK(hipHostGetDevicePointer((void**)&s->hseq_dev, s->hseq, 0));
    s->hseq[0] = 0;
    CK(hipStreamSynchronize(s->stream));   // the zero-fills above are complete before the (null-stream) index uploads below
    {   // dense default of the structure table: every column group visits all constraint rows
        std::vector<int> kr(4 * KG);
        for (size_t g = 0; g < KG; ++g) { kr[4 * g] = 0; kr[4 * g + 1] = d.ne; kr[4 * g + 2] = 0; kr[4 * g + 3] = d.nc; }
        CK(hipMemcpy(s->krange, kr.data(), sizeof(int) * kr.size(), hipMemcpyHostToDevice));
    }
    if (s->zgrp) CK(hipMemcpy(s->zgrp, zgrp.data(), sizeof(int) * zgrp.size(), hipMemcpyHostToDevice));
    if (d.n_soc) {
        CK(hipMemcpy(s->cone.soc_start, s->h_soc_start.data(), sizeof(int) * d.n_soc, hipMemcpyHostToDevice));
        CK(hipMemcpy(s->cone.soc_dim, s->h_soc_dim.data(), sizeof(int) * d.n_soc, hipMemcpyHostToDevice));
        CK(hipMemcpy(s->cone.soc_woff, s->h_soc_woff.data(), sizeof(int) * d.n_soc, hipMemcpyHostToDevice));
    }
    if (d.n_wide) {
        std::vector<int> wide;
        for (int j = 0; j < d.n_soc; ++j) if (s->h_soc_dim[j] > 4) wide.push_back(j);
        CK(hipMemcpy(s->cone.wide, wide.data(), sizeof(int) * wide.size(), hipMemcpyHostToDevice));
    }
    std::vector<int> es((size_t)std::max(1, d.nc), -1);
    for (int j = 0; j < d.n_soc; ++j)
        for (int k = 0; k < s->h_soc_dim[j]; ++k) es[s->h_soc_start[j] + k] = j;
    if (d.nc) CK(hipMemcpy(s->cone.entry_soc, es.data(), sizeof(int) * d.nc, hipMemcpyHostToDevice));
    s->hpoint.assign(N, 0.0);
    s->hparams.assign((size_t)d.np, 0.0);
    Options& o = s->opt;
#define OD(f) s->optd["opt." #f] = &o.f
    OD(residual_norm); OD(constraint_norm); OD(scaling_line_search); OD(iterative_refinement_tolerance); OD(central_path_initial);
    OD(central_path_update_tolerance); OD(central_path_scaling); OD(central_path_exponent); OD(penalty_initial); OD(penalty_scaling);
    OD(dual_initial); OD(residual_tolerance); OD(optimality_tolerance); OD(slack_tolerance); OD(equality_tolerance);
    OD(complementarity_tolerance); OD(min_regularization); OD(primal_regularization_initial); OD(dual_regularization_initial);
    OD(max_regularization); OD(dual_regularization); OD(dual_regularization_exponent); OD(scaling_regularization_initial);
    OD(scaling_regularization); OD(scaling_regularization_last); OD(min_central_path); OD(max_penalty); OD(constraint_tensor);
    OD(update_factorization); OD(violation_tolerance); OD(violation_exponent); OD(merit_tolerance); OD(merit_exponent);
    OD(armijo_tolerance); OD(machine_tolerance); OD(max_filter); OD(differentiate); OD(warmstart);
#undef OD
    const int mf = (int)o.max_filter;
    s->filter_theta.assign(mf, 1.0e8); s->filter_merit.assign(mf, 1.0e8);
    s->cache_theta.assign(mf, 1.0e8); s->cache_merit.assign(mf, 1.0e8);
    if (cp) {
        // the blocks and the multifrontal plan of the Schur complement are part of the handle from the start (there is no dense path to fall back to)
        CK(hipStreamSynchronize(s->stream));
        int brc = calipso::blocks_install(s, plan);
        if (brc < 0) return brc;
        brc = calipso_hip_set_stage_parallel(s, 1, 1, nullptr);
        if (brc < 0) return brc;
    }
    return CALIPSO_OK;
}

int32_t calipso_hip_create(int64_t nx, int64_t np, int64_t ne, int64_t nc, int64_t n_nonneg, const int64_t* nonneg_idx, int64_t n_soc,
                           const int64_t* soc_ptr, const int64_t* soc_idx, int32_t device, H** out) {
    return create_impl(nx, np, ne, nc, n_nonneg, nonneg_idx, n_soc, soc_ptr, soc_idx, device, nullptr, out);
}

// Solver(...) for a stage-structured problem whose sparsity is known up front (the reference's methods.*_sparsity lists, src/trajectory_optimization/
// sparsity.jl:28-129): constraint row k of [equality; cone] touches columns row_first[k]..row_last[k] (1-based, inclusive; first > last: an empty row),
// the Lagrangian Hessian is block diagonal with blocks starting at columns hessian_block_start[0] = 1 < ... .  The handle holds ONLY the blocks (packed,
// both orientations), the tiles of the Schur complement that the blocks couple, and the multifrontal factor: O(stages x block^2) device memory instead of the
// dense nx^2 / m nx / NP^2 buffers.  Uploads: calipso_hip_set_field with the dense host arrays (packed on the host, entries outside the declared structure
// are an error), calipso_hip_set_sparsity + calipso_hip_scatter_* (straight into the blocks), calipso_hip_qp_attach.  Not available on such a handle:
// device evaluators, differentiate!, calipso_hip_analyze_structure / clear_structure (the structure is fixed).
int32_t calipso_hip_create_structured(int64_t nx, int64_t np, int64_t ne, int64_t nc, int64_t n_nonneg, const int64_t* nonneg_idx, int64_t n_soc,
                                      const int64_t* soc_ptr, const int64_t* soc_idx, int32_t device, const int64_t* row_first, const int64_t* row_last,
                                      int64_t n_hessian_blocks, const int64_t* hessian_block_start, H** out) {
    if (((ne + nc) > 0 && (!row_first || !row_last)) || !hessian_block_start || n_hessian_blocks < 1) { g_create_err = "calipso_hip_create_structured: structure arrays missing"; return CALIPSO_ERR_ARGUMENT; }
    const StructureSpec spec{row_first, row_last, n_hessian_blocks, hessian_block_start};
    const int32_t rc = create_impl(nx, np, ne, nc, n_nonneg, nonneg_idx, n_soc, soc_ptr, soc_idx, device, &spec, out);
    if (rc != CALIPSO_OK && out && *out) { g_create_err = (*out)->err; (void)calipso_hip_destroy(*out); *out = nullptr; }
    return rc;
}

int32_t calipso_hip_destroy(H* s) {
    if (!s) return CALIPSO_OK;
    if (s->owner) calipso::group_member_destroyed(s->owner, s);   // a live group must never touch this handle again (group.hip)
    (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    nonsymmetric_release(s);
    ldlsolver_release(s);
    calipso::lfac_release(s);
    scatter_release(s);
    if (s->spS) { (void)calipso_hip_sparse_destroy(s->spS); s->spS = nullptr; }
    if (s->spS_src) { (void)hipFree(s->spS_src); s->spS_src = nullptr; }
    if (s->spS_inv) { (void)hipFree(s->spS_inv); s->spS_inv = nullptr; }
    if (s->d_reach) { (void)hipFree(s->d_reach); s->d_reach = nullptr; }
    calipso::blocks_release(s);
    double* dp[] = {s->slab, s->Kdense, s->multi_rhs, s->dsym_multi, s->evalL, s->evalZ};
    for (double* p : dp) if (p) (void)hipFree(p);
    int* ip[] = {s->cone.soc_start, s->cone.soc_dim, s->cone.soc_woff, s->cone.entry_soc, s->cone.wide, s->zgrp, s->gate};
    for (int* p : ip) if (p) (void)hipFree(p);
    if (s->hscal) (void)hipHostFree(s->hscal);
    if (s->hicount) (void)hipHostFree(s->hicount);
    if (s->hseq) (void)hipHostFree(s->hseq);
    for (auto& e : s->ev) if (e) (void)hipEventDestroy(e);
    calipso::ldl_drop_graphs(s);
    for (auto& e : s->ev_side) if (e) (void)hipEventDestroy(e);
    if (s->hprog) (void)hipHostFree(s->hprog);
    if (s->stream2) (void)hipStreamDestroy(s->stream2);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
    return CALIPSO_OK;
}

}  // extern "C"

// ---- field table --------------------------------------------------------------------------------------------------------
struct Field { double* dev; double* host; int64_t len; int64_t rows = 0, ld = 0; };   // rows/ld != 0: a column-major sub-block of a stacked matrix

static bool find_field(H* s, const std::string& name, Field& f) {
    const Dims& d = s->d;
    f.dev = nullptr; f.host = nullptr; f.len = 0; f.rows = 0; f.ld = 0;
    auto D = [&](double* p, int64_t len) { f.dev = p; f.len = len; return true; };
    auto Hh = [&](double* p) { f.host = p; f.len = 1; return true; };
    if (name == "objective") return D(s->dscal + 0, 1);
    if (name == "barrier") return D(s->dscal + 1, 1);
    if (name == "objective_gradient_variables") return D(s->fx, d.nx);
    if (name == "equality_constraint") return D(s->g, d.ne);
    if (name == "equality_jacobian_variables") { f.rows = d.ne; f.ld = d.m; return D(s->gx, (int64_t)d.ne * d.nx); }
    if (name == "equality_dual_jacobian_variables") return D(s->gyx, d.nx);
    if (name == "cone_constraint") return D(s->hc, d.nc);
    if (name == "cone_jacobian_variables") { f.rows = d.nc; f.ld = d.m; return D(s->hx, (int64_t)d.nc * d.nx); }
    if (name == "cone_dual_jacobian_variables") return D(s->hzx, d.nx);
    if (name == "lagrangian_hessian") return D(s->Lxx, (int64_t)d.nx * d.nx);
    if (name == "cone_product") return D(s->cone_product, d.nc);
    if (name == "cone_target") return D(s->cone_target, d.nc);
    if (name == "barrier_gradient") return D(s->barrier_gradient, d.nc);
    if (name == "lagrangian_gradient_parameters") return D(s->lgp, (int64_t)d.nx * d.np);
    if (name == "equality_jacobian_parameters") return D(s->gp, (int64_t)d.ne * d.np);
    if (name == "cone_jacobian_parameters") return D(s->hp, (int64_t)d.nc * d.np);
    if (name == "jacobian_parameters") return D(s->jacobian_parameters, (int64_t)d.N * d.np);
    if (name == "solution_sensitivity") return D(s->solution_sensitivity, (int64_t)d.N * d.np);
    if (name == "solution") return D(s->solution, d.N);
    if (name == "candidate") return D(s->candidate, d.N);
    if (name == "residual") return D(s->residual, d.N);
    if (name == "residual_error") return D(s->residual_error, d.N);
    if (name == "step") return D(s->step, d.N);
    if (name == "step_correction") return D(s->step_correction, d.N);
    if (name == "residual_symmetric") return D(s->residual_symmetric, d.n);
    if (name == "step_symmetric") return D(s->step_symmetric, d.n);
    if (name == "merit_gradient") return D(s->merit_gradient, d.n);
    if (name == "jacobian_variables_symmetric") return D(s->Kdense, (int64_t)d.n * d.n);
    if (name == "parameters") return D(s->parameters, d.np);
    if (name == "dual") return D(s->lambda, d.ne);
    if (name == "central_path") return Hh(&s->sc.kappa);
    if (name == "fraction_to_boundary") return Hh(&s->sc.tau);
    if (name == "penalty") return Hh(&s->sc.rho);
    if (name == "primal_regularization") return Hh(&s->sc.ep);
    if (name == "primal_regularization_last") return Hh(&s->sc.ep_last);
    if (name == "dual_regularization") return Hh(&s->sc.ed);
    auto it = s->optd.find(name);
    if (it != s->optd.end()) return Hh(it->second);
    // integer options as doubles
    Options& o = s->opt;
    static thread_local double tmp;
    (void)tmp;
    struct IO { const char* n; calipso::i64* p; };
    IO ios[] = {{"opt.max_outer_iterations", &o.max_outer_iterations}, {"opt.max_residual_iterations", &o.max_residual_iterations},
                {"opt.max_residual_line_search", &o.max_residual_line_search}, {"opt.max_cone_line_search", &o.max_cone_line_search},
                {"opt.iterative_refinement", &o.iterative_refinement}, {"opt.max_iterative_refinement", &o.max_iterative_refinement},
                {"opt.min_iterative_refinement", &o.min_iterative_refinement}, {"opt.solve_block", &s->solve_block}, {"opt.solve_wform", &s->solve_wform}};
    for (auto& io : ios)
        if (name == io.n) { f.host = (double*)io.p; f.len = -1; return true; }   // len -1 marks an int64 slot
    return false;
}

extern "C" {

int32_t calipso_hip_set_field(H* s, const char* name, const double* data, int64_t len) {
    if (!s || !name || (!data && len > 0)) return CALIPSO_ERR_ARGUMENT;
    Field f;
    if (!find_field(s, name, f)) return fail_arg(s, std::string("unknown field: ") + name);
    const std::string nm = name;
    if (f.len == -1) {
        if (len != 1) return fail_arg(s, "scalar expected");
        const calipso::i64 v = (calipso::i64)llround(data[0]);
        if (nm == "opt.max_cone_line_search" && (v < 0 || v + 1 > CONE_MASK_TRIALS)) return fail_arg(s, "opt.max_cone_line_search must be in 0..831");
        if (nm == "opt.solve_block") {            // not an option of the reference: a tuning knob of the device factorisation (ldl.hip)
            if (v != 512 && v != 1024 && v != 2048) return fail_arg(s, "opt.solve_block must be 512, 1024 or 2048");
            if (v != s->solve_block) {            // the captured launch sequences and the layout of the inverse blocks change with it
                CK(hipSetDevice(s->device)); CK(hipStreamSynchronize(s->stream));
                calipso::ldl_drop_graphs(s);
                if (!s->compact) { CK(hipMemsetAsync(s->Tinv, 0, sizeof(double) * calipso::tinv_doubles(s->d.NP), s->stream)); CK(hipStreamSynchronize(s->stream)); }
            }
        }
        if (nm == "opt.solve_wform") {            // not an option of the reference either: the stacked [Tinv; W] form of the solves (ldl.hip)
            if (v != 0 && v != 1) return fail_arg(s, "opt.solve_wform must be 0 or 1");
            if (v != s->solve_wform) { CK(hipSetDevice(s->device)); CK(hipStreamSynchronize(s->stream)); calipso::ldl_drop_graphs(s); }
        }
        *(calipso::i64*)f.host = v;
        return CALIPSO_OK;
    }
    if (len != f.len) return fail_arg(s, std::string("wrong length for field ") + name);
    if (f.host) {
        if (nm == "opt.scaling_line_search" && !(data[0] > 0.0 && data[0] < 1.0)) return fail_arg(s, "opt.scaling_line_search must be in (0, 1)");
        if (nm == "opt.max_filter") { if (!(data[0] >= 1.0)) return fail_arg(s, "opt.max_filter must be >= 1"); filter_resize(s, (calipso::i64)data[0]); }   // filter.jl:7-13 sizes it from the option
        *f.host = data[0];
        return CALIPSO_OK;
    }
    if (!f.dev) return fail_arg(s, std::string("field not allocated: ") + name);
    if (len == 0) return CALIPSO_OK;
    CK(hipSetDevice(s->device));
    if (s->compact && (nm == "lagrangian_hessian" || nm == "equality_jacobian_variables" || nm == "cone_jacobian_variables"))      // structured handle: the dense host array goes into the blocks
        return blocks_upload_dense(s, nm == "lagrangian_hessian" ? 0 : (nm == "equality_jacobian_variables" ? 1 : 2), data, 1.0);
    if (f.ld)
        CK(hipMemcpy2DAsync(f.dev, sizeof(double) * f.ld, data, sizeof(double) * f.rows, sizeof(double) * f.rows, len / f.rows, hipMemcpyHostToDevice, s->stream));
    else
        CK(hipMemcpyAsync(f.dev, data, sizeof(double) * len, hipMemcpyHostToDevice, s->stream));
    if (nm == "parameters") s->hparams.assign(data, data + len);
    if (nm == "lagrangian_hessian") s->hessian_dirty = true;
    // an analysed stage-banded structure (structure.hip) is a promise about where these three blocks are non-zero: re-check it
    // on the device against what was just uploaded; a block that breaks it sends the handle back to the dense treatment
    if (structure_active(s) && (nm == "lagrangian_hessian" || nm == "equality_jacobian_variables" || nm == "cone_jacobian_variables")) {
        const int rc = structure_validate(s, nm == "lagrangian_hessian" ? 0 : (nm == "equality_jacobian_variables" ? 1 : 2));
        if (rc < 0) return rc;
        blocks_pack(s, nm != "lagrangian_hessian", nm == "lagrangian_hessian");     // stage blocks (if still on): the packed copies follow the upload
    }
    SYNC();
    return CALIPSO_OK;
}

int32_t calipso_hip_get_field(H* s, const char* name, double* data, int64_t len) {
    if (!s || !name || (!data && len > 0)) return CALIPSO_ERR_ARGUMENT;
    Field f;
    if (!find_field(s, name, f)) return fail_arg(s, std::string("unknown field: ") + name);
    if (f.len == -1) { if (len != 1) return fail_arg(s, "scalar expected"); data[0] = (double)*(calipso::i64*)f.host; return CALIPSO_OK; }
    if (len != f.len) return fail_arg(s, std::string("wrong length for field ") + name);
    if (f.host) { data[0] = *f.host; return CALIPSO_OK; }
    if (!f.dev) return fail_arg(s, std::string("field not computed yet: ") + name);
    if (len == 0) return CALIPSO_OK;
    CK(hipSetDevice(s->device));
    {
        const std::string nm = name;
        if (s->compact && (nm == "lagrangian_hessian" || nm == "equality_jacobian_variables" || nm == "cone_jacobian_variables"))
            return blocks_download_dense(s, nm == "lagrangian_hessian" ? 0 : (nm == "equality_jacobian_variables" ? 1 : 2), data);
    }
    if (f.ld)
        CK(hipMemcpy2DAsync(data, sizeof(double) * f.rows, f.dev, sizeof(double) * f.ld, sizeof(double) * f.rows, len / f.rows, hipMemcpyDeviceToHost, s->stream));
    else
        CK(hipMemcpyAsync(data, f.dev, sizeof(double) * len, hipMemcpyDeviceToHost, s->stream));
    SYNC();
    return CALIPSO_OK;
}

// Indices(nx, np, ne, nc; nonnegative, second_order)  indices.jl:20-63 — 1-based contiguous ranges
int64_t calipso_hip_get_index(H* s, const char* name, int64_t* out, int64_t cap) {
    if (!s || !name) return CALIPSO_ERR_ARGUMENT;
    const Dims& d = s->d;
    const std::string n = name;
    int64_t off = 0, len = -1;
    if (n == "variables") { off = 0; len = d.nx; }
    else if (n == "equality_slack") { off = d.nx; len = d.ne; }
    else if (n == "cone_slack") { off = d.nx + d.ne; len = d.nc; }
    else if (n == "equality_dual") { off = d.nx + d.ne + d.nc; len = d.ne; }
    else if (n == "cone_dual") { off = d.nx + d.ne + d.nc + d.ne; len = d.nc; }
    else if (n == "cone_slack_dual") { off = d.nx + d.ne + d.nc + d.ne + d.nc; len = d.nc; }
    else if (n == "symmetric" || n == "primals") { off = 0; len = d.nx + d.ne + d.nc; }
    else if (n == "symmetric_equality") { off = d.nx; len = d.ne; }
    else if (n == "symmetric_cone") { off = d.nx + d.ne; len = d.nc; }
    else if (n == "duals") { off = d.nx + d.ne + d.nc; len = d.ne + 2 * d.nc; }
    else if (n == "violation_equality") { off = 0; len = d.ne; }
    else if (n == "violation_cone") { off = d.ne; len = d.nc; }
    else if (n == "parameters") { off = 0; len = d.np; }
    if (len >= 0) {
        if (out) for (int64_t i = 0; i < len && i < cap; ++i) out[i] = off + i + 1;
        return len;
    }
    const std::vector<int64_t>* v = nullptr;
    if (n == "cone_nonnegative") v = &s->h_nonneg;
    else if (n == "cone_second_order") v = &s->h_soc_idx;
    else if (n == "cone_second_order_ptr") v = &s->h_soc_ptr;
    if (!v) return fail_arg(s, std::string("unknown index set: ") + name);
    if (out) for (size_t i = 0; i < v->size() && (int64_t)i < cap; ++i) out[i] = (*v)[i];
    return (int64_t)v->size();
}

int32_t calipso_hip_synchronize(H* s) { if (!s) return CALIPSO_ERR_ARGUMENT; SYNC(); CK(hipGetLastError()); return CALIPSO_OK; }

}  // extern "C"

// ---- do the streams of two handles really run side by side? --------------------------------------------------------------------
// Independent Solvers (solver.jl:46-150) stepped from different host threads share the GPU only as far as the runtime lets their streams: HIP streams are
// multiplexed onto a few hardware queues, and two queues that end up behind one dispatcher serialise — a short kernel of one handle then waits until a long
// kernel of the other has been dispatched COMPLETELY (all its workgroups placed), which is most of its duration.  Which streams collide depends on the order
// in which the process created them and cannot be queried: BASELINE config 4's dense batch ran at 1290 or 1480 steps/s per GPU depending on how many handles
// the process had created before (profiles/r06_ab_closing.txt).  So it is MEASURED: a kernel of far more workgroups than the chip holds on stream a, a
// one-workgroup kernel on stream b right behind it; side by side, the small one is through in about the time of one workgroup of the large one.
__global__ void k_probe_spin(long long ticks) {                       // (100 MHz wall clock)
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}
struct ProbeEvents {                       // four events, destroyed whatever way the probe leaves
    hipEvent_t e[4] = {nullptr, nullptr, nullptr, nullptr};
    bool ok = true;
    ProbeEvents() { for (auto& x : e) ok = ok && hipEventCreate(&x) == hipSuccess; }
    ~ProbeEvents() { for (auto& x : e) if (x) (void)hipEventDestroy(x); }
};
static int32_t probe_one_way(H* a, H* b, double* small_us, double* large_us) {
    ProbeEvents pe;
    if (!pe.ok) return CALIPSO_ERR_HIP;
    hipEvent_t ea0 = pe.e[0], ea1 = pe.e[1], eb0 = pe.e[2], eb1 = pe.e[3];
    int cus = 256;
    { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, a->device) == hipSuccess && pr.multiProcessorCount > 0) cus = pr.multiProcessorCount; }
    (void)hipStreamSynchronize(a->stream); (void)hipStreamSynchronize(b->stream);
    (void)hipEventRecord(ea0, a->stream);
    hipLaunchKernelGGL(k_probe_spin, dim3(cus * 8 * 8), dim3(256), 0, a->stream, (long long)2000);      // eight rounds of 20 us: the dispatch lasts ~140 us
    (void)hipEventRecord(ea1, a->stream);
    (void)hipEventRecord(eb0, b->stream);
    hipLaunchKernelGGL(k_probe_spin, dim3(1), dim3(64), 0, b->stream, (long long)100);
    (void)hipEventRecord(eb1, b->stream);
    const bool ok = hipStreamSynchronize(a->stream) == hipSuccess && hipStreamSynchronize(b->stream) == hipSuccess;
    float ta = 0.f, tb = 0.f;
    const bool ok2 = ok && hipEventElapsedTime(&ta, ea0, ea1) == hipSuccess && hipEventElapsedTime(&tb, eb0, eb1) == hipSuccess;
    if (!ok2) return CALIPSO_ERR_HIP;
    *small_us = 1e3 * (double)tb; *large_us = 1e3 * (double)ta;
    return CALIPSO_OK;
}

// second test: CHAINS of short dependent kernels (what a latency-bound group step is: ~90 launches of a few microseconds) on both streams at once against one stream
// alone.  Two streams of the HIGHEST priority class pass the first test and still run their chains one after the other (C4T, 32 instances in two lanes: 17.7 k
// steps/s = one group alone, against 27.5 k in any other pair of classes).
static int32_t probe_chains(H* a, H* b, double* alone_us, double* both_us) {
    constexpr int N = 48;
    ProbeEvents pe;
    if (!pe.ok) return CALIPSO_ERR_HIP;
    hipEvent_t (&e)[4] = pe.e;
    (void)hipStreamSynchronize(a->stream); (void)hipStreamSynchronize(b->stream);
    (void)hipEventRecord(e[0], a->stream);
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_probe_spin, dim3(1), dim3(64), 0, a->stream, (long long)800);
    (void)hipEventRecord(e[1], a->stream);
    bool ok = hipStreamSynchronize(a->stream) == hipSuccess;
    float t1 = 0.f, ta = 0.f, tb = 0.f;
    ok = ok && hipEventElapsedTime(&t1, e[0], e[1]) == hipSuccess;
    (void)hipEventRecord(e[0], a->stream); (void)hipEventRecord(e[2], b->stream);
    for (int i = 0; i < N; ++i) {
        hipLaunchKernelGGL(k_probe_spin, dim3(1), dim3(64), 0, a->stream, (long long)800);
        hipLaunchKernelGGL(k_probe_spin, dim3(1), dim3(64), 0, b->stream, (long long)800);
    }
    (void)hipEventRecord(e[1], a->stream); (void)hipEventRecord(e[3], b->stream);
    ok = ok && hipStreamSynchronize(a->stream) == hipSuccess && hipStreamSynchronize(b->stream) == hipSuccess;
    ok = ok && hipEventElapsedTime(&ta, e[0], e[1]) == hipSuccess && hipEventElapsedTime(&tb, e[2], e[3]) == hipSuccess;
    if (!ok) return CALIPSO_ERR_HIP;
    *alone_us = 1e3 * (double)t1; *both_us = 1e3 * (double)std::max(ta, tb);
    return CALIPSO_OK;
}

extern "C" {

// out[0] = 1 if the two streams run side by side by BOTH tests, else 0.  Test 1: a short kernel on either stream gets through while a long kernel of the other is being
// dispatched (both directions, best of two tries each): out[1], out[2] = the short kernel's time behind a's / b's long kernel (us), out[3] = the long kernel's duration
// (us).  Test 2: chains of 48 short kernels on both streams at once take about as long as one chain alone: out[4] = one chain alone (us), out[5] = both at once (us).
int32_t calipso_hip_streams_concurrent(H* a, H* b, double out[6]) {
    if (!a || !b || !out || a == b || a->device != b->device) return CALIPSO_ERR_ARGUMENT;
    { H* s = a; CK(hipSetDevice(a->device)); }
    double worst[2] = {0.0, 0.0}, large = 0.0;
    for (int dir = 0; dir < 2; ++dir) {
        double best = 1e30;
        for (int rep = 0; rep < 2; ++rep) {
            double sm = 0.0, lg = 0.0;
            const int32_t rc = dir == 0 ? probe_one_way(a, b, &sm, &lg) : probe_one_way(b, a, &sm, &lg);
            if (rc != CALIPSO_OK) return rc;
            best = std::min(best, sm); large = std::max(large, lg);
        }
        worst[dir] = best;
    }
    out[1] = worst[0]; out[2] = worst[1]; out[3] = large;
    // side by side: the short kernel is through after about ONE round of the long kernel's workgroups (20 us of its ~170: measured 10 - 22 us both ways); the colliding
    // pairs measured 54 - 108 us one way (and a long kernel of 231 us instead of 170): the bar is a fifth of the long kernel
    const bool test1 = std::max(worst[0], worst[1]) < 0.2 * large;
    double alone = 0.0, both = 1e30;
    for (int rep = 0; rep < 2; ++rep) {          // (best of two: a hiccup of the host between the launches must not read as a collision)
        double t1 = 0.0, t2 = 0.0;
        const int32_t rc = probe_chains(a, b, &t1, &t2);
        if (rc != CALIPSO_OK) return rc;
        if (t2 / std::max(t1, 1e-9) < both / std::max(alone, 1e-9)) { alone = t1; both = t2; }
    }
    out[4] = alone; out[5] = both;
    out[0] = (test1 && both < 1.4 * alone) ? 1.0 : 0.0;
    return CALIPSO_OK;
}

// A NEW stream for the handle (the old one is drained and destroyed): the runtime binds a new stream to the least used hardware queue of its priority class, so a
// handle whose stream collides with another's (calipso_hip_streams_concurrent) gets another queue.  priority_class: 0, 1, 2 (the three classes calipso_hip_create
// deals out by creation order), -1: keep the class.  Nothing else of the handle changes (events, captured launch graphs and the second stream do not depend on it).
int32_t calipso_hip_rebind_stream(H* s, int32_t priority_class) {
    if (!s || priority_class < -1 || priority_class > 2) return CALIPSO_ERR_ARGUMENT;
    if (s->cur) return fail_arg(s, "a member of a live group launch cannot change its stream");
    CK(hipSetDevice(s->device));
    SYNC();
    if (s->stream2) CK(hipStreamSynchronize(s->stream2));
    int least = 0, greatest = 0, prio = 0;
    CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    if (priority_class < 0) CK(hipStreamGetPriority(s->stream, &prio));
    else { const int span = least - greatest; prio = span > 0 ? greatest + (priority_class % (span + 1)) : least; }
    hipStream_t fresh = nullptr;
    CK(hipStreamCreateWithPriority(&fresh, hipStreamNonBlocking, prio));      // (created BEFORE the old one goes: the runtime must not hand the same queue back)
    hipStream_t old = s->stream;
    s->stream = fresh;
    (void)hipStreamDestroy(old);
    return CALIPSO_OK;
}

}  // extern "C"

// ---- internal helpers -------------------------------------------------------------------------------------------------------
// Scalar read-backs (refinement norms, merit / step-length decisions, cone-search masks: ~14 per Newton step).  A hipMemcpyAsync to pinned
// memory + hipStreamSynchronize costs 11.4 us between two dependent kernels on this system; a one-workgroup kernel that stores the words
// straight into mapped pinned host memory, then (system-scope release) a sequence number the host spins on, costs 6.4 us
// (bench/readback_latency.hip, profiles/r02_readback_latency.txt).  Everything queued before the publish kernel has completed when the
// sequence number arrives (in-order stream).
__global__ __launch_bounds__(128) void k_publish_words(const unsigned* __restrict__ src, int words, unsigned* __restrict__ hdst, unsigned long long* __restrict__ hseq,
                                                        unsigned long long seq) {
    if ((int)threadIdx.x < words) hdst[threadIdx.x] = src[threadIdx.x];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(hseq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
static int wait_published(H* s, unsigned long long seq);
static int publish_and_wait(H* s, const void* dsrc, void* hdst_dev, int words) {
    const unsigned long long seq = ++s->pub_seq;
    hipLaunchKernelGGL(k_publish_words, dim3(1), dim3(128), 0, s->stream, static_cast<const unsigned*>(dsrc), words, static_cast<unsigned*>(hdst_dev), s->hseq_dev, seq);
    return wait_published(s, seq);
}
// spin until the sequence number `seq` (written by a publish kernel or by a producer kernel itself) has arrived
static int wait_published(H* s, unsigned long long seq) {
    if (launch_errors(s, "a kernel launch of this phase was refused")) return CALIPSO_ERR_HIP;
    hipError_t q = hipSuccess;
    // (>=: the sequence numbers of a handle only grow and only the publishes of its own in-order stream store them — a wait for an EARLIER number than the last one
    // published is satisfied, not a hang; what was published under the earlier number is still in its mapped words)
    const bool ok = host_wait([&] { return __atomic_load_n(s->hseq, __ATOMIC_ACQUIRE) >= seq; },
                              [&] { q = hipStreamQuery(s->stream); return q == hipErrorNotReady; });      // a faulted queue would never publish: look at the stream now and then
    if (ok) return 0;
    if (q != hipSuccess && q != hipErrorNotReady) return calipso::check(s, q, "publish_and_wait");
    s->err = "scalar read-back did not arrive";
    return CALIPSO_ERR_HIP;
}
static int read_scalars(H* s, int first, int count) { return publish_and_wait(s, s->dscal + first, s->hscal_dev + first, 2 * count); }
static int read_icount(H* s, int first, int count) { return publish_and_wait(s, s->icount + first, s->hicount_dev + first, count); }
static double* point_of(H* s, int which) { return which == 0 ? s->solution : s->candidate; }

// phase times of the last factorisation from its events (do_factorize does not wait for them when the chain published the inertia counts)
static void factor_times(H* s) {
    if (!s->factor_times_pending) return;
    s->factor_times_pending = false;
    if (hipEventSynchronize(s->ev[13]) != hipSuccess) return;
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, s->ev[10], s->ev[11]); s->phase_ms[1] = ms;   // cone pivots + Omega*hx
    (void)hipEventElapsedTime(&ms, s->ev[11], s->ev[12]); s->phase_ms[7] = ms;   // Schur complement (MFMA kernel), one launch
    (void)hipEventElapsedTime(&ms, s->ev[12], s->ev[13]); s->phase_ms[3] = ms;   // LDL^T of S
    s->kernel_ms[0] = hipEventElapsedTime(&ms, s->ev[12], s->ev[14]) == hipSuccess ? ms : 0.0;   // of which the panel steps (the pivot chain)
}

static int do_factorize(H* s, int64_t inertia[3], bool rhs_ahead_ok = false) {
    factor_times(s);                      // (before the events are recorded again)
    (void)hipEventRecord(s->ev[10], s->stream);
    launch_cone_weights(s);
    // the operands of the first condensed solve (search_direction_symmetric!: residual_symmetric!, then b_x += [gx; hx]'(Omega b_m)) need these pivots and nothing of the
    // factor: on the handle's second stream they run beside k_schur (ldl.hip: ldl_rhs_stream; joined with the finish of the factorisation) instead of behind it
    s->rhs_ahead = false; s->rhs_joined = false;
    if (rhs_ahead_ok) {
        if (hipStream_t st2 = calipso::ldl_rhs_stream(s)) {
            (void)hipEventRecord(s->ev_side[6], s->stream);
            (void)hipStreamWaitEvent(st2, s->ev_side[6], 0);
            hipStream_t keep = s->stream;
            s->stream = st2;
            launch_residual_symmetric(s, s->residual);
            if (s->d.m) calipso::gemv_t(s, s->d.m, s->d.nx, s->Z, s->d.m, s->t1, s->xbuf, 1.0, 1.0, calipso::SP_Z);
            s->stream = keep;
            s->rhs_ahead = true;
        }
    }
    launch_scale_rows(s);
    calipso::inject_refused_launch(s->stream);
    (void)hipEventRecord(s->ev[11], s->stream);
    launch_schur(s);
    (void)hipEventRecord(s->ev[12], s->stream);
    launch_ldl(s);
    (void)hipEventRecord(s->ev[13], s->stream);
    s->factor_times_pending = true;
    if (s->ldl_failed) { s->ldl_failed = false; SYNC(); return CALIPSO_ERR_HIP; }       // (s->err says why: launch_ldl / a dense mat-vec on a structured handle)
    if (s->ldl_pub_seq) {
        // the last diagonal block published the counts when the pivot chain ended: the host goes on queueing behind the finish of the last solve block
        if (wait_published(s, s->ldl_pub_seq)) return CALIPSO_ERR_HIP;
    } else {
        if (launch_errors(s, "a kernel launch of the factorisation was refused")) return CALIPSO_ERR_HIP;
        CK(hipMemcpyAsync(s->hicount, s->icount, sizeof(int) * 6, hipMemcpyDeviceToHost, s->stream));
        SYNC();
        factor_times(s);
    }
    s->stats.factorizations += 1;
    s->phase_ms[8] += 1.0;
    const int64_t pos = s->hicount[0] + s->hicount[3], nonpos = s->hicount[1] + s->hicount[4], zero = s->hicount[2] + s->hicount[5];
    inertia[0] = pos; inertia[1] = nonpos; inertia[2] = zero;
    if (zero > 0) { inertia[0] = -1; return CALIPSO_WARN_ZERO_PIVOT; }   // qdldl.jl:456,579: posDCount = -1
    return CALIPSO_OK;
}

// inertia.jl:30-80.  Quirk kept: the `primal_regularization_last == 0.0` test of :48 compares a Vector with a Float64 and is
// always false, so IC-3 always takes max(min_regularization, scaling_regularization_last * eps_last).
// first_in / first_rc: IC-1 (the factorisation with the initial regularisation) was queued — and its inertia read — by the caller already (inner_iteration: ahead of the
// host's exit tests); the loop goes on from its result
static int do_inertia_correction(H* s, int64_t* nfact, bool rhs_ahead_ok = false, const int64_t* first_in = nullptr, int first_rc = 0) {
    Options& o = s->opt; Scalars& sc = s->sc;
    int64_t in[3];
    int64_t count = 0;
    sc.ep = o.primal_regularization_initial;
    sc.ed = o.dual_regularization_initial;
    int rc;
    if (first_in) { in[0] = first_in[0]; in[1] = first_in[1]; in[2] = first_in[2]; rc = first_rc; }
    else rc = do_factorize(s, in, rhs_ahead_ok);
    count++;                                                     // IC-1
    if (rc < 0) return rc;
    if (inertia_ok(s, in)) { if (nfact) *nfact = count; return CALIPSO_OK; }
    if (in[2] != 0) sc.ed = o.dual_regularization * std::pow(sc.kappa, o.dual_regularization_exponent);   // IC-2
    sc.ep = std::max(o.min_regularization, o.scaling_regularization_last * sc.ep_last);                   // IC-3
    while (!inertia_ok(s, in)) {
        rc = do_factorize(s, in, rhs_ahead_ok); count++;         // IC-4 (the operands are formed again with the new regularisation)
        if (rc < 0) return rc;
        if (inertia_ok(s, in)) break;
        if (sc.ep_last == 0.0) sc.ep = o.scaling_regularization_initial * sc.ep;   // IC-5
        else sc.ep = o.scaling_regularization * sc.ep;
        if (sc.ep > o.max_regularization) { if (nfact) *nfact = count; s->err = "inertia correction failure"; return CALIPSO_ERR_INERTIA; }   // IC-6
    }
    sc.ep_last = sc.ep;
    if (nfact) *nfact = count;
    return CALIPSO_OK;
}

// refine_follows: the caller goes straight on to a refinement residual of s->step (which = 0 only: the local rows of that residual then come out of the same launch)
static void do_sds(H* s, int which, double* accumulate = nullptr, bool refine_follows = false, bool use_ahead = false) {      // use_ahead: the caller is do_search_direction, right behind the factorisation that queued the operands ahead
    const double* res = which == 0 ? s->residual : s->residual_error;
    double* st = which == 0 ? s->step : s->step_correction;
    if (which == 0 && s->rhs_ahead && !s->rhs_joined && s->stream2 && s->ev_side[7]) {      // (also when the operands are not used: nothing of the second stream stays in flight)      // (a finish that did not join the second stream: join it here)
        (void)hipEventRecord(s->ev_side[7], s->stream2);
        (void)hipStreamWaitEvent(s->stream, s->ev_side[7], 0);
        s->rhs_joined = true;
    }
    const bool ready = use_ahead && which == 0 && s->rhs_ahead && s->rhs_joined;      // (do_factorize queued them on the second stream and the factorisation's finish has joined it)
    s->rhs_ahead = false;
    if (!ready) launch_residual_symmetric(s, res);     // b, and the first operands of the condensed solve (xbuf, t1)
    // one launch for t2 = [gx; hx] dx, the back-substitution and the recovery (vectors.hip: k_solve_tail) where the handle allows it (which = 0, no accumulation: its zsx rule)
    const bool tail = which == 0 && !accumulate && s->d.m > 0;
    linear_solve_device(s, !tail, ready);  // dx = S^-1(...) in xbuf (, t2 = [gx; hx] dx)
    if (tail && launch_solve_tail(s, 0, false, refine_follows)) return;
    if (tail) gemv_n(s, s->d.m, s->d.nx, s->Z, s->d.m, s->xbuf, s->t2, 1.0, 0.0, SP_Z);
    // dy, dz back-substitution + dr, ds, dt recovery (+ step += correction); which = 0 also leaves zsx = [gx; hx] step_x = t2 for the refinement
    launch_recover(s, st, res, accumulate, which == 0 ? 1 : 0);
}

// residual_error = residual - H step and its inf-norm (dscal[7]), computed so that the operands of the NEXT condensed solve fall out of the same
// passes (vectors.hip: k_refine_local / k_refine_x): needs zsx = [gx; hx] step_x
static void refine_residual(H* s, bool publish = false) {
    const Dims& d = s->d;
    launch_refine_local(s);
    // [gx; hx]'(two vectors) and Lxx step_x: one launch (gemv.hip); the partial sums of the second are combined by the kernel that consumes them
    const int nchunk = gemv_refine_pair(s, s->step + d.oy(), s->t1, s->w1, s->w2, s->step, s->lxv, true);
    if (nchunk > 0) launch_refine_x_fused(s, publish, nchunk); else launch_refine_x(s, publish);
}
// the condensed solve for the operands refine_residual left (xbuf, residual_symmetric); step += correction, zsx += [gx; hx] dx
static void refine_solve(H* s) {
    const Dims& d = s->d;
    launch_trsv(s, s->xbuf);
    if (d.m && launch_solve_tail(s, 1, true, true)) return;       // t2, recovery, step += correction, zsx += t2 and the local rows of the next residual: one launch
    if (d.m) gemv_n(s, d.m, d.nx, s->Z, d.m, s->xbuf, s->t2, 1.0, 0.0, SP_Z);
    launch_recover(s, s->step_correction, s->residual_error, s->step, 2);
}

// iterative_refinement.jl:1-52.  zsx_valid: zsx already holds [gx; hx] step_x (it does right after do_sds(s, 0))
// Speculative rounds: a refinement of this handle usually takes as many rounds as its last one did, so the initial residual and that many rounds are queued WITHOUT
// waiting for the norms in between — every kernel of a round carries the handle's gate and leaves at once when an earlier residual of this refinement has met the
// stopping test of iterative_refinement.jl:14-16 (the test itself runs in the residual kernel, with the reference's operands) — and the host reads ONE report at the
// end: the first norm, the rounds taken, the last norm, whether the test was met.  Not met yet: the loop goes on as before, one wait per round.  Same kernels in the
// same order on the same data as the round-by-round loop: same bits, same round counts.
static bool spec_refinement_ok(H* s) {
    static const bool env = [] { const char* e = getenv("CALIPSO_HIP_SPEC_REFINE"); return !e || atoi(e) != 0; }();
    return env && !s->cur && s->gate && s->d.m > 0 && !s->compact && !s->blocks.on && !(s->stage_parallel && s->spS) && wform_on(s) && solve_tail_available(s) &&
           s->opt.max_iterative_refinement >= 1;
}
// the round-by-round part of iterative_refinement! from a known state (norm of the current residual_error, the initial norm, rounds done); ran_more: a round was run here
static int refinement_loop(H* s, double norm, double norm0, int it, int* rounds, double* final_norm, bool* ran_more);
static int do_refinement(H* s, int* rounds, double* final_norm, bool zsx_valid = false) {
    const Options& o = s->opt; const Dims& d = s->d;
    // (fill!(step_correction, 0) of iterative_refinement.jl:5 is only launched when no round follows: the first round's k_recover writes every entry)
    if (!zsx_valid && d.m) gemv_n(s, d.m, d.nx, s->Z, d.m, s->step, s->zsx, 1.0, 0.0, SP_Z);
    double norm, norm0;
    int it = 0;
    bool met = false;
    if (spec_refinement_ok(s)) {
        const int spec = (int)std::min<calipso::i64>(o.max_iterative_refinement, std::max<calipso::i64>(1, s->stats.last_refine > 0 ? s->stats.last_refine : o.min_iterative_refinement));
        if (++s->gate_counter == 0) s->gate_counter = 1;
        s->gate_epoch = s->gate_counter;
        for (int k = 0; k <= spec; ++k) {
            if (k > 0) {
                launch_trsv_direct(s, s->xbuf);
                (void)launch_solve_tail(s, 1, true, true);
            }
            launch_refine_local(s);                 // (a no-op behind a solve tail; the initial residual of a handle whose solve had no tail runs it — ungated, k = 0)
            const int nchunk = gemv_refine_pair(s, s->step + d.oy(), s->t1, s->w1, s->w2, s->step, s->lxv, true);
            launch_refine_x_fused(s, true, nchunk, k, k == spec);
        }
        s->gate_epoch = 0;
        // refine_defer (inner_iteration): the caller queues what follows the search direction behind these rounds and reads the report with ITS read-back
        // (do_refinement_resume) — no host wait here
        if (s->refine_defer) { s->refine_pending = true; if (rounds) *rounds = -1; return CALIPSO_OK; }
        if (wait_published(s, s->pub_seq)) return CALIPSO_ERR_HIP;
        norm = s->hscal[7]; norm0 = s->hscal[20]; it = (int)s->hscal[21]; met = s->hscal[22] != 0.0;
    } else {
        refine_residual(s, true);                  // (k_refine_x itself publishes the norm: no separate read-back launch)
        if (wait_published(s, s->pub_seq)) return CALIPSO_ERR_HIP;
        norm = s->hscal[7];
        norm0 = norm;
    }
    (void)met;
    return refinement_loop(s, norm, norm0, it, rounds, final_norm, nullptr);
}
// the report of the speculative rounds has arrived with a later read-back of the caller: go on from it
static int do_refinement_resume(H* s, int* rounds, bool* ran_more) {
    s->refine_pending = false;
    return refinement_loop(s, s->hscal[7], s->hscal[20], (int)s->hscal[21], rounds, nullptr, ran_more);
}
static int refinement_loop(H* s, double norm, double norm0, int it, int* rounds, double* final_norm, bool* ran_more) {
    const Options& o = s->opt;
    if (ran_more) *ran_more = false;
    while (it <= o.max_iterative_refinement) {
        if (norm <= o.iterative_refinement_tolerance && it >= o.min_iterative_refinement) {
            if (it == 0) fill_d(s, s->step_correction, s->d.N, 0.0);
            if (rounds) *rounds = it;
            if (final_norm) *final_norm = norm;
            s->stats.last_refine = it; s->stats.refine_max = std::max<calipso::i64>(s->stats.refine_max, it);
            return CALIPSO_OK;
        }
        // a residual with a NaN in it reports +inf (vectors.hip: rabs).  The reference's norm is NaN there: `norm <= tol` is never true, so its loop (`while iteration <=
        // max_iterative_refinement`, iterative_refinement.jl:14-44) runs ALL its rounds on NaNs before it fails (:45-51).  DEVIATION, same outcome: the rounds that cannot
        // change the verdict are not run — the loop leaves as soon as the minimum number of rounds is done, fails (WARN_REFINEMENT -> the H \ residual fallback) and
        // reports the reference's round count (max_iterative_refinement + 1) in rounds / stats so that the statistics agree with the reference's
        if (!std::isfinite(norm) && it >= o.min_iterative_refinement) { it = (int)std::max<calipso::i64>(it, o.max_iterative_refinement + 1); break; }
        if (ran_more) *ran_more = true;
        refine_solve(s);                   // step += step_correction fused into the recovery kernel
        refine_residual(s, true);
        if (wait_published(s, s->pub_seq)) return CALIPSO_ERR_HIP;
        norm = s->hscal[7];
        it += 1;
    }
    if (it == 0) fill_d(s, s->step_correction, s->d.N, 0.0);      // (max_iterative_refinement < 0)
    if (rounds) *rounds = it;
    if (final_norm) *final_norm = norm;
    s->stats.last_refine = it; s->stats.refine_max = std::max<calipso::i64>(s->stats.refine_max, it);
    if (std::isfinite(norm) && norm <= norm0) return CALIPSO_OK;
    s->stats.refine_fail += 1;
    return CALIPSO_WARN_REFINEMENT;
}

// defer: the refinement's speculative rounds are queued and NOT waited for (s->refine_pending; the caller reads their report later and calls search_direction_finish)
static int do_search_direction(H* s, int64_t* nfact, int* rounds, const int64_t* first_in = nullptr, int first_rc = 0, bool defer = false) {
    int rc = do_inertia_correction(s, nfact, true, first_in, first_rc);
    if (rc < 0) {
        // operands queued ahead on the second stream belong to a factorisation that failed: wait for them, forget them (a later solve forms its own)
        if (s->rhs_ahead && s->stream2) (void)hipStreamSynchronize(s->stream2);
        s->rhs_ahead = false; s->rhs_joined = false;
        return rc;
    }
    do_sds(s, 0, nullptr, s->opt.iterative_refinement != 0, true);
    if (s->opt.iterative_refinement) {
        s->refine_defer = defer; s->refine_pending = false;
        rc = do_refinement(s, rounds, nullptr, true);
        s->refine_defer = false;
        if (rc < 0) return rc;
        if (rc == CALIPSO_WARN_REFINEMENT) {
            // the reference falls back to `H \ residual` on the unreduced system (search_direction.jl:22,113): fallback.hip
            const int fr = nonsymmetric_solve(s, s->residual, s->step);
            if (fr < 0) return fr;
            return CALIPSO_WARN_REFINEMENT;
        }
    }
    return CALIPSO_OK;
}
// the rest of do_search_direction once the deferred refinement report has arrived; step_changed: the step is not the one the caller queued work on (further rounds ran,
// or the fallback replaced it)
static int search_direction_finish(H* s, int* rounds, bool* step_changed) {
    bool more = false;
    const int rc = do_refinement_resume(s, rounds, &more);
    *step_changed = more;
    if (rc < 0) return rc;
    if (rc == CALIPSO_WARN_REFINEMENT) {
        const int fr = nonsymmetric_solve(s, s->residual, s->step);
        if (fr < 0) return fr;
        *step_changed = true;
        return CALIPSO_WARN_REFINEMENT;
    }
    return CALIPSO_OK;
}
// Queueing ahead of the host's knowledge inside a Newton step (a single handle with a device-side evaluator; CALIPSO_HIP_SPEC_STEP=0: every decision waited for in
// place, as up to round 5): IC-1 of the search direction is queued before the host has seen the norms of the exit tests, and the cone search, the first candidate and its
// merit are queued behind the refinement before its report is read.  The host takes every decision the reference takes, from the same numbers — it only takes
// them later; work queued on a prediction that fails (an exit, a refinement that needs more rounds) is repeated the plain way.
static bool spec_step_ok(const H* s) {
    static const bool env = [] { const char* e = getenv("CALIPSO_HIP_SPEC_STEP"); return !e || atoi(e) != 0; }();
    return env && !s->cur && (s->qp.attached || s->dev_eval || s->dev_block_eval);
}

static int do_cone_search(H* s, double* a_s, double* a_t, bool emit_candidate = true) {
    const Options& o = s->opt;
    if (s->d.nc == 0) { *a_s = 1.0; *a_t = 1.0; return CALIPSO_OK; }
    {   // the kernel publishes its masks itself (no k_publish_words launch behind it)
        const unsigned long long seq = ++s->pub_seq;
        launch_cone_search(s, seq);
        if (wait_published(s, seq)) return CALIPSO_ERR_HIP;
    }
    const int ks = first_feasible_trial(s->hicount + 6, o.max_cone_line_search), kt = first_feasible_trial(s->hicount + 32, o.max_cone_line_search);
    if (ks < 0 || kt < 0) { s->err = "cone search failure"; return CALIPSO_ERR_CONE_SEARCH; }   // solve.jl:210,220
    // step sizes as the reference forms them: repeated multiplication by scaling_line_search (the kernel tested exactly these)
    double as = 1.0, at = 1.0;
    for (int k = 0; k < ks; ++k) as = o.scaling_line_search * as;
    for (int k = 0; k < kt; ++k) at = o.scaling_line_search * at;
    *a_s = as; *a_t = at;
    if (emit_candidate) launch_cone_candidate(s, as, at);
    return CALIPSO_OK;
}

// user evaluation on the device: hand the evaluator the device addresses of the point and of the ProblemData fields; it enqueues its
// kernels on the handle's stream (the launches that follow are ordered behind them) — no download of the point, no upload of blocks
static int device_evaluate(H* s, const double* pt, uint32_t flags) {
    const Dims& d = s->d;
    const uint32_t hess_flags = CALIPSO_EVAL_OBJECTIVE_HESSIAN | CALIPSO_EVAL_EQUALITY_DUAL_HESSIAN | CALIPSO_EVAL_CONE_DUAL_HESSIAN;
    if (s->compact && s->dev_block_eval) {
        // the evaluator writes the packed blocks themselves: no dense scratch, nothing to check (nothing outside the blocks exists), only the second orientation
        // of the blocks it was asked for to refresh behind it
        if (int rc = calipso::blocks_descriptors(s)) return rc;
        calipso_device_block_data o;
        o.objective = s->dscal + 0; o.objective_gradient_variables = s->fx;
        o.equality_constraint = d.ne ? s->g : nullptr; o.cone_constraint = d.nc ? s->hc : nullptr;
        o.equality_dual_jacobian_variables = s->gyx; o.cone_dual_jacobian_variables = s->hzx;
        o.lagrangian_gradient_parameters = d.np ? s->lgp : nullptr;
        o.equality_jacobian_parameters = (d.np && d.ne) ? s->gp : nullptr;
        o.cone_jacobian_parameters = (d.np && d.nc) ? s->hp : nullptr;
        o.nx = d.nx; o.np = d.np; o.ne = d.ne; o.nc = d.nc;
        const calipso::StageBlocks& B = s->blocks;
        o.n_jacobian_blocks = (int64_t)B.h_jdesc.size(); o.jacobian_blocks = B.h_jdesc.data(); o.jacobian_blocks_device = B.d_jdesc;
        o.n_hessian_blocks = (int64_t)B.h_hdesc.size(); o.hessian_blocks = B.h_hdesc.data(); o.hessian_blocks_device = B.d_hdesc;
        const int rc = s->dev_block_eval(s->dev_block_eval_user, flags, pt, pt + d.oy(), pt + d.oz(), s->parameters, &o, (void*)s->stream);
        if (rc != 0) { s->err = "device evaluator failed"; return CALIPSO_ERR_CALLBACK; }
        if (flags & hess_flags) s->hessian_dirty = true;
        calipso::blocks_mirror(s, (flags & hess_flags) != 0, (flags & CALIPSO_EVAL_EQUALITY_JACOBIAN) != 0 && d.ne > 0, (flags & CALIPSO_EVAL_CONE_JACOBIAN) != 0 && d.nc > 0);
        return CALIPSO_OK;
    }
    calipso_device_problem_data o;
    o.objective = s->dscal + 0;
    o.objective_gradient_variables = s->fx;
    o.equality_constraint = d.ne ? s->g : nullptr;
    o.cone_constraint = d.nc ? s->hc : nullptr;
    o.equality_dual_jacobian_variables = s->gyx;
    o.cone_dual_jacobian_variables = s->hzx;
    o.lagrangian_hessian = s->Lxx;
    o.equality_jacobian_variables = d.ne ? s->gx : nullptr;
    o.cone_jacobian_variables = d.nc ? s->hx : nullptr;
    o.jacobian_ld = d.m;
    if (s->compact) {
        if (!s->evalL) {
            if (dalloc(s, &s->evalL, (size_t)d.nx * d.nx) || dalloc(s, &s->evalZ, (size_t)std::max(1, d.m) * d.nx)) return CALIPSO_ERR_HIP;
            s->scratch_bytes = sizeof(double) * ((size_t)d.nx * d.nx + (size_t)std::max(1, d.m) * d.nx);
        }
        o.lagrangian_hessian = s->evalL;
        o.equality_jacobian_variables = d.ne ? s->evalZ : nullptr;
        o.cone_jacobian_variables = d.nc ? s->evalZ + d.ne : nullptr;
    }
    o.lagrangian_gradient_parameters = d.np ? s->lgp : nullptr;
    o.equality_jacobian_parameters = (d.np && d.ne) ? s->gp : nullptr;
    o.cone_jacobian_parameters = (d.np && d.nc) ? s->hp : nullptr;
    o.nx = d.nx; o.np = d.np; o.ne = d.ne; o.nc = d.nc;
    const int rc = s->dev_eval(s->dev_eval_user, flags, pt, pt + d.oy(), pt + d.oz(), s->parameters, &o, (void*)s->stream);
    if (rc != 0) { s->err = "device evaluator failed"; return CALIPSO_ERR_CALLBACK; }
    const uint32_t hess = CALIPSO_EVAL_OBJECTIVE_HESSIAN | CALIPSO_EVAL_EQUALITY_DUAL_HESSIAN | CALIPSO_EVAL_CONE_DUAL_HESSIAN;
    if (flags & hess) s->hessian_dirty = true;
    if (s->compact) {
        const bool jg = (flags & CALIPSO_EVAL_EQUALITY_JACOBIAN) != 0 && d.ne > 0, jh = (flags & CALIPSO_EVAL_CONE_JACOBIAN) != 0 && d.nc > 0, hl = (flags & hess) != 0;
        return (jg || jh || hl) ? blocks_pack_from(s, s->evalL, s->evalZ, hl, jg, jh) : CALIPSO_OK;
    }
    // blocks written behind our back: an analysed stage-banded structure has to be re-checked against them (as set_field does)
    if (structure_active(s) && (flags & hess)) { const int v = structure_validate(s, 0); if (v < 0) return v; }
    if (structure_active(s) && (flags & CALIPSO_EVAL_EQUALITY_JACOBIAN) && d.ne) { const int v = structure_validate(s, 1); if (v < 0) return v; }
    if (structure_active(s) && (flags & CALIPSO_EVAL_CONE_JACOBIAN) && d.nc) { const int v = structure_validate(s, 2); if (v < 0) return v; }
    blocks_pack(s, (flags & (CALIPSO_EVAL_EQUALITY_JACOBIAN | CALIPSO_EVAL_CONE_JACOBIAN)) != 0, (flags & hess) != 0);
    return CALIPSO_OK;
}
static int evaluate(H* s, calipso_eval_fn eval, void* user, int which, uint32_t flags);
namespace calipso { int evaluate_point(calipso_hip_solver* s, calipso_eval_fn eval, void* user, int which, uint32_t flags) { return evaluate(s, eval, user, which, flags); } }
static int evaluate(H* s, calipso_eval_fn eval, void* user, int which, uint32_t flags) {
    double* pt = point_of(s, which);
    if (s->qp.attached) { launch_qp_evaluate(s, pt, flags); return CALIPSO_OK; }
    if (s->dev_eval || s->dev_block_eval) return device_evaluate(s, pt, flags);
    if (!eval) { s->err = "no evaluation callback and no device evaluator attached"; return CALIPSO_ERR_ARGUMENT; }
    CK(hipMemcpyAsync(s->hpoint.data(), pt, sizeof(double) * s->d.N, hipMemcpyDeviceToHost, s->stream));
    SYNC();
    const double* w = s->hpoint.data();
    const int rc = eval(user, flags, w, w + s->d.oy(), w + s->d.oz(), s->hparams.data());
    if (rc != 0) { s->err = "evaluation callback failed"; return CALIPSO_ERR_CALLBACK; }
    return CALIPSO_OK;
}

static int candidate_merit(H* s, calipso_eval_fn eval, void* user, double* Mh, double* thetah, bool with_dd = false, bool queue_only = false) {   // with_dd: dscal[6] (launch_dot_merit, queued before) travels along
    int rc = evaluate(s, eval, user, 1, CALIPSO_EVAL_OBJECTIVE | CALIPSO_EVAL_EQUALITY | CALIPSO_EVAL_CONE);   // solve.jl:231-235
    if (rc < 0) return rc;
    launch_cone(s, s->candidate, CALIPSO_CONE_BARRIER | CALIPSO_CONE_BARRIER_GRADIENT);                         // :237-240
    launch_merit_and_constraint(s, s->candidate, 4, with_dd ? 3 : 2);                                           // merit + violation in one launch; the kernel publishes: no read-back launch
    if (queue_only) return CALIPSO_OK;                                                                          // (the caller waits for s->pub_seq and reads hscal[4..6])
    if (wait_published(s, s->pub_seq)) return CALIPSO_ERR_HIP;
    *Mh = s->hscal[4]; *thetah = s->hscal[5];
    return CALIPSO_OK;
}

// one pass of the inner loop body of solve! (solve.jl:98-353)
static int inner_iteration(H* s, calipso_eval_fn eval, void* user, double equality_violation, double cone_product_violation, IterInfo& info,
                           double* eq_viol_out, double* cp_viol_out, bool violations_unread = false) {
    const Options& o = s->opt; Scalars& sc = s->sc; const Dims& d = s->d;
    int rc;
    EV(0);
    rc = evaluate(s, eval, user, 0, CALIPSO_EVAL_OBJECTIVE_GRADIENT | CALIPSO_EVAL_EQUALITY_DUAL_GRADIENT | CALIPSO_EVAL_CONE_DUAL_GRADIENT);   // :100-104
    if (rc < 0) return rc;
    launch_cone(s, s->solution, CALIPSO_CONE_BARRIER | CALIPSO_CONE_BARRIER_GRADIENT);   // :106-109
    launch_merit_and_gradient(s);                                                       // :112-116, :118-124 (one launch)
    launch_residual(s);                                                                 // :127
    launch_violations_and_constraint(s, 4, 14);                                         // :130-135 and :170-172 (computed early: one read-back, published by the kernel itself)
    uint32_t fl = CALIPSO_EVAL_OBJECTIVE_HESSIAN | CALIPSO_EVAL_EQUALITY_JACOBIAN | CALIPSO_EVAL_CONE_JACOBIAN;
    if (o.constraint_tensor != 0.0) fl |= CALIPSO_EVAL_EQUALITY_DUAL_HESSIAN | CALIPSO_EVAL_CONE_DUAL_HESSIAN;
    // Queue-ahead (spec_step_ok): when the last step of this handle went on to a search direction with its optimality error well above the exit thresholds, this one
    // very likely does too — :175-181 and IC-1 are queued BEFORE the host has seen the norms (they arrive with the inertia counts, in stream order).  Should an exit test
    // hold after all, the factorisation was for nothing: its traces (regularisation, counters, operands queued on the second stream) are undone.
    const bool spec = spec_step_ok(s);
    const bool ahead = spec && s->spec_ahead_ok;
    const Scalars sc_before = sc;
    int64_t in0[3] = {0, 0, 0};
    int rc0 = 0;
    if (ahead) {
        EV(1);
        rc = evaluate(s, eval, user, 0, fl);                                            // :175-181
        if (rc < 0) return rc;
        EV(2);
        sc.ep = o.primal_regularization_initial; sc.ed = o.dual_regularization_initial;
        rc0 = do_factorize(s, in0, true);                                               // IC-1 of inertia_correction! (its read-back is behind the norms' in the stream)
        if (rc0 < 0) return rc0;
    } else if (wait_published(s, s->pub_seq)) return CALIPSO_ERR_HIP;
    auto undo_ahead = [&] {
        if (!ahead) return;
        if (s->rhs_ahead && s->stream2) (void)hipStreamSynchronize(s->stream2);
        s->rhs_ahead = false; s->rhs_joined = false;
        sc.ep = sc_before.ep; sc.ed = sc_before.ed;
        s->stats.factorizations -= 1; s->phase_ms[8] -= 1.0;
        s->spec_ahead_ok = false;
    };
    const double* hs = s->hscal;
    info.M = hs[4]; info.theta = hs[5];
    info.residual_violation = hs[8] / (double)d.N;
    const double sd = (d.ne + d.nc > 0) ? std::max(100.0, (hs[13] + hs[14]) / (double)(d.ne + d.nc)) / 100.0 : 1.0;   // optimality_error.jl:8
    const double scn = (d.nc > 0) ? std::max(100.0, hs[15] / (double)d.nc) / 100.0 : 1.0;                             // :9
    info.optimality = std::max(std::max(hs[9] / sd, hs[10]), std::max(hs[11], hs[12] / scn));
    info.slack_violation = std::max(hs[10], hs[11]);
    if (!ahead) EV(1);
    if (info.residual_violation < o.residual_tolerance && info.slack_violation < o.slack_tolerance &&
        equality_violation <= o.equality_tolerance && cone_product_violation <= o.complementarity_tolerance) {   // :138-143
        undo_ahead();
        s->spec_ahead_ok = false;
        info.exit_kind = 1;
        return CALIPSO_OK;
    }
    const double exit2 = std::max(o.central_path_update_tolerance * sc_before.kappa, o.optimality_tolerance);
    if (info.optimality <= exit2) {                                                                               // :165
        undo_ahead();
        s->spec_ahead_ok = false;
        info.exit_kind = 2;
        return CALIPSO_OK;
    }
    s->spec_ahead_ok = info.optimality > 4.0 * exit2 && info.residual_violation >= o.residual_tolerance;          // (the next step's prediction)
    if (!ahead) {
        rc = evaluate(s, eval, user, 0, fl);                                            // :175-181
        if (rc < 0) return rc;
        // cone!(jacobian=true) (:183-185): the arrow/diagonal Jacobians are functions of (s, t) and are formed inside the kernels
        EV(2);
    }
    s->time_matvec = true;                                                              // (the first refinement residual of the step is timed: kernel_times [4])
    int warn = do_search_direction(s, &info.nfact, &info.rounds, ahead ? in0 : nullptr, rc0, spec && d.nc > 0);   // :187
    if (warn < 0) return warn;
    double step_size = 1.0, Mh = 0.0, thetah = 0.0;
    bool tail_done = false;
    if (s->refine_pending) {
        // the refinement's rounds are queued, its report not read: the cone search (:190-221), the first candidate — its step sizes taken from the masks on the
        // device — and the candidate's merit / violation (:231-250) go behind them, ONE read-back brings the report, the masks and the three scalars
        EV(3);
        launch_cone_search(s, ++s->pub_seq);
        launch_first_candidate_from_masks(s);
        rc = candidate_merit(s, eval, user, &Mh, &thetah, true, true);
        if (rc < 0) return rc;
        if (wait_published(s, s->pub_seq)) return CALIPSO_ERR_HIP;
        bool step_changed = false;
        const int w2 = search_direction_finish(s, &info.rounds, &step_changed);
        if (w2 < 0) return w2;
        warn = std::max(warn, w2);
        if (!step_changed) {
            const int ks = first_feasible_trial(s->hicount + 6, o.max_cone_line_search), kt = first_feasible_trial(s->hicount + 32, o.max_cone_line_search);
            if (ks < 0 || kt < 0) { s->err = "cone search failure"; return CALIPSO_ERR_CONE_SEARCH; }   // solve.jl:210,220
            double as = 1.0, at = 1.0;                                                   // (what k_first_candidate_masks formed from the same masks)
            for (int k = 0; k < ks; ++k) as = o.scaling_line_search * as;
            for (int k = 0; k < kt; ++k) at = o.scaling_line_search * at;
            info.step_size = as; info.step_size_t = at; step_size = as;
            Mh = s->hscal[4]; thetah = s->hscal[5];
            tail_done = true;
        }
    }
    if (!tail_done) {
        EV(3);
        rc = do_cone_search(s, &info.step_size, &info.step_size_t, false);              // :190-221
        if (rc < 0) return rc;
        step_size = info.step_size;
        // candidate s, t (:206-218), candidate x, r (:224-229) and the directional derivative of the merit function (its gradient is that of :118-124: the
        // point has not moved) in one launch
        launch_first_candidate(s, info.step_size, info.step_size_t);
        rc = candidate_merit(s, eval, user, &Mh, &thetah, true);                        // :231-250
        if (rc < 0) return rc;
    }
    const double dd = s->hscal[6];
    const double M = info.M, theta = info.theta;
    calipso::i64 residual_iteration = 0;
    while (residual_iteration < o.max_residual_line_search) {                           // :254-302
        if (check_filter(s, thetah, Mh)) {
            if (theta <= o.slack_tolerance && switching_condition(step_size, dd, o.merit_exponent, theta, o.violation_exponent, 1.0) &&
                armijo(M, Mh, dd, step_size, o.armijo_tolerance, o.machine_tolerance)) {
                break;
            } else if (sufficient_progress(theta, thetah, M, Mh, o.violation_tolerance, o.merit_tolerance, o.machine_tolerance)) {
                break;
            }
        }
        step_size = o.scaling_line_search * step_size;
        launch_axpy_points(s, step_size, 1);                                            // :268-276
        rc = candidate_merit(s, eval, user, &Mh, &thetah);                              // :278-297
        if (rc < 0) return rc;
        residual_iteration += 1;
    }
    if (residual_iteration >= o.max_residual_line_search) warn = std::max(warn, (int)CALIPSO_WARN_LINE_SEARCH);
    // augment_filter!(solver, ...)  filter.jl:81-89
    if (!switching_condition(step_size, dd, o.merit_exponent, theta, o.violation_exponent, 1.0) ||
        !armijo(M, Mh, dd, step_size, o.armijo_tolerance, o.machine_tolerance))
        augment_filter(s, (1.0 - o.violation_tolerance) * theta, M - o.merit_tolerance * theta);
    launch_accept(s, step_size);                                                        // :309-326
    launch_cone(s, s->solution, CALIPSO_CONE_PRODUCT);                                  // :328-330
    launch_violations(s, 16, 2);                                                        // ||g||inf, ||s o t||inf  :332-333
    // (violations_unread: a benchmark step that another one follows in the same call — nobody reads the two norms, the kernel computes and publishes them all the
    // same, and the next step's first read-back is ordered behind them: no host wait here)
    if (!violations_unread) {
        if (wait_published(s, s->pub_seq)) return CALIPSO_ERR_HIP;
        *eq_viol_out = s->hscal[16]; *cp_viol_out = s->hscal[17];
    }
    EV(4);
    info.step_size = step_size; info.Mh = Mh; info.thetah = thetah;
    s->stats.newton_steps += 1;
    return warn;
}

extern "C" {

int32_t calipso_hip_cone(H* s, int32_t which, int32_t flags) { if (!s) return CALIPSO_ERR_ARGUMENT; launch_cone(s, point_of(s, which), flags); return CALIPSO_OK; }
int32_t calipso_hip_residual(H* s) { if (!s) return CALIPSO_ERR_ARGUMENT; launch_residual(s); return CALIPSO_OK; }

int32_t calipso_hip_violations(H* s, double out[5]) {
    if (!s || !out) return CALIPSO_ERR_ARGUMENT;
    const Dims& d = s->d;
    launch_violations(s);
    if (read_scalars(s, 8, 10)) return CALIPSO_ERR_HIP;
    const double* hs = s->hscal;
    const double sd = (d.ne + d.nc > 0) ? std::max(100.0, (hs[13] + hs[14]) / (double)(d.ne + d.nc)) / 100.0 : 1.0;
    const double scn = (d.nc > 0) ? std::max(100.0, hs[15] / (double)d.nc) / 100.0 : 1.0;
    out[0] = hs[8] / (double)d.N;
    out[1] = std::max(std::max(hs[9] / sd, hs[10]), std::max(hs[11], hs[12] / scn));
    out[2] = std::max(hs[10], hs[11]);
    out[3] = hs[16];
    out[4] = hs[17];
    return CALIPSO_OK;
}

int32_t calipso_hip_residual_jacobian_variables_symmetric(H* s) {
    if (!s) return CALIPSO_ERR_ARGUMENT;
    if (!s->Kdense) { if (dalloc(s, &s->Kdense, (size_t)s->d.n * s->d.n)) return CALIPSO_ERR_HIP; }
    launch_cone_weights(s);
    if (s->compact) {                        // inspection only: dense temporaries of the blocks for the dense assembly kernel
        double *L = nullptr, *Z = nullptr;
        const int rc = blocks_unpack_dense(s, &L, &Z);
        if (rc == CALIPSO_OK) { double* kl = s->Lxx; double* kz = s->Z; s->Lxx = L; s->Z = Z; s->gx = Z; s->hx = Z + s->d.ne; launch_assemble_K(s); (void)hipStreamSynchronize(s->stream); s->Lxx = kl; s->Z = kz; s->gx = kz; s->hx = kz; }
        if (L) (void)hipFree(L);
        if (Z) (void)hipFree(Z);
        return rc;
    }
    launch_assemble_K(s);
    return CALIPSO_OK;
}

int32_t calipso_hip_jacobian_variables_mul(H* s, const double* v, double* out) {
    if (!s || !v || !out) return CALIPSO_ERR_ARGUMENT;
    double* dv = s->vtmp; double* dout = s->vtmp + s->d.N;
    CK(hipMemcpyAsync(dv, v, sizeof(double) * s->d.N, hipMemcpyHostToDevice, s->stream));
    launch_Hmul(s, dv, dout);
    CK(hipMemcpyAsync(out, dout, sizeof(double) * s->d.N, hipMemcpyDeviceToHost, s->stream));
    SYNC();
    return CALIPSO_OK;
}

int32_t calipso_hip_factorize(H* s, int64_t inertia[3]) { if (!s || !inertia) return CALIPSO_ERR_ARGUMENT; return do_factorize(s, inertia); }
int32_t calipso_hip_inertia_correction(H* s, int64_t* nf) { if (!s) return CALIPSO_ERR_ARGUMENT; return do_inertia_correction(s, nf); }
int32_t calipso_hip_residual_symmetric(H* s, int32_t which) { if (!s) return CALIPSO_ERR_ARGUMENT; launch_residual_symmetric(s, which == 0 ? s->residual : s->residual_error); return CALIPSO_OK; }
int32_t calipso_hip_linear_solve(H* s) {
    if (!s) return CALIPSO_ERR_ARGUMENT;
    // b = "residual_symmetric" as currently held by the handle (set by calipso_hip_residual_symmetric or by the caller):
    // rebuild the solve operands from it, solve, and back-substitute into "step_symmetric"
    launch_solve_from_b(s);
    return CALIPSO_OK;
}
int32_t calipso_hip_search_direction_symmetric(H* s, int32_t which) { if (!s) return CALIPSO_ERR_ARGUMENT; do_sds(s, which); return CALIPSO_OK; }
int32_t calipso_hip_iterative_refinement(H* s, int32_t* rounds, double* final_norm) { if (!s) return CALIPSO_ERR_ARGUMENT; int r = 0; int rc = do_refinement(s, &r, final_norm); if (rounds) *rounds = r; return rc; }
int32_t calipso_hip_search_direction_nonsymmetric(H* s) { if (!s) return CALIPSO_ERR_ARGUMENT; return nonsymmetric_solve(s, s->residual, s->step); }
int32_t calipso_hip_search_direction(H* s) { if (!s) return CALIPSO_ERR_ARGUMENT; return do_search_direction(s, nullptr, nullptr); }
int32_t calipso_hip_cone_search(H* s, double* a, double* b) { if (!s || !a || !b) return CALIPSO_ERR_ARGUMENT; return do_cone_search(s, a, b); }

int32_t calipso_hip_cone_violation(H* s, const double* xhat, const double* x, double tau, int32_t* violated) {
    if (!s || !violated || ((!xhat || !x) && s->d.nc)) return CALIPSO_ERR_ARGUMENT;
    *violated = 0;
    if (s->d.nc == 0) return CALIPSO_OK;
    double* a = s->vtmp; double* b = s->vtmp + s->d.nc;
    CK(hipMemcpyAsync(a, xhat, sizeof(double) * s->d.nc, hipMemcpyHostToDevice, s->stream));
    CK(hipMemcpyAsync(b, x, sizeof(double) * s->d.nc, hipMemcpyHostToDevice, s->stream));
    launch_cone_violation_host(s, a, b, tau);
    CK(hipMemcpyAsync(s->hicount + 6, s->icount + 6, sizeof(int), hipMemcpyDeviceToHost, s->stream));
    SYNC();
    *violated = s->hicount[6] != 0;
    return CALIPSO_OK;
}

int32_t calipso_hip_candidate(H* s, double step_size, int32_t with_s) { if (!s) return CALIPSO_ERR_ARGUMENT; launch_axpy_points(s, step_size, with_s); return CALIPSO_OK; }
int32_t calipso_hip_merit(H* s, int32_t which, double* M) { if (!s || !M) return CALIPSO_ERR_ARGUMENT; launch_merit(s, point_of(s, which)); if (read_scalars(s, 4, 1)) return CALIPSO_ERR_HIP; *M = s->hscal[4]; return CALIPSO_OK; }
int32_t calipso_hip_merit_gradient(H* s) { if (!s) return CALIPSO_ERR_ARGUMENT; launch_merit_gradient(s); return CALIPSO_OK; }
int32_t calipso_hip_constraint_violation(H* s, int32_t which, double* th) { if (!s || !th) return CALIPSO_ERR_ARGUMENT; launch_constraint_violation(s, point_of(s, which)); if (read_scalars(s, 5, 1)) return CALIPSO_ERR_HIP; *th = s->hscal[5]; return CALIPSO_OK; }
int32_t calipso_hip_merit_directional(H* s, double* dd) { if (!s || !dd) return CALIPSO_ERR_ARGUMENT; launch_dot_merit(s); if (read_scalars(s, 6, 1)) return CALIPSO_ERR_HIP; *dd = s->hscal[6]; return CALIPSO_OK; }
int32_t calipso_hip_accept(H* s, double a) { if (!s) return CALIPSO_ERR_ARGUMENT; launch_accept(s, a); return CALIPSO_OK; }

int32_t calipso_hip_initialize(H* s, const double* guess) {
    if (!s || !guess) return CALIPSO_ERR_ARGUMENT;
    CK(hipMemcpyAsync(s->solution, guess, sizeof(double) * s->d.nx, hipMemcpyHostToDevice, s->stream));
    SYNC();
    return CALIPSO_OK;
}

int32_t calipso_hip_set_device_evaluator(H* s, calipso_device_eval_fn fn, void* user) {
    if (!s) return CALIPSO_ERR_ARGUMENT;
    // (a structured handle holds no dense ProblemData arrays: the evaluator then writes into dense scratch arrays of the handle — allocated on its first evaluation:
    // nx^2 + (ne + nc) nx doubles — whose entries go into the blocks behind it; what lies outside the declared structure must be zero: device_evaluate checks)
    s->dev_eval = fn; s->dev_eval_user = user;
    if (fn) s->dev_block_eval = nullptr;
    return CALIPSO_OK;
}

int32_t calipso_hip_set_device_block_evaluator(H* s, calipso_device_block_eval_fn fn, void* user) {
    if (!s) return CALIPSO_ERR_ARGUMENT;
    if (!s->compact) return fail_arg(s, "calipso_hip_set_device_block_evaluator: only a structured handle (calipso_hip_create_structured) has blocks to write");
    s->dev_block_eval = fn; s->dev_block_eval_user = user;
    if (fn) s->dev_eval = nullptr;
    return CALIPSO_OK;
}

int32_t calipso_hip_device_evaluate(H* s, int32_t which, uint32_t flags) {
    if (!s) return CALIPSO_ERR_ARGUMENT;
    if (!s->dev_eval && !s->dev_block_eval) { s->err = "no device evaluator installed (calipso_hip_set_device_evaluator)"; return CALIPSO_ERR_ARGUMENT; }
    CK(hipSetDevice(s->device));
    return device_evaluate(s, point_of(s, which), flags);
}

int32_t calipso_hip_set_callbacks(H* s, calipso_callback_fn inner, calipso_callback_fn outer, void* user) {
    if (!s) return CALIPSO_ERR_ARGUMENT;
    s->cb_inner = inner; s->cb_outer = outer; s->cb_user = user;
    return CALIPSO_OK;
}

int32_t calipso_hip_stats(H* s, int64_t out[8]) {
    if (!s || !out) return CALIPSO_ERR_ARGUMENT;
    const Stats& t = s->stats;
    out[0] = t.total_iterations; out[1] = t.outer; out[2] = t.factorizations; out[3] = t.refine_fail; out[4] = t.refine_max;
    out[5] = t.fallbacks; out[6] = t.last_refine; out[7] = t.newton_steps;
    return CALIPSO_OK;
}

// differentiate!  differentiate.jl:1-61
int32_t calipso_hip_differentiate(H* s, calipso_eval_fn eval, void* user) {
    if (!s) return CALIPSO_ERR_ARGUMENT;
    const Dims& d = s->d;
    if (d.np == 0) return CALIPSO_OK;
    int rc = evaluate(s, eval, user, 0, CALIPSO_EVAL_OBJECTIVE_JACOBIAN_PARAMETERS | CALIPSO_EVAL_EQUALITY_JACOBIAN_PARAMETERS |
                                           CALIPSO_EVAL_EQUALITY_DUAL_JACOBIAN_PARAMETERS | CALIPSO_EVAL_CONE_JACOBIAN_PARAMETERS |
                                           CALIPSO_EVAL_CONE_DUAL_JACOBIAN_PARAMETERS);
    if (rc < 0) return rc;
    int64_t in[3];
    rc = do_factorize(s, in);                      // :13-20 (same regularisation as the last search direction)
    if (rc < 0) return rc;
    launch_jacobian_parameters(s);                 // :23
    // :29-58 — the reference solves one condensed system per parameter column (no refinement); here all np columns go through
    // the same factors together: condensation per column, mat-vecs as GEMMs, block triangular solves as TRSMs
    const int p = d.np;
    const size_t NPd = d.NP, M = d.m, n = d.n;
    if (!s->multi_rhs) {
        if (dalloc(s, &s->multi_rhs, (n + 3 * NPd + 2 * M) * (size_t)p) || dalloc(s, &s->dsym_multi, n * (size_t)p)) return CALIPSO_ERR_HIP;
    }
    double* rsymM = s->multi_rhs;                  // n  x p   condensed right-hand sides
    double* xbufM = rsymM + n * p;                 // NP x p   b_x (zero padded) -> dx
    double* uM = xbufM + NPd * p;                  // NP x p   forward-substitution scratch
    double* zM = uM + NPd * p;                     // NP x p
    double* t1M = zM + NPd * p;                    // m  x p   Omega b_m
    double* t2M = t1M + M * p;                     // m  x p   [gx; hx] dx
    launch_residual_symmetric_multi(s, s->jacobian_parameters, p, rsymM, xbufM, t1M);
    // (a handle that works on stage blocks — every structured handle — takes the products block by block, all columns in one launch each; its factor lives in the
    // fronts of the multifrontal LDL^T, which take all columns through the tree together: trsm_multi)
    if (d.m && !blocks_gemm_t(s, t1M, d.m, xbufM, d.NP, p, 1.0)) {
        if (s->compact) { s->err = "calipso_hip_differentiate: the block products are not available on this structured handle"; return CALIPSO_ERR_HIP; }
        gemm(s, d.nx, p, d.m, 1.0, s->Z, d.m, true, t1M, d.m, 1.0, xbufM, d.NP);           // b_x + [gx; hx]' Omega b_m
    }
    trsm_multi(s, xbufM, p, uM, zM);                                                        // dx = S^-1 (...)
    if (d.m && !blocks_gemm_n(s, xbufM, d.NP, t2M, d.m, p)) {
        if (s->compact) { s->err = "calipso_hip_differentiate: the block products are not available on this structured handle"; return CALIPSO_ERR_HIP; }
        gemm(s, d.m, p, d.nx, 1.0, s->Z, d.m, false, xbufM, d.NP, 0.0, t2M, d.m);          // [gx; hx] dx
    }
    launch_recover_multi(s, s->jacobian_parameters, p, rsymM, xbufM, t2M, s->solution_sensitivity, -1.0);   // :54-56 sensitivity = -step
    SYNC();
    return CALIPSO_OK;
}

// solve!(solver)  solve.jl:8-377
int32_t calipso_hip_solve(H* s, calipso_eval_fn eval, void* user) {
    if (!s) return CALIPSO_ERR_ARGUMENT;
    CK(hipSetDevice(s->device));
    Options& o = s->opt; Scalars& sc = s->sc; const Dims& d = s->d;
    (void)hipGetLastError();      // (launch_errors: this call's launches only)
    s->stats = Stats();
    s->spec_ahead_ok = false;
    int rc;
    if (o.warmstart == 0.0) {
        rc = evaluate(s, eval, user, 0, CALIPSO_EVAL_EQUALITY | CALIPSO_EVAL_CONE);      // initialize_slacks! initialize.jl:15-29
        if (rc < 0) return rc;
        launch_init_point(s);                                                           // + initialize_duals! :31-36
    }
    sc.kappa = o.central_path_initial; sc.tau = std::max(0.99, 1.0 - sc.kappa);          // initialize.jl:38-42
    sc.rho = o.penalty_initial;                                                         // :44-48
    {
        std::vector<double> l0((size_t)std::max(1, d.ne), o.dual_initial);
        if (d.ne) CK(hipMemcpyAsync(s->lambda, l0.data(), sizeof(double) * d.ne, hipMemcpyHostToDevice, s->stream));
        SYNC();
    }
    calipso::i64 total_iterations = 1;
    rc = evaluate(s, eval, user, 0, CALIPSO_EVAL_OBJECTIVE | CALIPSO_EVAL_EQUALITY | CALIPSO_EVAL_EQUALITY_JACOBIAN | CALIPSO_EVAL_CONE);   // :78-83
    if (rc < 0) return rc;
    launch_violations(s);
    if (read_scalars(s, 16, 2)) return CALIPSO_ERR_HIP;
    double equality_violation = s->hscal[16];            // :85
    double cone_product_violation = s->hscal[17];        // :86 — read BEFORE cone!(product) below, i.e. stale on first use (reference quirk)
    launch_cone(s, s->solution, CALIPSO_CONE_PRODUCT | CALIPSO_CONE_TARGET);   // :88-91
    filter_reset(s);                                      // :95
    int worst = 0;
    for (calipso::i64 j = 1; j <= o.max_outer_iterations; ++j) {
        s->stats.outer = j;
        for (calipso::i64 i = 1; i <= o.max_residual_iterations; ++i) {
            IterInfo info;
            rc = inner_iteration(s, eval, user, equality_violation, cone_product_violation, info, &equality_violation, &cone_product_violation);
            if (rc < 0) return rc;
            worst = std::max(worst, rc);
            if (info.exit_kind == 1) {
                if (o.differentiate != 0.0 && d.np > 0) { rc = calipso_hip_differentiate(s, eval, user); if (rc < 0) return rc; }
                s->stats.total_iterations = total_iterations;
                SYNC();
                return 1;
            }
            if (info.exit_kind == 2) break;
            if (s->cb_inner) { SYNC(); s->cb_inner(s->cb_user, s); }     // callback_inner(custom, solver)  solve.jl:350
            total_iterations += 1;
            s->stats.total_iterations = total_iterations;
        }
        sc.kappa = std::max(o.residual_tolerance / 10.0, std::min(o.central_path_scaling * sc.kappa, std::pow(sc.kappa, o.central_path_exponent)));   // :356
        sc.tau = std::max(0.99, 1.0 - sc.kappa);                                         // :359
        launch_lambda_update(s);                                                        // :362-364
        sc.rho = std::min(std::max(o.penalty_scaling * sc.rho, 1.0 / sc.kappa), o.max_penalty);   // :365
        filter_reset(s);                                                                // :368
        if (s->cb_outer) { SYNC(); s->cb_outer(s->cb_user, s); }         // callback_outer(custom, solver)  solve.jl:371
    }
    s->stats.total_iterations = total_iterations;
    SYNC();
    return 0;
}

// ---- device QP evaluator -----------------------------------------------------------------------------------------------------
__global__ void k_scale_copy(const double* __restrict__ src, double* __restrict__ dst, size_t n, double a) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = a * src[i];
}

__global__ void k_qp_install(Dims d, const double* __restrict__ Gtmp, const double* __restrict__ b, const double* __restrict__ h,
                             double* __restrict__ hx, double* __restrict__ bh) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (size_t)d.nc * d.nx) { const int c = (int)(i % d.nc); const size_t col = i / d.nc; hx[c + col * d.m] = -Gtmp[i]; }   // hx = -G into the stacked Jacobian
    if (i < (size_t)d.ne) bh[i] = -b[i];
    if (i < (size_t)d.nc) bh[d.ne + i] = h[i];
}

int32_t calipso_hip_qp_attach(H* s, const double* P, const double* q, const double* A, const double* b, const double* G, const double* h,
                              double objective_scale) {
    if (!s || !P || !q) return CALIPSO_ERR_ARGUMENT;
    const Dims& d = s->d;
    if ((d.ne && (!A || !b)) || (d.nc && (!G || !h))) return CALIPSO_ERR_ARGUMENT;
    CK(hipSetDevice(s->device));
    if (s->compact) {
        // structured handle: Lxx = 2c P, gx = A, hx = -G straight into the blocks (packed on the host; entries outside the declared structure are an error)
        int rc = blocks_upload_dense(s, 0, P, 2.0 * objective_scale);
        if (rc == CALIPSO_OK && d.ne) rc = blocks_upload_dense(s, 1, A, 1.0);
        if (rc == CALIPSO_OK && d.nc) rc = blocks_upload_dense(s, 2, G, -1.0);
        if (rc < 0) return rc;
        std::vector<double> bh((size_t)d.m);
        for (int k = 0; k < d.ne; ++k) bh[(size_t)k] = -b[k];
        for (int k = 0; k < d.nc; ++k) bh[(size_t)d.ne + k] = h[k];
        CK(hipMemcpyAsync(s->qp.q, q, sizeof(double) * d.nx, hipMemcpyHostToDevice, s->stream));
        if (d.m) CK(hipMemcpyAsync(s->qp.bh, bh.data(), sizeof(double) * d.m, hipMemcpyHostToDevice, s->stream));
        SYNC();
        s->qp.attached = true; s->qp.scale = objective_scale;
        return CALIPSO_OK;
    }
    if (structure_active(s)) { const int rc = calipso_hip_clear_structure(s); if (rc < 0) return rc; }   // new blocks: any analysed structure is void
    const size_t nx = d.nx;
    // Lxx = 2c P ; gx = A ; hx = -G   (constant Hessian / Jacobians of the QP); bh = [-b; h]
    CK(hipMemcpyAsync(s->S, P, sizeof(double) * nx * nx, hipMemcpyHostToDevice, s->stream));   // S is free before the first factorisation
    hipLaunchKernelGGL(k_scale_copy, dim3((unsigned)((nx * nx + 255) / 256)), dim3(256), 0, s->stream, s->S, s->Lxx, nx * nx, 2.0 * objective_scale);
    CK(hipMemcpyAsync(s->qp.q, q, sizeof(double) * nx, hipMemcpyHostToDevice, s->stream));
    if (d.ne) {
        CK(hipMemcpy2DAsync(s->gx, sizeof(double) * d.m, A, sizeof(double) * d.ne, sizeof(double) * d.ne, nx, hipMemcpyHostToDevice, s->stream));
        CK(hipMemcpyAsync(s->t1, b, sizeof(double) * d.ne, hipMemcpyHostToDevice, s->stream));
    }
    if (d.nc) {
        CK(hipMemcpyAsync(s->WH, G, sizeof(double) * d.nc * nx, hipMemcpyHostToDevice, s->stream));
        CK(hipMemcpyAsync(s->t2, h, sizeof(double) * d.nc, hipMemcpyHostToDevice, s->stream));
    }
    const size_t work = std::max<size_t>((size_t)d.nc * nx, (size_t)d.m);
    if (work) hipLaunchKernelGGL(k_qp_install, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, s->stream, d, s->WH, s->t1, s->t2, s->hx, s->qp.bh);
    SYNC();
    s->hessian_dirty = true;
    s->qp.attached = true;
    s->qp.scale = objective_scale;
    return CALIPSO_OK;
}

int32_t calipso_hip_qp_evaluate(H* s, int32_t which, uint32_t flags) {
    if (!s || !s->qp.attached) return CALIPSO_ERR_ARGUMENT;
    launch_qp_evaluate(s, point_of(s, which), flags);
    return CALIPSO_OK;
}

static int32_t newton_step_impl(H* s, int32_t advance, double info_out[6], bool last);
int32_t calipso_hip_newton_step(H* s, int32_t advance, double info_out[6]) { return newton_step_impl(s, advance, info_out, true); }
// `count` steps in ONE call: what a host loop of calipso_hip_newton_step calls does, without returning to the host language between the steps (no stream
// synchronisation and no event queries between them either: the steps wait for what they need through the published words, as a step does internally)
int32_t calipso_hip_newton_steps(H* s, int32_t count, int32_t advance, double* info_out, int32_t* status_out) {
    if (!s || count < 0 || (count > 0 && !status_out)) return CALIPSO_ERR_ARGUMENT;
    for (int32_t k = 0; k < count; ++k) {
        const int32_t rc = newton_step_impl(s, advance, info_out ? info_out + 6 * (size_t)k : nullptr, k + 1 == count);
        status_out[k] = rc;
        if (rc < 0) { CK(hipStreamSynchronize(s->stream)); return rc; }
    }
    return CALIPSO_OK;
}
static int32_t newton_step_impl(H* s, int32_t advance, double info_out[6], bool last) {
    if (!s) return CALIPSO_ERR_ARGUMENT;
    (void)hipGetLastError();      // (launch_errors reports what THIS call's launches leave behind, not an earlier call's)
    if (!s->qp.attached && !s->dev_eval && !s->dev_block_eval) { s->err = "calipso_hip_newton_step needs a device evaluator (calipso_hip_qp_attach or calipso_hip_set_device_evaluator)"; return CALIPSO_ERR_ARGUMENT; }
    const Dims& d = s->d;
    const Scalars saved_sc = s->sc;
    std::vector<double> ft, fm; calipso::i64 fi = 0;
    if (!advance) {
        {
            double* const dst[4] = {s->saved_point, s->saved_g, s->saved_h, s->dscal + 32};
            const double* const src[4] = {s->solution, s->g, s->hc, s->dscal};
            const size_t n[4] = {(size_t)d.N, (size_t)d.ne, (size_t)d.nc, 2};
            copy4_d(s, dst, src, n);
        }
        ft = s->filter_theta; fm = s->filter_merit; fi = s->filter_index;
    }
    IterInfo info;
    double ev = 1.0e30, cv = 1.0e30;   // never "converged": the benchmark step always computes a direction
    EV(8);
    int rc = inner_iteration(s, nullptr, nullptr, ev, cv, info, &ev, &cv, !last);
    EV(9);
    if (rc < 0) return rc;
    if (!advance) {
        {
            double* const dst[4] = {s->solution, s->g, s->hc, s->dscal};
            const double* const src[4] = {s->saved_point, s->saved_g, s->saved_h, s->dscal + 32};
            const size_t n[4] = {(size_t)d.N, (size_t)d.ne, (size_t)d.nc, 2};
            copy4_d(s, dst, src, n);
        }
        launch_cone(s, s->solution, CALIPSO_CONE_PRODUCT);
        s->filter_theta = ft; s->filter_merit = fm; s->filter_index = fi;
        const double keep_ep = s->sc.ep, keep_ed = s->sc.ed;
        s->sc = saved_sc; s->sc.ep = keep_ep; s->sc.ed = keep_ed;   // eps_last restored: every benchmark step repeats IC-1
    }
    if (info_out) {
        info_out[0] = info.step_size; info_out[1] = info.step_size_t; info_out[2] = info.rounds; info_out[3] = (double)info.nfact;
        info_out[4] = info.Mh; info_out[5] = info.thetah;
    }
    if (!last) return rc;             // (calipso_hip_newton_steps: the next step follows in stream order)
    SYNC();
    float ms = 0.f;
    if (info.exit_kind == 0) {
        (void)hipEventElapsedTime(&ms, s->ev[0], s->ev[1]); s->phase_ms[0] = ms;
        (void)hipEventElapsedTime(&ms, s->ev[2], s->ev[3]); s->phase_ms[2] = ms;   // search direction (factor + solve + refine)
        (void)hipEventElapsedTime(&ms, s->ev[3], s->ev[4]); s->phase_ms[5] = ms;
    }
    (void)hipEventElapsedTime(&ms, s->ev[8], s->ev[9]); s->phase_ms[6] = ms;
    return rc;
}

}  // extern "C"

extern "C" {

int32_t calipso_hip_phase_times(H* s, double out[9]) {
    if (!s || !out) return CALIPSO_ERR_ARGUMENT;
    factor_times(s);
    for (int i = 0; i < 9; ++i) out[i] = s->phase_ms[i];
    return CALIPSO_OK;
}

int32_t calipso_hip_kernel_times(H* s, double out[8]) {
    if (!s || !out) return CALIPSO_ERR_ARGUMENT;
    for (int i = 0; i < 8; ++i) out[i] = 0.0;
    factor_times(s);
    out[0] = s->kernel_ms[0];
    out[1] = (double)s->ldl_step_launches;              // k_ldl_diag + the k_ldl_step launches the last blocked factorisation queued
    out[2] = (double)s->d.NP;
    out[3] = (double)(s->slab_doubles * sizeof(double) + s->scratch_bytes);
    { double lf[8]; calipso::lfac_describe(s, lf); out[6] = s->lfac_last ? 1.0 : (s->lfac_failed ? -1.0 : 0.0); out[7] = lf[5]; }      // [6] = -1: the left-looking schedule was wanted and could not be had (no plan / no memory; calipso_hip_last_error says which)
    if (s->matvec_timed && hipEventSynchronize(s->ev[6]) == hipSuccess) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s->ev[5], s->ev[6]) == hipSuccess) { out[4] = ms; out[5] = 8.0 * ((double)s->d.m * s->d.nx + (double)s->d.nx * s->d.nx); }
    }
    return CALIPSO_OK;
}

// Work of one Newton step on a handle that uses its stage structure (stage blocks and / or the multifrontal factorisation of S): what bench.py prices such
// handles with (the dense nx^3 / 3 says nothing about a factorisation over the stage tree).
int32_t calipso_hip_structure_work(H* s, double out[8]) {
    if (!s || !out) return CALIPSO_ERR_ARGUMENT;
    for (int i = 0; i < 8; ++i) out[i] = 0.0;
    if (s->blocks.on) { out[0] = s->blocks.schur_flops; out[1] = (double)s->blocks.packed; out[2] = (double)s->blocks.npairs; }
    if (s->stage_parallel && s->spS) { double w[3]; calipso::sparse_work(s->spS, w); out[3] = w[0]; out[4] = w[1]; out[5] = w[2]; }
    out[6] = s->compact ? 1.0 : 0.0;
    return CALIPSO_OK;
}

int32_t calipso_hip_splitmix_uniform(uint64_t problem_id, uint64_t stream_id, double lo, double hi, int64_t count, double* out) {
    if (!out && count > 0) return CALIPSO_ERR_ARGUMENT;
    uint64_t state = 0xCA11B50000000000ULL + 4096ULL * problem_id + stream_id;
    for (int64_t i = 0; i < count; ++i) {
        state += 0x9E3779B97F4A7C15ULL;
        uint64_t z = state;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        z = z ^ (z >> 31);
        out[i] = lo + (hi - lo) * ((double)(z >> 11) * (1.0 / 9007199254740992.0));
    }
    return CALIPSO_OK;
}

}  // extern "C"
