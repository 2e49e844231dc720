// group.hip — lockstep stepping / solving of several same-shape handles through the same kernel launches.
#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>

#include "internal.hpp"
#include "device_utils.hpp"
#include "host_logic.hpp"

using namespace calipso;

// =============================================================================================================================
// Groups: several handles of ONE shape stepped in lockstep through the same launches (BASELINE config C4: many independent
// problem instances per GPU).  Every kernel takes the instance from blockIdx.z (internal.hpp: Batch), so the latency-bound
// launches of the step (the pivot chain of the LDL^T, the triangular solves, the O(N) vector kernels) cost the same for B
// instances as for one.  The host logic below is the per-instance logic of inner_iteration() applied to every member, with
// the members that need another round (re-factorisation, refinement, back-tracking) forming the next launch's instance list.
// Results per member are bit-identical to stepping that member alone (tests/test_gpu_group.py).
// =============================================================================================================================
struct calipso_hip_group {
    std::vector<H*> hs;
    H* base = nullptr;                 // hs[0]: its stream carries every launch, its buffers are the address origin
    calipso::BatchSc desc;             // instance list of the launches being enqueued (base->cur points here)
    double *hgather = nullptr, *hgather_dev = nullptr;   // MAX_BATCH x 64 doubles: pinned host buffer and its device-side address
    // read-backs without hipStreamSynchronize: the gather kernel's last workgroup stores a sequence number behind the data (system-scope release) and the host spins on it
    // (host_logic.hpp: host_wait) — what a single handle's read-backs do (api.hip: wait_published): a stream synchronisation costs ~5 us more per read-back, and a group step
    // has ten of them on its critical path
    unsigned long long *hseq = nullptr, *hseq_dev = nullptr; unsigned long long seq = 0;
    unsigned* ticket = nullptr;                          // device word: workgroups of a gather launch that have stored their rows
    int *higather = nullptr, *higather_dev = nullptr;    // MAX_BATCH x 64 ints
    // the per-member scalars of the launches (BatchSc::sctab): a ring of device tables, each uploaded from its pinned twin in stream order when the active set or a
    // member's scalars differ from what the current table holds (steady Newton steps re-use one table: the same members, the same kappa / rho / regularisation)
    static constexpr int SC_RING = 8;
    calipso::Scalars *sc_dev = nullptr, *sc_pin = nullptr;   // SC_RING x MAX_BATCH each
    int sc_slot = 0, sc_n = -1, sc_pending = 0;              // the table in use, how many members it holds (-1: none yet), uploads since the last synchronisation
    std::vector<calipso_eval_fn> evals;              // host evaluation callbacks of the members without a device evaluator
    std::vector<void*> users;
    int saved_band = 0, saved_hb = 0;                // the base handle's own structure while a group call overrides it
    bool dead = false;                               // a member was destroyed: every further call fails (no dangling handle is touched)
    std::string err;
};
typedef calipso_hip_group G;
typedef std::vector<int> Set;          // member indices

// (hseq != nullptr: the last workgroup of the launch to arrive publishes `seq`)
__device__ __forceinline__ void gather_publish(unsigned* __restrict__ ticket, unsigned long long* __restrict__ hseq, unsigned long long seq) {
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0 && hseq) {
        if (atomicAdd(ticket, 1u) == gridDim.z - 1) {
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence_system();
            __hip_atomic_store(hseq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
__global__ void k_gather_d(Batch bt, const double* __restrict__ src, int count, double* __restrict__ dst, unsigned* __restrict__ ticket = nullptr,
                           unsigned long long* __restrict__ hseq = nullptr, unsigned long long seq = 0) {
    inst_shift(bt, src);
    if ((int)threadIdx.x < count) dst[blockIdx.z * 64 + threadIdx.x] = src[threadIdx.x];
    gather_publish(ticket, hseq, seq);
}
__global__ void k_gather_i(Batch bt, const int* __restrict__ src, int count, int* __restrict__ dst, unsigned* __restrict__ ticket = nullptr,
                           unsigned long long* __restrict__ hseq = nullptr, unsigned long long seq = 0) {
    inst_shift_i(bt, src);
    if ((int)threadIdx.x < count) dst[blockIdx.z * 64 + threadIdx.x] = src[threadIdx.x];
    gather_publish(ticket, hseq, seq);
}
// wait for the sequence number of the last gather launch (everything queued before it on the group's stream has completed by then)
static int g_wait_seq(calipso_hip_group* g);

static void g_activate(G* g, const Set& a) {
    BatchSc& b = g->desc;
    b.b.n = (int)a.size();
    const calipso::Scalars* cur = g->sc_n == (int)a.size() ? g->sc_pin + (size_t)g->sc_slot * MAX_BATCH : nullptr;
    bool same = cur != nullptr;
    for (size_t k = 0; k < a.size(); ++k) {
        H* h = g->hs[a[k]];
        b.b.delta[k] = (long long)((reinterpret_cast<intptr_t>(h->slab) - reinterpret_cast<intptr_t>(g->base->slab)) / (intptr_t)sizeof(double));
        b.b.slot[k] = (int)a[k];
        if (same && std::memcmp(&cur[k], &h->sc, sizeof(calipso::Scalars)) != 0) same = false;
    }
    if (!same) {
        // a new table: the next ring slot, filled on the host and copied in stream order (the launches queued so far keep reading the previous one).  A pinned
        // twin is only re-used SC_RING uploads later; should that many be pending without a synchronisation in between, drain the stream first
        if (++g->sc_pending >= G::SC_RING - 1) { (void)hipStreamSynchronize(g->base->stream); g->sc_pending = 0; }
        g->sc_slot = (g->sc_slot + 1) % G::SC_RING;
        calipso::Scalars* pin = g->sc_pin + (size_t)g->sc_slot * MAX_BATCH;
        for (size_t k = 0; k < a.size(); ++k) pin[k] = g->hs[a[k]]->sc;
        (void)hipMemcpyAsync(g->sc_dev + (size_t)g->sc_slot * MAX_BATCH, pin, sizeof(calipso::Scalars) * a.size(), hipMemcpyHostToDevice, g->base->stream);
        g->sc_n = (int)a.size();
    }
    b.sctab = g->sc_dev + (size_t)g->sc_slot * MAX_BATCH;
    g->base->cur = &g->desc;
}
static int g_wait_seq(G* g) {
    H* s = g->base;
    static const bool sync_env = [] { const char* e = getenv("CALIPSO_HIP_GROUP_SYNC"); return e && atoi(e) != 0; }();      // (experiment switch: the stream synchronisation of rounds 1-5)
    if (sync_env) { SYNC(); g->sc_pending = 0; return 0; }
    const unsigned long long want = g->seq;
    hipError_t q = hipSuccess;
    const bool ok = host_wait([&] { return __atomic_load_n(g->hseq, __ATOMIC_ACQUIRE) >= want; },
                              [&] { q = hipStreamQuery(s->stream); return q == hipErrorNotReady; });
    g->sc_pending = 0;
    if (ok) return 0;
    if (q != hipSuccess && q != hipErrorNotReady) return calipso::check(s, q, "group read-back");
    s->err = "group read-back did not arrive";
    return CALIPSO_ERR_HIP;
}
// dscal[first .. first+count) of every member of `a` (the active set) -> that member's hscal
static int g_read_d(G* g, const Set& a, int first, int count) {
    H* s = g->base;
    if (launch_errors(s, "a kernel launch of this phase of the group step was refused")) return CALIPSO_ERR_HIP;
    // the gather kernel stores straight into pinned host memory (one launch, no separate copy)
    hipLaunchKernelGGL(k_gather_d, dim3(1, 1, (unsigned)a.size()), dim3(64), 0, s->stream, g->desc.b, s->dscal + first, count, g->hgather_dev, g->ticket, g->hseq_dev, ++g->seq);
    if (g_wait_seq(g)) return CALIPSO_ERR_HIP;
    for (size_t k = 0; k < a.size(); ++k)
        for (int i = 0; i < count; ++i) g->hs[a[k]]->hscal[first + i] = g->hgather[k * 64 + i];
    return 0;
}
static int g_read_i(G* g, const Set& a, int first, int count) {
    H* s = g->base;
    if (launch_errors(s, "a kernel launch of this phase of the group step was refused")) return CALIPSO_ERR_HIP;
    hipLaunchKernelGGL(k_gather_i, dim3(1, 1, (unsigned)a.size()), dim3(64), 0, s->stream, g->desc.b, s->icount + first, count, g->higather_dev, g->ticket, g->hseq_dev, ++g->seq);
    if (g_wait_seq(g)) return CALIPSO_ERR_HIP;
    for (size_t k = 0; k < a.size(); ++k)
        for (int i = 0; i < count; ++i) g->hs[a[k]]->hicount[first + i] = g->higather[k * 64 + i];
    return 0;
}

// icount[ifirst ..) and dscal[dfirst ..) of every member of `a` with ONE stream synchronisation (two gather launches)
static int g_read_both(G* g, const Set& a, int ifirst, int icnt, int dfirst, int dcnt) {
    H* s = g->base;
    if (launch_errors(s, "a kernel launch of this phase of the group step was refused")) return CALIPSO_ERR_HIP;
    hipLaunchKernelGGL(k_gather_i, dim3(1, 1, (unsigned)a.size()), dim3(64), 0, s->stream, g->desc.b, s->icount + ifirst, icnt, g->higather_dev);
    hipLaunchKernelGGL(k_gather_d, dim3(1, 1, (unsigned)a.size()), dim3(64), 0, s->stream, g->desc.b, s->dscal + dfirst, dcnt, g->hgather_dev, g->ticket, g->hseq_dev, ++g->seq);
    if (g_wait_seq(g)) return CALIPSO_ERR_HIP;
    for (size_t k = 0; k < a.size(); ++k) {
        for (int i = 0; i < icnt; ++i) g->hs[a[k]]->hicount[ifirst + i] = g->higather[k * 64 + i];
        for (int i = 0; i < dcnt; ++i) g->hs[a[k]]->hscal[dfirst + i] = g->hgather[k * 64 + i];
    }
    return 0;
}

// evaluate! for the members of `a` at their current (which = 0) or candidate (1) point: one batched launch sequence for the members
// with a device evaluator, the host callback (on the member's own stream, after the group's stream has drained) for the others
static int gb_evaluate(G* g, const Set& a, int which, uint32_t flags) {
    H* s = g->base;
    Set dev, user_dev, host;
    for (int i : a) (g->hs[i]->qp.attached ? dev : ((g->hs[i]->dev_eval || g->hs[i]->dev_block_eval) ? user_dev : host)).push_back(i);
    if (!dev.empty()) {
        g_activate(g, dev);
        launch_qp_evaluate(s, which == 0 ? s->solution : s->candidate, flags);
    }
    // members with a user device evaluator: their kernels are enqueued on the GROUP's stream (the member's handle borrows it for the
    // call), one member after the other, without leaving the device
    for (int i : user_dev) {
        H* h = g->hs[i];
        hipStream_t own = h->stream;
        h->stream = s->stream;
        const int rc = evaluate_point(h, nullptr, nullptr, which, flags);
        h->stream = own;
        if (rc < 0) { s->err = "member " + std::to_string(i) + ": " + h->err; return rc; }
    }
    if (!host.empty()) {
        SYNC();
        for (int i : host) {
            H* h = g->hs[i];
            const int rc = evaluate_point(h, i < (int)g->evals.size() ? g->evals[i] : nullptr, i < (int)g->users.size() ? g->users[i] : nullptr, which, flags);
            if (rc < 0) { s->err = "member " + std::to_string(i) + ": " + h->err; return rc; }
            CK(hipStreamSynchronize(h->stream));      // the callback's uploads are complete before the group's stream goes on
        }
    }
    g_activate(g, a);
    return CALIPSO_OK;
}

// the band the group's launches use: the widest of its members' (a launch covers all of them); dense if any member is dense
static void g_effective_band(G* g) {
    H* s = g->base;
    int band = 0, hb = 0;
    bool dense = false;
    for (H* h : g->hs) { if (h->band64 == 0) dense = true; band = std::max(band, h->band64); hb = std::max(hb, h->half_bandwidth); }
    g->saved_band = s->band64; g->saved_hb = s->half_bandwidth;
    s->band64 = dense ? 0 : band; s->half_bandwidth = dense ? 0 : hb;
}
static void g_restore_band(G* g) { g->base->band64 = g->saved_band; g->base->half_bandwidth = g->saved_hb; g->base->blocks_effective = true; }
// stage blocks (blocks.hip): the group's launches walk the block tables of the base handle, so either every member uses blocks of the same structure
// or none does.  A mixed group is an error (the slab region of Lsym means different things on its members).
static int g_effective_blocks(G* g) {
    H* s = g->base;
    bool any = false, all = true;
    for (H* h : g->hs) { any = any || h->blocks.on; all = all && h->blocks.on && h->blocks.signature == s->blocks.signature; }
    if (any && !all) { s->err = "the members of a group must agree on calipso_hip_set_stage_blocks (all on with one block structure, or all off)"; return CALIPSO_ERR_ARGUMENT; }
    for (H* h : g->hs)     // the inverse blocks of the solves are laid out per member for ITS "opt.solve_block"; the group's launches use the base handle's
        if (h->solve_block != s->solve_block) { s->err = "the members of a group must agree on opt.solve_block"; return CALIPSO_ERR_ARGUMENT; }
    for (H* h : g->hs)
        if (h->solve_wform != s->solve_wform) { s->err = "the members of a group must agree on opt.solve_wform"; return CALIPSO_ERR_ARGUMENT; }
    s->blocks_effective = true;
    return CALIPSO_OK;
}

// factorize! + compute_inertia! for the members of `a`
static int gb_factorize(G* g, const Set& a, std::vector<std::array<int64_t, 3>>& in) {
    H* s = g->base;
    {   // Lsym of the members whose Hessian was re-uploaded since the last factorisation
        Set dirty;
        for (int i : a) if (g->hs[i]->hessian_dirty) dirty.push_back(i);
        if (!dirty.empty()) { g_activate(g, dirty); launch_symmetrize(s); for (int i : dirty) g->hs[i]->hessian_dirty = false; }
    }
    g_activate(g, a);
    (void)hipEventRecord(s->ev[10], s->stream);
    launch_cone_weights(s);
    launch_scale_rows(s);
    calipso::inject_refused_launch(s->stream);
    (void)hipEventRecord(s->ev[11], s->stream);
    launch_schur(s);
    (void)hipEventRecord(s->ev[12], s->stream);
    launch_ldl(s);
    (void)hipEventRecord(s->ev[13], s->stream);
    if (g_read_i(g, a, 0, 6)) return CALIPSO_ERR_HIP;
    {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, s->ev[10], s->ev[11]); s->phase_ms[1] = ms;
        (void)hipEventElapsedTime(&ms, s->ev[11], s->ev[12]); s->phase_ms[7] = ms;
        (void)hipEventElapsedTime(&ms, s->ev[12], s->ev[13]); s->phase_ms[3] = ms;
        (void)hipEventElapsedTime(&ms, s->ev[12], s->ev[14]); s->kernel_ms[0] = ms;   // of which the panel steps (the pivot chain)
        s->phase_ms[8] += 1.0;
    }
    for (int i : a) {
        H* h = g->hs[i];
        h->stats.factorizations += 1;
        const int64_t pos = h->hicount[0] + h->hicount[3], nonpos = h->hicount[1] + h->hicount[4], zero = h->hicount[2] + h->hicount[5];
        in[i] = {pos, nonpos, zero};
        if (zero > 0) in[i][0] = -1;
    }
    return CALIPSO_OK;
}

// inertia_correction! (inertia.jl:30-80) for every member of `a`; rc[i] < 0 marks a member that failed
static int gb_inertia_correction(G* g, const Set& a, std::vector<int>& rc, std::vector<int64_t>& nfact) {
    std::vector<std::array<int64_t, 3>> in(g->hs.size());
    for (int i : a) { H* h = g->hs[i]; h->sc.ep = h->opt.primal_regularization_initial; h->sc.ed = h->opt.dual_regularization_initial; nfact[i] = 0; }
    int e = gb_factorize(g, a, in);                              // IC-1
    if (e < 0) return e;
    Set pending;
    for (int i : a) {
        H* h = g->hs[i]; nfact[i] += 1;
        if (inertia_ok(h, in[i].data())) continue;
        Options& o = h->opt; Scalars& sc = h->sc;
        if (in[i][2] != 0) sc.ed = o.dual_regularization * std::pow(sc.kappa, o.dual_regularization_exponent);   // IC-2
        sc.ep = std::max(o.min_regularization, o.scaling_regularization_last * sc.ep_last);                     // IC-3
        pending.push_back(i);
    }
    while (!pending.empty()) {
        e = gb_factorize(g, pending, in);                        // IC-4
        if (e < 0) return e;
        Set next;
        for (int i : pending) {
            H* h = g->hs[i]; nfact[i] += 1;
            Options& o = h->opt; Scalars& sc = h->sc;
            if (inertia_ok(h, in[i].data())) { sc.ep_last = sc.ep; continue; }
            if (sc.ep_last == 0.0) sc.ep = o.scaling_regularization_initial * sc.ep;   // IC-5
            else sc.ep = o.scaling_regularization * sc.ep;
            if (sc.ep > o.max_regularization) { h->err = "inertia correction failure"; rc[i] = CALIPSO_ERR_INERTIA; continue; }   // IC-6
            next.push_back(i);
        }
        pending.swap(next);
    }
    return CALIPSO_OK;
}

static void gb_sds(G* g, const Set& a, int which, bool accumulate, bool refine_follows = false) {
    H* s = g->base;
    g_activate(g, a);
    const double* res = which == 0 ? s->residual : s->residual_error;
    launch_residual_symmetric(s, res);
    const bool tail = which == 0 && !accumulate && s->d.m > 0;          // (api.hip: do_sds)
    linear_solve_device(s, !tail);
    if (tail && launch_solve_tail(s, 0, false, refine_follows)) return;
    if (tail) gemv_n(s, s->d.m, s->d.nx, s->Z, s->d.m, s->xbuf, s->t2, 1.0, 0.0, SP_Z);
    launch_recover(s, which == 0 ? s->step : s->step_correction, res, accumulate ? s->step : nullptr, which == 0 ? 1 : 0);   // which = 0: zsx = [gx; hx] step_x
}
// the two halves of a refinement round for the active members (api.hip: refine_residual / refine_solve)
static void gb_refine_residual(G* g) {
    H* s = g->base; const Dims& d = s->d;
    launch_refine_local(s);
    const int nchunk = gemv_refine_pair(s, s->step + d.oy(), s->t1, s->w1, s->w2, s->step, s->lxv, true);      // (api.hip: refine_residual)
    if (nchunk > 0) launch_refine_x_fused(s, false, nchunk); else launch_refine_x(s);
}
static void gb_refine_solve(G* g) {
    H* s = g->base; const Dims& d = s->d;
    launch_trsv(s, s->xbuf);
    if (d.m && launch_solve_tail(s, 1, true, true)) return;       // (api.hip: refine_solve)
    if (d.m) gemv_n(s, d.m, d.nx, s->Z, d.m, s->xbuf, s->t2, 1.0, 0.0, SP_Z);
    launch_recover(s, s->step_correction, s->residual_error, s->step, 2);
}

// iterative_refinement! (iterative_refinement.jl:1-52) for every member of `a` (called right after gb_sds(g, a, 0, ...): zsx is valid)
static int gb_refinement(G* g, const Set& a, std::vector<int>& rc, std::vector<int>& rounds) {
    H* s = g->base;
    const size_t B = g->hs.size();
    std::vector<double> norm(B, 0.0), norm0(B, 0.0);
    std::vector<int> it(B, 0);
    g_activate(g, a);
    gb_refine_residual(g);
    if (g_read_d(g, a, 7, 1)) return CALIPSO_ERR_HIP;
    for (int i : a) { norm[i] = g->hs[i]->hscal[7]; norm0[i] = norm[i]; }
    Set run = a;
    bool first_pass = true;
    while (!run.empty()) {
        Set sub, none;
        for (int i : run) {
            H* h = g->hs[i]; const Options& o = h->opt;
            bool finished = false;
            if (it[i] > o.max_iterative_refinement) {        // loop exhausted: fail <=> the final error exceeds the initial one
                finished = true;
                if (!(norm[i] <= norm0[i])) {           // search_direction.jl:22 -> H \ residual on the member's own stream
                    h->stats.refine_fail += 1;
                    const int fr = nonsymmetric_solve(h, h->residual, h->step);
                    rc[i] = fr < 0 ? fr : std::max(rc[i], (int)CALIPSO_WARN_REFINEMENT);
                }
            } else if (norm[i] <= o.iterative_refinement_tolerance && it[i] >= o.min_iterative_refinement) finished = true;
            if (finished) {
                rounds[i] = it[i];
                h->stats.last_refine = it[i]; h->stats.refine_max = std::max<calipso::i64>(h->stats.refine_max, it[i]);
                if (it[i] == 0) none.push_back(i);
            } else sub.push_back(i);
        }
        // fill!(step_correction, 0) (iterative_refinement.jl:5) only for the members that take no round: the first round's k_recover writes every entry
        if (first_pass && !none.empty()) { g_activate(g, none); fill_d(s, s->step_correction, s->d.N, 0.0); }
        first_pass = false;
        if (sub.empty()) break;
        g_activate(g, sub);
        gb_refine_solve(g);
        gb_refine_residual(g);
        if (g_read_d(g, sub, 7, 1)) return CALIPSO_ERR_HIP;
        for (int i : sub) { norm[i] = g->hs[i]->hscal[7]; it[i] += 1; }
        run.swap(sub);
    }
    return CALIPSO_OK;
}

static Set alive(const Set& a, const std::vector<int>& rc) { Set r; for (int i : a) if (rc[i] >= 0) r.push_back(i); return r; }

static int gb_candidate_merit(G* g, const Set& a, std::vector<double>& Mh, std::vector<double>& thetah, bool queue_only = false, int extra = 0) {      // extra: dscal[6] travels along
    H* s = g->base;
    const int e = gb_evaluate(g, a, 1, CALIPSO_EVAL_OBJECTIVE | CALIPSO_EVAL_EQUALITY | CALIPSO_EVAL_CONE);
    if (e < 0) return e;
    launch_cone(s, s->candidate, CALIPSO_CONE_BARRIER | CALIPSO_CONE_BARRIER_GRADIENT);
    launch_merit(s, s->candidate);
    launch_constraint_violation(s, s->candidate);
    if (queue_only) return CALIPSO_OK;
    if (g_read_d(g, a, 4, 2 + extra)) return CALIPSO_ERR_HIP;
    for (int i : a) { Mh[i] = g->hs[i]->hscal[4]; thetah[i] = g->hs[i]->hscal[5]; }
    return CALIPSO_OK;
}

// the body of the inner loop of solve! (solve.jl:98-353) for every member of `a0`, device evaluator attached
static int gb_inner_iteration(G* g, const Set& a0, std::vector<IterInfo>& info, std::vector<int>& rc,
                              const std::vector<double>* eq_violation = nullptr, const std::vector<double>* cp_violation = nullptr, bool read_final = true) {
    H* s = g->base;
    const Dims& d = s->d;
    const size_t B = g->hs.size();
    g_activate(g, a0);
    EV(0);
    {
        const int e0 = gb_evaluate(g, a0, 0, CALIPSO_EVAL_OBJECTIVE_GRADIENT | CALIPSO_EVAL_EQUALITY_DUAL_GRADIENT | CALIPSO_EVAL_CONE_DUAL_GRADIENT);   // :100-104
        if (e0 < 0) return e0;
    }
    launch_cone(s, s->solution, CALIPSO_CONE_BARRIER | CALIPSO_CONE_BARRIER_GRADIENT);
    launch_merit_and_gradient(s);
    launch_residual(s);
    launch_violations_and_constraint(s);
    if (g_read_d(g, a0, 4, 14)) return CALIPSO_ERR_HIP;
    Set a;
    for (int i : a0) {
        H* h = g->hs[i]; const Options& o = h->opt; const double* hs = h->hscal; IterInfo& f = info[i];
        f.M = hs[4]; f.theta = hs[5];
        f.residual_violation = hs[8] / (double)d.N;
        const double sd = (d.ne + d.nc > 0) ? std::max(100.0, (hs[13] + hs[14]) / (double)(d.ne + d.nc)) / 100.0 : 1.0;
        const double scn = (d.nc > 0) ? std::max(100.0, hs[15] / (double)d.nc) / 100.0 : 1.0;
        f.optimality = std::max(std::max(hs[9] / sd, hs[10]), std::max(hs[11], hs[12] / scn));
        f.slack_violation = std::max(hs[10], hs[11]);
        if (eq_violation && f.residual_violation < o.residual_tolerance && f.slack_violation < o.slack_tolerance &&
            (*eq_violation)[i] <= o.equality_tolerance && (*cp_violation)[i] <= o.complementarity_tolerance) { f.exit_kind = 1; continue; }   // :138-143
        // (the benchmark step passes no violations: the outer-convergence exit cannot trigger there)
        if (f.optimality <= std::max(o.central_path_update_tolerance * h->sc.kappa, o.optimality_tolerance)) { f.exit_kind = 2; continue; }   // :165
        a.push_back(i);
    }
    EV(1);
    if (a.empty()) { EV(2); EV(3); EV(4); return CALIPSO_OK; }
    {   // :175-181 (a no-op for the device QP evaluator, whose Hessian / Jacobians are constant)
        Set cb;
        for (int i : a) if (!g->hs[i]->qp.attached) cb.push_back(i);     // (host callbacks and user device evaluators alike)
        if (!cb.empty()) {
            for (int i : cb) {
                uint32_t fl = CALIPSO_EVAL_OBJECTIVE_HESSIAN | CALIPSO_EVAL_EQUALITY_JACOBIAN | CALIPSO_EVAL_CONE_JACOBIAN;
                if (g->hs[i]->opt.constraint_tensor != 0.0) fl |= CALIPSO_EVAL_EQUALITY_DUAL_HESSIAN | CALIPSO_EVAL_CONE_DUAL_HESSIAN;
                const int e1 = gb_evaluate(g, Set{i}, 0, fl);
                if (e1 < 0) return e1;
            }
            g_activate(g, a);
        }
    }
    EV(2);
    std::vector<int64_t> nfact(B, 0);
    std::vector<int> rounds(B, 0);
    int e = gb_inertia_correction(g, a, rc, nfact);                                      // :187 search_direction!
    if (e < 0) return e;
    a = alive(a, rc);
    if (!a.empty()) {
        Set ref;
        for (int i : a) if (g->hs[i]->opt.iterative_refinement) ref.push_back(i);
        gb_sds(g, a, 0, false, ref.size() == a.size());      // every member refines: the local rows of the first residual come out of the solve's last launch
        if (!ref.empty()) { e = gb_refinement(g, ref, rc, rounds); if (e < 0) return e; }
    }
    for (int i : a) { info[i].nfact = nfact[i]; info[i].rounds = rounds[i]; }
    EV(3);
    if (a.empty()) { EV(4); return CALIPSO_OK; }
    // cone search (:190-221), first candidate (:206-229) and its merit / violation (:231-250).  When every member evaluates on the device, the three are queued back to
    // back — the step sizes of the first candidate formed from the search's masks ON THE DEVICE (vectors.hip: k_first_candidate_masks, the host's operations) — and ONE
    // synchronisation brings the masks and the three scalars of every member (were three synchronisations); the host forms the same step sizes from the same masks.
    std::vector<double> as(B, 1.0), at(B, 1.0), step_size(B, 1.0);
    std::vector<double> Mh(B, 0.0), thetah(B, 0.0), dd(B, 0.0);
    g_activate(g, a);
    bool all_device = d.nc > 0;
    for (int i : a) all_device = all_device && (g->hs[i]->qp.attached || g->hs[i]->dev_eval || g->hs[i]->dev_block_eval);
    auto step_sizes_from_masks = [&](const Set& set) {
        for (int i : set) {
            H* h = g->hs[i]; const Options& o = h->opt;
            const int ks = first_feasible_trial(h->hicount + 6, o.max_cone_line_search), kt = first_feasible_trial(h->hicount + 32, o.max_cone_line_search);
            if (ks < 0 || kt < 0) { h->err = "cone search failure"; rc[i] = CALIPSO_ERR_CONE_SEARCH; continue; }
            for (int k = 0; k < ks; ++k) as[i] = o.scaling_line_search * as[i];
            for (int k = 0; k < kt; ++k) at[i] = o.scaling_line_search * at[i];
        }
    };
    if (all_device) {
        launch_cone_search(s);
        launch_first_candidate_from_masks(s);
        e = gb_candidate_merit(g, a, Mh, thetah, true);
        if (e < 0) return e;
        if (g_read_both(g, a, 6, 58, 4, 3)) return CALIPSO_ERR_HIP;
        step_sizes_from_masks(a);
        for (int i : a) { Mh[i] = g->hs[i]->hscal[4]; thetah[i] = g->hs[i]->hscal[5]; dd[i] = g->hs[i]->hscal[6]; }
        a = alive(a, rc);
        if (a.empty()) { EV(4); return CALIPSO_OK; }
        g_activate(g, a);
        for (int i : a) { info[i].step_size = as[i]; info[i].step_size_t = at[i]; step_size[i] = as[i]; }
    } else {
        if (d.nc) {
            launch_cone_search(s);
            if (g_read_i(g, a, 6, 58)) return CALIPSO_ERR_HIP;
            step_sizes_from_masks(a);
            a = alive(a, rc);
            if (a.empty()) { EV(4); return CALIPSO_OK; }
            g_activate(g, a);
        }
        for (int i : a) { info[i].step_size = as[i]; info[i].step_size_t = at[i]; step_size[i] = as[i]; }
        {   // candidate s, t (:206-218), candidate x, r (:224-229) and the directional derivative of the merit function in one launch (api.hip: inner_iteration)
            std::vector<double> la, lt;
            for (int i : a) { la.push_back(as[i]); lt.push_back(at[i]); }
            launch_first_candidate_batch(s, la.data(), lt.data());
        }
        e = gb_candidate_merit(g, a, Mh, thetah, false, 1);                                // :231-250 (+ the directional derivative, dscal[6])
        if (e < 0) return e;
        for (int i : a) dd[i] = g->hs[i]->hscal[6];
    }
    // residual line search (:254-302): the members still back-tracking form the next launch's instance list
    std::vector<calipso::i64> ls_it(B, 0);
    Set run = a;
    while (!run.empty()) {
        Set sub;
        for (int i : run) {
            H* h = g->hs[i]; const Options& o = h->opt;
            const double M = info[i].M, theta = info[i].theta;
            if (!(ls_it[i] < o.max_residual_line_search)) continue;
            if (check_filter(h, thetah[i], Mh[i])) {
                if (theta <= o.slack_tolerance && switching_condition(step_size[i], dd[i], o.merit_exponent, theta, o.violation_exponent, 1.0) &&
                    armijo(M, Mh[i], dd[i], step_size[i], o.armijo_tolerance, o.machine_tolerance)) continue;
                else if (sufficient_progress(theta, thetah[i], M, Mh[i], o.violation_tolerance, o.merit_tolerance, o.machine_tolerance)) continue;
            }
            step_size[i] = o.scaling_line_search * step_size[i];
            sub.push_back(i);
        }
        if (sub.empty()) break;
        g_activate(g, sub);
        std::vector<double> la;
        for (int i : sub) la.push_back(step_size[i]);
        launch_axpy_points_batch(s, la.data(), 1);                                         // :268-276
        e = gb_candidate_merit(g, sub, Mh, thetah);                                      // :278-297
        if (e < 0) return e;
        for (int i : sub) ls_it[i] += 1;
        run.swap(sub);
    }
    for (int i : a) {
        H* h = g->hs[i]; const Options& o = h->opt;
        const double M = info[i].M, theta = info[i].theta;
        if (ls_it[i] >= o.max_residual_line_search) rc[i] = std::max(rc[i], (int)CALIPSO_WARN_LINE_SEARCH);
        if (!switching_condition(step_size[i], dd[i], o.merit_exponent, theta, o.violation_exponent, 1.0) ||
            !armijo(M, Mh[i], dd[i], step_size[i], o.armijo_tolerance, o.machine_tolerance))
            augment_filter(h, (1.0 - o.violation_tolerance) * theta, M - o.merit_tolerance * theta);   // filter.jl:81-89
        info[i].step_size = step_size[i]; info[i].Mh = Mh[i]; info[i].thetah = thetah[i];
        h->stats.newton_steps += 1;
    }
    g_activate(g, a);
    {
        std::vector<double> la;
        for (int i : a) la.push_back(step_size[i]);
        launch_accept_batch(s, la.data());                                                 // :309-326
    }
    launch_cone(s, s->solution, CALIPSO_CONE_PRODUCT);                                     // :328-330
    launch_violations(s);                                                                  // :332-333
    // (read_final = false: a benchmark step — nobody reads the two norms; the kernel computes them all the same and the caller's synchronisation is behind it)
    if (read_final && g_read_d(g, a, 16, 2)) return CALIPSO_ERR_HIP;
    EV(4);
    return CALIPSO_OK;
}

namespace calipso {
// Finalisers (Julia, Python) run in unspecified order: a member may be destroyed before its group.  The group then drops every
// handle pointer; its remaining entry points return CALIPSO_ERR_ARGUMENT and calipso_hip_group_destroy only frees its own memory.
void group_member_destroyed(calipso_hip_group* g, calipso_hip_solver* dying) {
    if (!g) return;
    for (H* h : g->hs) {
        if (h == g->base && h->stream) { (void)hipSetDevice(h->device); (void)hipStreamSynchronize(h->stream); }
        h->cur = nullptr;
        h->owner = nullptr;
    }
    (void)dying;
    g->hs.clear();
    g->base = nullptr;
    g->dead = true;
}
}  // namespace calipso

// members of a group share one cone-search launch: the options that launch takes by value must agree
static int g_check_options(G* g) {
    const Options& o = g->base->opt;
    for (H* h : g->hs)
        if (h->opt.scaling_line_search != o.scaling_line_search || h->opt.max_cone_line_search != o.max_cone_line_search) {
            g->base->err = "group members must share opt.scaling_line_search and opt.max_cone_line_search";
            return CALIPSO_ERR_ARGUMENT;
        }
    return CALIPSO_OK;
}

extern "C" {

int32_t calipso_hip_group_create(calipso_hip_solver** handles, int32_t count, calipso_hip_group** out) {
    if (!handles || !out || count < 1 || count > MAX_BATCH) return CALIPSO_ERR_ARGUMENT;
    H* b = handles[0];
    if (!b) return CALIPSO_ERR_ARGUMENT;
    for (int i = 0; i < count; ++i) {
        H* h = handles[i];
        if (!h) return CALIPSO_ERR_ARGUMENT;
        const bool same = h->device == b->device && h->slab_doubles == b->slab_doubles && std::memcmp(&h->d, &b->d, sizeof(Dims)) == 0 &&
                          h->h_soc_start == b->h_soc_start && h->h_soc_dim == b->h_soc_dim && h->h_nonneg == b->h_nonneg;
        if (!same) { b->err = "calipso_hip_group_create: members must have the same shape, cone layout and device"; return CALIPSO_ERR_ARGUMENT; }
        for (int j = 0; j < i; ++j) if (handles[j] == h) { b->err = "calipso_hip_group_create: duplicate member"; return CALIPSO_ERR_ARGUMENT; }
        if (h->owner) { b->err = "calipso_hip_group_create: a handle can be a member of one group at a time"; return CALIPSO_ERR_ARGUMENT; }
        if (h->compact != b->compact || (h->compact && h->blocks.signature != b->blocks.signature)) { b->err = "calipso_hip_group_create: structured members must share one structure"; return CALIPSO_ERR_ARGUMENT; }
    }
    if (b->compact && count > 1) {           // the multifrontal storage of the leader covers the whole group (a structured handle has no blocked path to fall back to)
        const int rc = calipso_hip_set_stage_parallel(b, 1, count, nullptr);
        if (rc < 0) return rc;
    }
    G* g = new G();
    g->hs.assign(handles, handles + count);
    g->base = b;
    for (H* h : g->hs) h->owner = g;
    H* s = b;
    *out = g;
    CK(hipSetDevice(b->device));
    CK(hipHostMalloc((void**)&g->hgather, sizeof(double) * 64 * MAX_BATCH, hipHostMallocMapped));
    CK(hipHostMalloc((void**)&g->higather, sizeof(int) * 64 * MAX_BATCH, hipHostMallocMapped));
    CK(hipHostGetDevicePointer((void**)&g->hgather_dev, g->hgather, 0));
    CK(hipHostGetDevicePointer((void**)&g->higather_dev, g->higather, 0));
    CK(hipHostMalloc((void**)&g->hseq, sizeof(unsigned long long), hipHostMallocMapped));
    g->hseq[0] = 0;
    CK(hipHostGetDevicePointer((void**)&g->hseq_dev, g->hseq, 0));
    CK(hipMalloc((void**)&g->ticket, sizeof(unsigned)));
    CK(hipMemset(g->ticket, 0, sizeof(unsigned)));
    CK(hipMalloc((void**)&g->sc_dev, sizeof(calipso::Scalars) * G::SC_RING * MAX_BATCH));
    CK(hipHostMalloc((void**)&g->sc_pin, sizeof(calipso::Scalars) * G::SC_RING * MAX_BATCH, hipHostMallocDefault));
    return CALIPSO_OK;
}

// host evaluation callbacks (calipso_eval_fn, include/calipso_hip.h) of the members that have no device evaluator; entries may be NULL
int32_t calipso_hip_group_set_evaluators(calipso_hip_group* g, const calipso_eval_fn* evals, void* const* users) {
    if (!g || g->dead) return CALIPSO_ERR_ARGUMENT;
    g->evals.assign(g->hs.size(), nullptr);
    g->users.assign(g->hs.size(), nullptr);
    for (size_t i = 0; i < g->hs.size(); ++i) { if (evals) g->evals[i] = evals[i]; if (users) g->users[i] = users[i]; }
    return CALIPSO_OK;
}

int32_t calipso_hip_group_destroy(calipso_hip_group* g) {
    if (!g) return CALIPSO_OK;
    if (g->base) { (void)hipSetDevice(g->base->device); (void)hipStreamSynchronize(g->base->stream); g->base->cur = nullptr; }
    for (H* h : g->hs) h->owner = nullptr;
    if (g->hgather) (void)hipHostFree(g->hgather);
    if (g->higather) (void)hipHostFree(g->higather);
    if (g->hseq) (void)hipHostFree(g->hseq);
    if (g->ticket) (void)hipFree(g->ticket);
    if (g->sc_dev) (void)hipFree(g->sc_dev);
    if (g->sc_pin) (void)hipHostFree(g->sc_pin);
    delete g;
    return CALIPSO_OK;
}

// calipso_hip_newton_step for every member at once; info = count x 6 (row-major, same fields), status = count entries
int32_t calipso_hip_group_newton_step(calipso_hip_group* g, int32_t advance, double* info_out, int32_t* status) {
    if (!g || !g->base || g->dead) return CALIPSO_ERR_ARGUMENT;
    H* s = g->base;
    const Dims& d = s->d;
    const size_t B = g->hs.size();
    CK(hipSetDevice(s->device));
    (void)hipGetLastError();      // (launch_errors: this call's launches only)
    { const int oc = g_check_options(g); if (oc < 0) return oc; }
    Set all;
    for (size_t i = 0; i < B; ++i) {
        H* h = g->hs[i];
        if (!h->qp.attached && !h->dev_eval && !h->dev_block_eval) { s->err = "calipso_hip_group_newton_step needs a device evaluator on every member (calipso_hip_qp_attach or calipso_hip_set_device_evaluator)"; return CALIPSO_ERR_ARGUMENT; }
        if (h != s) CK(hipStreamSynchronize(h->stream));   // uploads made through the member's own stream are complete
        all.push_back((int)i);
    }
    g_effective_band(g);
    { const int brc = g_effective_blocks(g); if (brc < 0) { g_restore_band(g); return brc; } }
    struct Finally { G* g; ~Finally() { g->base->cur = nullptr; g_restore_band(g); } } fin{g};
    {   // Lsym of the members whose Hessian changed
        Set dirty;
        for (int i : all) if (g->hs[i]->hessian_dirty) dirty.push_back(i);
        if (!dirty.empty()) { g_activate(g, dirty); launch_symmetrize(s); for (int i : dirty) g->hs[i]->hessian_dirty = false; }
    }
    std::vector<Scalars> saved_sc(B);
    std::vector<std::vector<double>> ft(B), fm(B);
    std::vector<calipso::i64> fi(B, 0);
    g_activate(g, all);
    if (!advance) {
        {   // (one launch, as for a single handle: api.hip)
            double* const dst[4] = {s->saved_point, s->saved_g, s->saved_h, s->dscal + 32};
            const double* const src[4] = {s->solution, s->g, s->hc, s->dscal};
            const size_t n[4] = {(size_t)d.N, (size_t)d.ne, (size_t)d.nc, 2};
            copy4_d(s, dst, src, n);
        }
        for (int i : all) { H* h = g->hs[i]; saved_sc[i] = h->sc; ft[i] = h->filter_theta; fm[i] = h->filter_merit; fi[i] = h->filter_index; }
    }
    std::vector<IterInfo> info(B);
    std::vector<int> rc(B, 0);
    EV(8);
    int e = gb_inner_iteration(g, all, info, rc, nullptr, nullptr, false);
    EV(9);
    if (e < 0) return e;
    if (!advance) {
        g_activate(g, all);
        {
            double* const dst[4] = {s->solution, s->g, s->hc, s->dscal};
            const double* const src[4] = {s->saved_point, s->saved_g, s->saved_h, s->dscal + 32};
            const size_t n[4] = {(size_t)d.N, (size_t)d.ne, (size_t)d.nc, 2};
            copy4_d(s, dst, src, n);
        }
        launch_cone(s, s->solution, CALIPSO_CONE_PRODUCT);
        for (int i : all) {
            H* h = g->hs[i];
            h->filter_theta = ft[i]; h->filter_merit = fm[i]; h->filter_index = fi[i];
            const double keep_ep = h->sc.ep, keep_ed = h->sc.ed;
            h->sc = saved_sc[i]; h->sc.ep = keep_ep; h->sc.ed = keep_ed;
        }
    }
    SYNC();
    for (size_t i = 0; i < B; ++i) {
        if (info_out) {
            double* o = info_out + 6 * i;
            o[0] = info[i].step_size; o[1] = info[i].step_size_t; o[2] = info[i].rounds; o[3] = (double)info[i].nfact; o[4] = info[i].Mh; o[5] = info[i].thetah;
        }
        if (status) status[i] = rc[i];
    }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, s->ev[0], s->ev[1]); s->phase_ms[0] = ms;
    (void)hipEventElapsedTime(&ms, s->ev[2], s->ev[3]); s->phase_ms[2] = ms;
    (void)hipEventElapsedTime(&ms, s->ev[3], s->ev[4]); s->phase_ms[5] = ms;
    (void)hipEventElapsedTime(&ms, s->ev[8], s->ev[9]); s->phase_ms[6] = ms;
    return CALIPSO_OK;
}

// solve!(solver) (solve.jl:8-377) for every member in lockstep (device evaluators attached): one pass of the inner loop body per
// round for all members still iterating; members whose inner loop ends (central-path update due, or iteration cap) take their
// outer update before the next round, converged members drop out.  result[i] = 1 converged, 0 iteration caps reached, < 0 error.
int32_t calipso_hip_group_solve(calipso_hip_group* g, int32_t* result) {
    if (!g || !g->base || g->dead) return CALIPSO_ERR_ARGUMENT;
    H* s = g->base;
    const Dims& d = s->d;
    const size_t B = g->hs.size();
    CK(hipSetDevice(s->device));
    (void)hipGetLastError();      // (launch_errors: this call's launches only)
    { const int oc = g_check_options(g); if (oc < 0) return oc; }
    Set all;
    for (size_t i = 0; i < B; ++i) {
        H* h = g->hs[i];
        if (!h->qp.attached && !h->dev_eval && !h->dev_block_eval && (i >= g->evals.size() || !g->evals[i])) {
            s->err = "calipso_hip_group_solve: member without a device evaluator and without a callback (calipso_hip_group_set_evaluators)";
            return CALIPSO_ERR_ARGUMENT;
        }
        if (h != s) CK(hipStreamSynchronize(h->stream));
        all.push_back((int)i);
    }
    g_effective_band(g);
    { const int brc = g_effective_blocks(g); if (brc < 0) { g_restore_band(g); return brc; } }
    struct Finally { G* g; ~Finally() { g->base->cur = nullptr; g_restore_band(g); } } fin{g};
    {
        Set dirty;
        for (int i : all) if (g->hs[i]->hessian_dirty) dirty.push_back(i);
        if (!dirty.empty()) { g_activate(g, dirty); launch_symmetrize(s); for (int i : dirty) g->hs[i]->hessian_dirty = false; }
    }
    const uint32_t eval0 = CALIPSO_EVAL_EQUALITY | CALIPSO_EVAL_CONE;
    Set cold;
    for (int i : all) { g->hs[i]->stats = Stats(); if (g->hs[i]->opt.warmstart == 0.0) cold.push_back(i); }
    if (!cold.empty()) {                                                                   // initialize_slacks!/duals! initialize.jl:15-36
        const int e0 = gb_evaluate(g, cold, 0, eval0);
        if (e0 < 0) return e0;
        launch_init_point(s);
    }
    for (int i : all) {
        H* h = g->hs[i]; Options& o = h->opt; Scalars& sc = h->sc;
        sc.kappa = o.central_path_initial; sc.tau = std::max(0.99, 1.0 - sc.kappa);       // initialize.jl:38-42
        sc.rho = o.penalty_initial;                                                       // :44-48
        g_activate(g, Set{i});
        fill_d(s, s->lambda, d.ne, o.dual_initial);
        filter_reset(h);                                                                  // solve.jl:95
    }
    {
        const int e1 = gb_evaluate(g, all, 0, CALIPSO_EVAL_OBJECTIVE | CALIPSO_EVAL_EQUALITY | CALIPSO_EVAL_EQUALITY_JACOBIAN | CALIPSO_EVAL_CONE);   // :78-83
        if (e1 < 0) return e1;
    }
    launch_violations(s);
    if (g_read_d(g, all, 16, 2)) return CALIPSO_ERR_HIP;
    std::vector<double> ev(B), cv(B);
    for (int i : all) { ev[i] = g->hs[i]->hscal[16]; cv[i] = g->hs[i]->hscal[17]; }        // :85-86 (cone product read before cone!: reference quirk)
    launch_cone(s, s->solution, CALIPSO_CONE_PRODUCT | CALIPSO_CONE_TARGET);               // :88-91
    std::vector<calipso::i64> outer(B, 1), inner(B, 1), total(B, 1);
    std::vector<int> res(B, 0), worst(B, 0);
    for (int i : all) g->hs[i]->stats.outer = 1;
    Set active = all;
    while (!active.empty()) {
        std::vector<IterInfo> info(B);
        std::vector<int> rc(B, 0);
        const int e = gb_inner_iteration(g, active, info, rc, &ev, &cv);
        if (e < 0) return e;
        Set next, upd;
        for (int i : active) {
            H* h = g->hs[i]; const Options& o = h->opt;
            if (rc[i] < 0) { res[i] = rc[i]; continue; }
            worst[i] = std::max(worst[i], rc[i]);
            if (info[i].exit_kind == 1) {                                                                 // converged  :138-160
                h->stats.total_iterations = total[i]; res[i] = 1;
                if (o.differentiate != 0.0 && d.np > 0) {                                                 // differentiate! on the member's own stream
                    SYNC();
                    s->cur = nullptr;                                                                     // a single-handle call (h may be the base)
                    const int dr = calipso_hip_differentiate(h, i < (int)g->evals.size() ? g->evals[i] : nullptr, i < (int)g->users.size() ? g->users[i] : nullptr);
                    if (dr < 0) res[i] = dr;
                }
                continue;
            }
            bool inner_done = info[i].exit_kind == 2;                                                     // :165
            if (!inner_done) {
                ev[i] = h->hscal[16]; cv[i] = h->hscal[17];                                               // :332-333
                if (h->cb_inner) { SYNC(); h->cb_inner(h->cb_user, h); }
                total[i] += 1; h->stats.total_iterations = total[i];
                inner[i] += 1;
                if (inner[i] > o.max_residual_iterations) inner_done = true;
            }
            if (inner_done) upd.push_back(i); else next.push_back(i);
        }
        if (!upd.empty()) {                                                                // outer updates  :356-371
            for (int i : upd) {
                H* h = g->hs[i]; const Options& o = h->opt; Scalars& sc = h->sc;
                sc.kappa = std::max(o.residual_tolerance / 10.0, std::min(o.central_path_scaling * sc.kappa, std::pow(sc.kappa, o.central_path_exponent)));
                sc.tau = std::max(0.99, 1.0 - sc.kappa);
            }
            g_activate(g, upd);                                                            // lambda += rho r with the OLD rho (:362-365)
            launch_lambda_update(s);
            for (int i : upd) {
                H* h = g->hs[i]; const Options& o = h->opt; Scalars& sc = h->sc;
                sc.rho = std::min(std::max(o.penalty_scaling * sc.rho, 1.0 / sc.kappa), o.max_penalty);
                filter_reset(h);
                if (h->cb_outer) { SYNC(); h->cb_outer(h->cb_user, h); }
                outer[i] += 1; inner[i] = 1;
                if (outer[i] > o.max_outer_iterations) { h->stats.total_iterations = total[i]; res[i] = 0; continue; }
                h->stats.outer = outer[i];
                next.push_back(i);
            }
            std::sort(next.begin(), next.end());
        }
        active.swap(next);
    }
    SYNC();
    if (result) for (size_t i = 0; i < B; ++i) result[i] = res[i];
    return CALIPSO_OK;
}

}  // extern "C"
