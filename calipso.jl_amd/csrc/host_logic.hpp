// host_logic.hpp — host-side pieces of the inner iteration shared by the single-handle driver (api.hip) and the group driver
// (group.hip): inertia test, filter, line-search predicates.  Pure C++ on the handle's host state; nothing here touches the device.
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <thread>

#include "internal.hpp"

typedef calipso_hip_solver H;
#define SYNC() CK(hipStreamSynchronize(s->stream))

// A kernel launch that the runtime refuses (a launch configuration the device cannot serve: LDS, registers, grid) reports through hipGetLastError only — the launch macros
// return nothing — and everything queued behind it would run on stale data.  Every host wait of a Newton step (api.hip: wait_published / publish_and_wait, group.hip:
// g_read_*) asks for it first: a refused launch of the phase just queued surfaces as CALIPSO_ERR_HIP at the phase's own read-back, not as a wrong inertia or a failed
// line search later.  (hipErrorNotReady is what the liveness polls of host_wait leave behind: not an error.)
static inline int launch_errors(H* s, const char* where) {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess || e == hipErrorNotReady) return 0;
    return calipso::check(s, e, where);
}

// One place for the host's waits on words the device publishes (sequence numbers of read-backs, the progress word of the panel launches): a short pure spin — the
// read-back of ONE handle is 5-10 us away and a yield costs more than that — then yields, then short sleeps: 24 handles waiting (3 lanes x 8 ranks) do not hold 24 cores.
// `ready` is polled; `alive` is asked now and then (false: the stream faulted or ended without publishing — stop waiting).  Returns `ready()`.
template <typename Ready, typename Alive> static inline bool host_wait(Ready ready, Alive alive) {
    for (unsigned spins = 0;; ++spins) {
        if (ready()) return true;
        if (spins < (1u << 15)) continue;                                     // ~30-60 us of pure spinning
        if ((spins & 0x3ffu) == 0 && !alive()) return ready();
        if (spins < (1u << 19)) std::this_thread::yield();
        else std::this_thread::sleep_for(std::chrono::microseconds(20));
    }
}

static inline bool inertia_ok(const H* s, const int64_t in[3]) { return in[0] == s->d.nx && in[1] == s->d.ne + s->d.nc && in[2] == 0; }   // inertia.jl:7-11


// first trial index k in 0..max_cone_line_search whose violation bit is clear (cones.hip: violation_masks), -1 if none: the
// reference raises "cone search failure" once cone_iteration exceeds max_cone_line_search (solve.jl:204-221)
static inline int first_feasible_trial(const int* mask, calipso::i64 max_cone_line_search) {
    const int nk = (int)std::min<calipso::i64>(max_cone_line_search + 1, calipso::CONE_MASK_TRIALS);
    for (int k = 0; k < nk; ++k) if (!(mask[k >> 5] & (1 << (k & 31)))) return k;
    return -1;
}

// ---- filter (filter.jl:1-89), host side ------------------------------------------------------------------------------------
static inline void filter_reset(H* s) {
    for (calipso::i64 i = 0; i < s->filter_index; ++i) { s->cache_theta[i] = 1.0e8; s->cache_merit[i] = 1.0e8; }
    for (calipso::i64 i = 0; i < s->filter_index; ++i) { s->filter_theta[i] = 1.0e8; s->filter_merit[i] = 1.0e8; }
    s->filter_index = 0;
}
static inline bool check_filter(const H* s, double theta, double merit) {
    for (size_t i = 0; i < s->filter_theta.size(); ++i)
        if (!(theta < s->filter_theta[i] || merit < s->filter_merit[i])) return false;
    return true;
}
static inline void filter_resize(H* s, calipso::i64 n) {
    if (n < 1) n = 1;
    s->filter_theta.resize((size_t)n, 1.0e8); s->filter_merit.resize((size_t)n, 1.0e8);
    s->cache_theta.resize((size_t)n, 1.0e8); s->cache_merit.resize((size_t)n, 1.0e8);
    if (s->filter_index > n) s->filter_index = n;
}
// returns false when the filter is full (the reference raises a BoundsError there: filter.jl:52-79 has no overflow guard)
static inline bool augment_filter(H* s, double theta, double merit) {
    if (s->filter_index >= (calipso::i64)s->filter_theta.size() && check_filter(s, theta, merit)) {
        // every kept pair plus the new one must fit: grow instead of writing past the end
        filter_resize(s, 2 * (calipso::i64)s->filter_theta.size());
    }
    if (s->filter_index == 0) { s->filter_theta[0] = theta; s->filter_merit[0] = merit; s->filter_index = 1; return true; }
    if (check_filter(s, theta, merit)) {
        const calipso::i64 nold = s->filter_index;
        for (calipso::i64 i = 0; i < nold; ++i) { s->cache_theta[i] = s->filter_theta[i]; s->cache_merit[i] = s->filter_merit[i]; }
        for (calipso::i64 i = 0; i < nold; ++i) { s->filter_theta[i] = 1.0e8; s->filter_merit[i] = 1.0e8; }
        s->filter_index = 0;
        s->filter_theta[0] = theta; s->filter_merit[0] = merit; s->filter_index = 1;
        for (calipso::i64 i = 0; i < nold; ++i)
            if (!(s->cache_theta[i] >= theta && s->cache_merit[i] >= merit)) {
                s->filter_theta[s->filter_index] = s->cache_theta[i];
                s->filter_merit[s->filter_index] = s->cache_merit[i];
                s->filter_index += 1;
            }
    }
    return true;
}
// line_search.jl:2-18 with d = dot(merit_gradient, step.primals) precomputed on the device
static inline bool switching_condition(double step_size, double dd, double merit_exponent, double violation, double violation_exponent, double reg) {
    return dd < 0.0 && step_size * std::pow(-dd, merit_exponent) > reg * std::pow(violation, violation_exponent);
}
static inline bool sufficient_progress(double v, double vc, double m, double mc, double vt, double mt, double mach) {
    return vc - 10.0 * mach * std::fabs(v) <= (1.0 - vt) * v || mc - 10.0 * mach * std::fabs(m) <= m - mt * v;
}
static inline bool armijo(double m, double mc, double dd, double step_size, double at, double mach) {
    return mc - m - 10.0 * mach * std::fabs(m) <= at * step_size * dd;
}


struct IterInfo {
    double step_size = 1.0, step_size_t = 1.0, M = 0.0, Mh = 0.0, theta = 0.0, thetah = 0.0;
    int rounds = 0;
    int64_t nfact = 0;
    int exit_kind = 0;   // 0 stepped, 1 outer convergence, 2 inner convergence
    double residual_violation = 0, optimality = 0, slack_violation = 0;
};

#define EV(i) (void)hipEventRecord(s->ev[i], s->stream)

