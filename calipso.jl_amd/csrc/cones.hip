// cones.hip — cone algebra kernels (reference: src/solver/cones/{cone,nonnegative,second_order}.jl).
// Cone vectors are a few thousand entries: these kernels are latency-bound, so each is ONE workgroup of 1024
// threads with deterministic in-workgroup reductions (no atomics, bit-reproducible run to run).
//   lanes stride over the nonnegative entries; second-order cones are handled one cone per lane for dimension <= 8
//   and one cone per wavefront (lanes stride over the cone's entries, DPP/shuffle reductions) above that.
#include "internal.hpp"
#include "device_utils.hpp"

namespace calipso {

constexpr int CONE_THREADS = 1024;
constexpr int SOC_WAVE_DIM = 8;   // cones larger than this use a whole wavefront

// cone!(...; barrier, barrier_gradient, product, target)  cones/cone.jl:71-106
//   barrier          Phi = sum log s_i + sum 1/2 log(s1^2 - |s2:|^2)           nonnegative.jl:11, second_order.jl:13
//   barrier_gradient 1/s_i ; [s1; -s2:]/(s1^2 - |s2:|^2)                       nonnegative.jl:12, second_order.jl:14
//   product          s_i t_i ; [s't; s1 t2: + t1 s2:]                          nonnegative.jl:15, second_order.jl:17
//   target           1 ; [1; 0...]                                             nonnegative.jl:26, second_order.jl:42
__global__ __launch_bounds__(CONE_THREADS) void k_cone(Batch bt, Dims d, ConeDev cd, const double* __restrict__ point, int flags,
                                                        double* __restrict__ product, double* __restrict__ target,
                                                        double* __restrict__ bgrad, double* __restrict__ dscal) {
    __shared__ double sm[CONE_THREADS / 64];
    inst_shift(bt, point, product, target, bgrad, dscal);
    const double* s = point + d.os();
    const double* t = point + d.ot();
    const int tid = threadIdx.x;
    double phi = 0.0;
    for (int i = tid; i < d.q; i += CONE_THREADS) {
        const double si = s[i];
        if (flags & CALIPSO_CONE_BARRIER) phi += log(si);
        if (flags & CALIPSO_CONE_BARRIER_GRADIENT) bgrad[i] = 1.0 / si;
        if (flags & CALIPSO_CONE_PRODUCT) product[i] = si * t[i];
        if (flags & CALIPSO_CONE_TARGET) target[i] = 1.0;
    }
    // small cones: one per lane
    for (int j = tid; j < d.n_soc; j += CONE_THREADS) {
        const int st = cd.soc_start[j], dim = cd.soc_dim[j];
        if (dim > SOC_WAVE_DIM) continue;
        double ss = 0.0, dot = 0.0;
        for (int k = 1; k < dim; ++k) ss += s[st + k] * s[st + k];
        for (int k = 0; k < dim; ++k) dot += s[st + k] * t[st + k];
        const double det = s[st] * s[st] - ss;
        if (flags & CALIPSO_CONE_BARRIER) phi += 0.5 * log(det);
        if (flags & CALIPSO_CONE_BARRIER_GRADIENT) {
            const double scale = 1.0 / det;
            bgrad[st] = scale * s[st];
            for (int k = 1; k < dim; ++k) bgrad[st + k] = scale * (-s[st + k]);
        }
        if (flags & CALIPSO_CONE_PRODUCT) {
            product[st] = dot;
            for (int k = 1; k < dim; ++k) product[st + k] = s[st] * t[st + k] + t[st] * s[st + k];
        }
        if (flags & CALIPSO_CONE_TARGET) {
            target[st] = 1.0;
            for (int k = 1; k < dim; ++k) target[st + k] = 0.0;
        }
    }
    // large cones: one per wavefront, shuffle reductions
    const int lane = tid & 63, wave = tid >> 6, nwave = CONE_THREADS / 64;
    for (int j = wave; j < d.n_soc; j += nwave) {
        const int st = cd.soc_start[j], dim = cd.soc_dim[j];
        if (dim <= SOC_WAVE_DIM) continue;
        double ss = 0.0, dot = 0.0;
        for (int k = lane; k < dim; k += 64) {
            const double sk = s[st + k];
            if (k > 0) ss += sk * sk;
            dot += sk * t[st + k];
        }
        ss = __shfl(wave_sum(ss), 0, 64);
        dot = __shfl(wave_sum(dot), 0, 64);
        const double s1 = s[st], t1 = t[st];
        const double det = s1 * s1 - ss;
        if ((flags & CALIPSO_CONE_BARRIER) && lane == 0) phi += 0.5 * log(det);
        for (int k = lane; k < dim; k += 64) {
            if (flags & CALIPSO_CONE_BARRIER_GRADIENT) bgrad[st + k] = (1.0 / det) * (k == 0 ? s1 : -s[st + k]);
            if (flags & CALIPSO_CONE_PRODUCT) product[st + k] = (k == 0) ? dot : s1 * t[st + k] + t1 * s[st + k];
            if (flags & CALIPSO_CONE_TARGET) target[st + k] = (k == 0) ? 1.0 : 0.0;
        }
    }
    if (flags & CALIPSO_CONE_BARRIER) {
        const double tot = block_sum(phi, sm);
        if (tid == 0) dscal[1] = tot;
    }
}

void launch_cone(calipso_hip_solver* s, const double* point, int flags) {
    if (s->d.nc == 0) {
        if (flags & CALIPSO_CONE_BARRIER) fill_d(s, s->dscal + 1, 1, 0.0);
        return;
    }
    const BatchSc B = batch_of(s);
    hipLaunchKernelGGL(k_cone, dim3(1, 1, B.b.n), dim3(CONE_THREADS), 0, s->stream, B.b, s->d, s->cone, point, flags, s->cone_product,
                       s->cone_target, s->barrier_gradient, s->dscal);
}

// cone_violation(xhat, x, tau)  cones/cone.jl:62-68, nonnegative.jl:29-34, second_order.jl:45-47 evaluated for all the
// candidate step sizes alpha_k = scaling_line_search^k (k = 0 .. max_cone_line_search) at once:  xhat = x - alpha_k * dx.
// Bit k & 31 of mask[k >> 5] is set  <=>  violation at alpha_k (CONE_MASK_WORDS words: up to 832 trial step sizes).
// The sequential shrinking of solve.jl:204-221 stops at the first k without violation, which is what the host picks.
// alpha_k is formed exactly as the reference does: alpha <- scaling_line_search * alpha, starting from 1.
__device__ __forceinline__ void violation_masks(const Dims& d, const ConeDev& cd, const double* __restrict__ x,
                                                const double* __restrict__ dx, double tau, double sls, int nk, int* __restrict__ mask) {
    const int tid = threadIdx.x;
    const double omt = 1.0 - tau;
    for (int i = tid; i < d.q; i += blockDim.x) {
        const double xi = x[i], dxi = dx[i];
        double a = 1.0;
        for (int k = 0; k < nk; ++k, a = sls * a) {
            if (xi - a * dxi <= omt * xi) atomicOr(&mask[k >> 5], 1 << (k & 31));
        }
    }
    for (int j = tid; j < d.n_soc; j += blockDim.x) {
        const int st = cd.soc_start[j], dim = cd.soc_dim[j];
        double a = 1.0;
        for (int k = 0; k < nk; ++k, a = sls * a) {
            double nrm = 0.0;
            for (int e = 1; e < dim; ++e) {
                const double df = (x[st + e] - a * dx[st + e]) - omt * x[st + e];
                nrm += df * df;
            }
            if ((x[st] - a * dx[st]) - omt * x[st] <= sqrt(nrm)) atomicOr(&mask[k >> 5], 1 << (k & 31));
        }
    }
}

__global__ __launch_bounds__(CONE_THREADS) void k_cone_search(BatchSc bt, Dims d, ConeDev cd, const double* __restrict__ sol,
                                                               const double* __restrict__ step, double sls, int nk,
                                                               int* __restrict__ icount, int* __restrict__ hdst = nullptr, unsigned long long* __restrict__ hseq = nullptr,
                                                               unsigned long long seq = 0, unsigned* __restrict__ ticket = nullptr) {
    inst_shift(bt.b, sol, step);
    inst_shift_i(bt.b, icount);
    const double tau = bt.scal(blockIdx.z).tau;
    // block 0: slack s with Delta s ; block 1: slack dual t with Delta t   (separate step sizes, solve.jl:190-221).  Each block owns its words of
    // icount (6 .. 31 / 32 .. 63): the masks are gathered in LDS and stored whole, so nothing has to clear them beforehand
    __shared__ int lm[32];
    if (threadIdx.x < 32) lm[threadIdx.x] = 0;
    __syncthreads();
    const int off = blockIdx.x == 0 ? d.os() : d.ot();
    violation_masks(d, cd, sol + off, step + off, tau, sls, nk, lm);
    __syncthreads();
    const int first = blockIdx.x == 0 ? 6 : 32, words = blockIdx.x == 0 ? 26 : 32;
    if ((int)threadIdx.x < words) icount[first + threadIdx.x] = lm[threadIdx.x];
    // a single handle: the masks go straight to the mapped host words as well, and the block that finishes second stores the sequence number the host spins on
    // (api.hip: wait_published) — what a k_publish_words launch behind this kernel did
    if (hdst) {
        if ((int)threadIdx.x < words) hdst[first + threadIdx.x] = lm[threadIdx.x];
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {
            if (atomicAdd(ticket, 1u) == 1u) {
                __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __threadfence_system();
                __hip_atomic_store(hseq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

void launch_cone_search(calipso_hip_solver* s, unsigned long long publish_seq) {
    if (s->d.nc == 0) { fill_i(s, s->icount + 6, 58, 0); return; }
    const int nk = (int)s->opt.max_cone_line_search + 1;
    const BatchSc B = batch_of(s);
    const bool pub = publish_seq != 0 && !s->cur;
    hipLaunchKernelGGL(k_cone_search, dim3(2, 1, B.b.n), dim3(CONE_THREADS), 0, s->stream, B, s->d, s->cone, s->solution, s->step,
                       s->opt.scaling_line_search, nk > CONE_MASK_TRIALS ? CONE_MASK_TRIALS : nk, s->icount, pub ? s->hicount_dev : (int*)nullptr,
                       pub ? s->hseq_dev : (unsigned long long*)nullptr, publish_seq, reinterpret_cast<unsigned*>(s->dscal + 62));
}

// candidate s, t for the chosen step sizes (solve.jl:206-208, 216-218)
struct StepSizes { double a_s[MAX_BATCH], a_t[MAX_BATCH]; };
__global__ void k_cone_candidate(Batch bt, Dims d, const double* __restrict__ sol, const double* __restrict__ step, double* __restrict__ cand,
                                 StepSizes a) {
    inst_shift(bt, sol, step, cand);
    const double a_s = a.a_s[blockIdx.z], a_t = a.a_t[blockIdx.z];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < d.nc) {
        cand[d.os() + i] = sol[d.os() + i] - a_s * step[d.os() + i];
        cand[d.ot() + i] = sol[d.ot() + i] - a_t * step[d.ot() + i];
    }
}

void launch_cone_candidate_batch(calipso_hip_solver* s, const double* a_s, const double* a_t) {   // one (a_s, a_t) per covered instance
    if (s->d.nc == 0) return;
    const BatchSc B = batch_of(s);
    StepSizes a;
    for (int k = 0; k < B.b.n; ++k) { a.a_s[k] = a_s[k]; a.a_t[k] = a_t[k]; }
    hipLaunchKernelGGL(k_cone_candidate, dim3((s->d.nc + 255) / 256, 1, B.b.n), dim3(256), 0, s->stream, B.b, s->d, s->solution, s->step,
                       s->candidate, a);
}
void launch_cone_candidate(calipso_hip_solver* s, double a_s, double a_t) { launch_cone_candidate_batch(s, &a_s, &a_t); }

// plain cone_violation(xhat, x, tau) on two device vectors of length nc: icount[6] != 0 <=> violation
__global__ __launch_bounds__(CONE_THREADS) void k_cone_violation(Batch bt, Dims d, ConeDev cd, const double* __restrict__ xhat,
                                                                  const double* __restrict__ x, double tau, int* __restrict__ icount) {
    inst_shift(bt, xhat, x);
    inst_shift_i(bt, icount);
    const int tid = threadIdx.x;
    const double omt = 1.0 - tau;
    for (int i = tid; i < d.q; i += blockDim.x)
        if (xhat[i] <= omt * x[i]) atomicOr(&icount[6], 1);
    for (int j = tid; j < d.n_soc; j += blockDim.x) {
        const int st = cd.soc_start[j], dim = cd.soc_dim[j];
        double nrm = 0.0;
        for (int e = 1; e < dim; ++e) {
            const double df = xhat[st + e] - omt * x[st + e];
            nrm += df * df;
        }
        if (xhat[st] - omt * x[st] <= sqrt(nrm)) atomicOr(&icount[6], 1);
    }
}

void launch_cone_violation_host(calipso_hip_solver* s, const double* xhat_dev, const double* x_dev, double tau) {
    fill_i(s, s->icount + 6, 1, 0);
    if (s->d.nc == 0) return;
    const BatchSc B = batch_of(s);
    hipLaunchKernelGGL(k_cone_violation, dim3(1, 1, B.b.n), dim3(CONE_THREADS), 0, s->stream, B.b, s->d, s->cone, xhat_dev, x_dev, tau, s->icount);
}

}  // namespace calipso
