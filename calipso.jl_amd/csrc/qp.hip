// qp.hip — device-resident evaluator for the conic QP  min c x'Px + q'x  s.t. Ax-b = 0, h-Gx in K  (SURVEY.md 8(d)),
// i.e. the user-evaluation part of evaluate! (src/solver/evaluate.jl:37-121) for the synthetic benchmark problems, done as
// HBM-bound mat-vecs so a Newton step never leaves the GPU.  With Lxx = 2cP, gx = A, hx = -G held by the handle:
//   f = 1/2 x'Lxx x + q'x      fx = Lxx x + q      g = gx x - b      (g'y)x = gx'y      h = hvec + hx x      (h'z)x = hx'z
#include "internal.hpp"
#include "device_utils.hpp"

namespace calipso {

__global__ __launch_bounds__(1024) void k_qp_objective(Batch bt, int nx, const double* __restrict__ x, const double* __restrict__ Lx,
                                                        const double* __restrict__ q, double* __restrict__ dscal) {
    __shared__ double sm[16];
    inst_shift(bt, x, Lx, q, dscal);
    double a = 0.0, b = 0.0;
    for (int i0 = threadIdx.x; i0 < nx; i0 += 4 * 1024) {           // four entries of a thread in flight together, same order of its sums
        double xv[4], lv[4], qv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + u * 1024; const bool in = i < nx; xv[u] = in ? x[i] : 0.0; lv[u] = in ? Lx[i] : 0.0; qv[u] = in ? q[i] : 0.0; }
#pragma unroll
        for (int u = 0; u < 4; ++u) if (i0 + u * 1024 < nx) { a += xv[u] * lv[u]; b += qv[u] * xv[u]; }
    }
    const double ra = block_sum(a, sm);
    const double rb = block_sum(b, sm);
    if (threadIdx.x == 0) dscal[0] = 0.5 * ra + rb;
}

__global__ void k_vec_add(Batch bt, int n, const double* __restrict__ a, const double* __restrict__ b, double sign, double* __restrict__ out) {
    inst_shift(bt, a, b, out);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] + sign * b[i];
}

void launch_qp_evaluate(calipso_hip_solver* s, const double* point, uint32_t flags) {
    const Dims& d = s->d;
    const double* x = point;
    const double* y = point + d.oy();
    const double* z = point + d.oz();
    double* Lx = s->vtmp + 3 * (size_t)d.N;   // scratch of length >= nx
    const Batch bt = batch_of(s).b;
    const unsigned nz = bt.n;
    if (flags & (CALIPSO_EVAL_OBJECTIVE | CALIPSO_EVAL_OBJECTIVE_GRADIENT)) {
        // (the vector a product is shifted by rides in the product's own epilogue — a + 1.0 * b either way: the same bits as the separate k_vec_add launches)
        if (!(flags & CALIPSO_EVAL_OBJECTIVE)) gemv_t(s, d.nx, d.nx, s->Lxx, d.nx, x, s->fx, 1.0, 0.0, SP_LXX, s->qp.q);   // gradient only: fx = Lxx x + q; Lxx is symmetric for the QP
        else {
            gemv_t(s, d.nx, d.nx, s->Lxx, d.nx, x, Lx, 1.0, 0.0, SP_LXX);
            hipLaunchKernelGGL(k_qp_objective, dim3(1, 1, nz), dim3(1024), 0, s->stream, bt, d.nx, x, Lx, s->qp.q, s->dscal);
            if (flags & CALIPSO_EVAL_OBJECTIVE_GRADIENT)
                hipLaunchKernelGGL(k_vec_add, dim3((d.nx + 255) / 256, 1, nz), dim3(256), 0, s->stream, bt, d.nx, Lx, s->qp.q, 1.0, s->fx);
        }
    }
    const bool want_g = (flags & CALIPSO_EVAL_EQUALITY) && d.ne, want_h = (flags & CALIPSO_EVAL_CONE) && d.nc;
    if (want_g && want_h) {            // [g; h] = [gx; hx] x + [-b; hvec]  — one pass over the stacked Jacobian
        gemv_n(s, d.m, d.nx, s->Z, d.m, x, s->gh, 1.0, 0.0, SP_Z, s->qp.bh);
    } else if (want_g) {
        gemv_n(s, d.ne, d.nx, s->gx, d.m, x, s->g, 1.0, 0.0, SP_GX, s->qp.bh);
    } else if (want_h) {
        gemv_n(s, d.nc, d.nx, s->hx, d.m, x, s->hc, 1.0, 0.0, SP_HX, s->qp.bh + d.ne);
    }
    if (flags & CALIPSO_EVAL_EQUALITY_DUAL_GRADIENT) {
        if (d.ne) gemv_t(s, d.ne, d.nx, s->gx, d.m, y, s->gyx, 1.0, 0.0, SP_GX);
        else fill_d(s, s->gyx, d.nx, 0.0);
    }
    if (flags & CALIPSO_EVAL_CONE_DUAL_GRADIENT) {
        if (d.nc) gemv_t(s, d.nc, d.nx, s->hx, d.m, z, s->hzx, 1.0, 0.0, SP_HX);
        else fill_d(s, s->hzx, d.nx, 0.0);
    }
    // Hessian / Jacobians are constant for a QP and were installed by calipso_hip_qp_attach
}

}  // namespace calipso
