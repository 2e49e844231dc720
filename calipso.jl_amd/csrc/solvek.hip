// solvek.hip — the condensed solve  K [dx; dy; dz] = b  with the factors of schur.hip / ldl.hip (the counterpart of
// linear_solve!/QDLDL_solve!, linear_solver.jl:52-60, qdldl.jl:330-351,592-640, in the order [z | y | x]):
//     dx = S^-1 ( b_x + gx'(omega_y b_y) + hx'(Omega_z b_z) )          forward/backward substitution with L, D of S
//     [dy; dz] = -Omega ( b_m - [gx; hx] dx )                           back-substitution (fused into k_recover, vectors.hip)
// plus a few O(N) helpers of the solve! driver.
#include "internal.hpp"
#include "device_utils.hpp"

namespace calipso {

__global__ void k_copy_pad(const double* __restrict__ src, int n, double* __restrict__ dst, int npad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < npad) dst[i] = i < n ? src[i] : 0.0;
}
void launch_copy_pad(calipso_hip_solver* s, const double* src, int n, double* dst, int npad) {
    hipLaunchKernelGGL(k_copy_pad, dim3((npad + 255) / 256), dim3(256), 0, s->stream, src, n, dst, npad);
}

__global__ void k_negate_copy(const double* __restrict__ src, double* __restrict__ dst, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = -1.0 * src[i];
}
void launch_negate_copy(calipso_hip_solver* s, const double* src, double* dst, int n) {
    hipLaunchKernelGGL(k_negate_copy, dim3((n + 255) / 256), dim3(256), 0, s->stream, src, dst, n);
}

// operands prepared by k_residual_symmetric: xbuf = [b_x; 0], t1 = Omega b_m.  Leaves dx in xbuf and t2 = [gx; hx] dx; the
// back-substitution [dy; dz] = -Omega (b_m - t2) is fused into k_recover.
void linear_solve_device(calipso_hip_solver* s, bool with_t2, bool rhs_ready) {
    const Dims& d = s->d;
    if (d.m && !rhs_ready) gemv_t(s, d.m, d.nx, s->Z, d.m, s->t1, s->xbuf, 1.0, 1.0, SP_Z);           // b_x + gx'(omega_y b_y) + hx'(Omega_z b_z)
    launch_trsv(s, s->xbuf);                                                       // xbuf = S^-1 xbuf
    if (d.m && with_t2) gemv_n(s, d.m, d.nx, s->Z, d.m, s->xbuf, s->t2, 1.0, 0.0, SP_Z);            // t2 = [gx; hx] dx
}

// stand-alone linear_solve! on a caller-provided right-hand side b (= "residual_symmetric"): operands, solve, back-substitution
__global__ void k_solve_prepare(Dims d, Scalars sc, ConeDev cd, const double* __restrict__ b, const double* __restrict__ wz,
                                const double* __restrict__ Wsoc, double* __restrict__ xbuf, double* __restrict__ t1) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < d.NP) { xbuf[i] = i < d.nx ? b[i] : 0.0; return; }
    const int e = i - d.NP;
    if (e >= d.m) return;
    const double* bm = b + d.nx;
    if (e < d.ne) {
        const double omega_y = -1.0 / (-1.0 / (sc.rho + sc.ep) + (0.0 - sc.ed));
        t1[e] = omega_y * bm[e];
    } else {
        const int c = e - d.ne;
        if (c < d.q) t1[e] = wz[c] * bm[e];
        else {
            const int j = cd.entry_soc[c], st = cd.soc_start[j], dim = cd.soc_dim[j];
            const double* W = Wsoc + cd.soc_woff[j];
            double v = 0.0;
            for (int q = 0; q < dim; ++q) v += W[(c - st) + q * dim] * bm[d.ne + st + q];
            t1[e] = v;
        }
    }
}
__global__ void k_solve_finish(Dims d, Scalars sc, ConeDev cd, const double* __restrict__ b, const double* __restrict__ dx,
                               const double* __restrict__ t2, const double* __restrict__ wz, const double* __restrict__ Wsoc,
                               double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.n) return;
    if (i < d.nx) { out[i] = dx[i]; return; }
    const int e = i - d.nx;
    const double* bm = b + d.nx;
    if (e < d.ne) {
        const double omega_y = -1.0 / (-1.0 / (sc.rho + sc.ep) + (0.0 - sc.ed));
        out[i] = -1.0 * omega_y * (bm[e] - t2[e]);
    } else {
        const int c = e - d.ne;
        if (c < d.q) out[i] = -1.0 * wz[c] * (bm[e] - t2[e]);
        else {
            const int j = cd.entry_soc[c], st = cd.soc_start[j], dim = cd.soc_dim[j];
            const double* W = Wsoc + cd.soc_woff[j];
            double v = 0.0;
            for (int q = 0; q < dim; ++q) v += W[(c - st) + q * dim] * (bm[d.ne + st + q] - t2[d.ne + st + q]);
            out[i] = -1.0 * v;
        }
    }
}
void launch_solve_from_b(calipso_hip_solver* s) {
    const Dims& d = s->d;
    hipLaunchKernelGGL(k_solve_prepare, dim3((d.NP + d.m + 255) / 256), dim3(256), 0, s->stream, d, s->sc, s->cone, s->residual_symmetric, s->wz,
                       s->Wsoc, s->xbuf, s->t1);
    linear_solve_device(s);
    hipLaunchKernelGGL(k_solve_finish, dim3((d.n + 255) / 256), dim3(256), 0, s->stream, d, s->sc, s->cone, s->residual_symmetric, s->xbuf, s->t2,
                       s->wz, s->Wsoc, s->step_symmetric);
}

// initialize_slacks! / initialize_duals!  initialize.jl:15-36: r = g(x0); nonnegative slacks/duals = 1;
// second-order = [1, .1, .1, ...]; y = z = 0  (cones/nonnegative.jl:2-8, second_order.jl:2-10)
__global__ void k_init_point(Batch bt, Dims d, ConeDev cd, const double* __restrict__ g, double* __restrict__ w) {
    inst_shift(bt, g, w);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < d.ne) { w[d.orr() + i] = g[i]; w[d.oy() + i] = 0.0; }
    if (i < d.nc) {
        double v = 1.0;
        const int j = cd.entry_soc[i];
        if (j >= 0 && i != cd.soc_start[j]) v = 0.1;
        w[d.os() + i] = v;
        w[d.ot() + i] = v;
        w[d.oz() + i] = 0.0;
    }
}
void launch_init_point(calipso_hip_solver* s) {
    const int n = s->d.ne > s->d.nc ? s->d.ne : s->d.nc;
    if (n == 0) return;
    const BatchSc B = batch_of(s);
    hipLaunchKernelGGL(k_init_point, dim3((n + 255) / 256, 1, B.b.n), dim3(256), 0, s->stream, B.b, s->d, s->cone, s->g, s->solution);
}

__global__ void k_lambda_update(BatchSc bt, Dims d, const double* __restrict__ w, double* __restrict__ lam) {
    inst_shift(bt.b, w, lam);
    const double rho = bt.scal(blockIdx.z).rho;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < d.ne) lam[i] = lam[i] + rho * w[d.orr() + i];
}
void launch_lambda_update(calipso_hip_solver* s) {
    if (s->d.ne == 0) return;
    const BatchSc B = batch_of(s);
    hipLaunchKernelGGL(k_lambda_update, dim3((s->d.ne + 255) / 256, 1, B.b.n), dim3(256), 0, s->stream, B, s->d, s->solution, s->lambda);
}

// residual_jacobian_parameters!  residual_jacobian_parameters.jl:1-40: rows x <- Lx_theta, y <- g_theta, z <- h_theta, rest 0
__global__ void k_jacobian_parameters(Dims d, const double* __restrict__ lgp, const double* __restrict__ gp, const double* __restrict__ hp,
                                      double* __restrict__ J) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int j = blockIdx.y;
    if (i >= d.N) return;
    double v = 0.0;
    if (i < d.nx) v = lgp[i + (size_t)j * d.nx];
    else if (i >= d.oy() && i < d.oz()) v = gp[(i - d.oy()) + (size_t)j * d.ne];
    else if (i >= d.oz() && i < d.ot()) v = hp[(i - d.oz()) + (size_t)j * d.nc];
    J[i + (size_t)j * d.N] = v;
}
void launch_jacobian_parameters(calipso_hip_solver* s) {
    if (s->d.np == 0) return;
    hipLaunchKernelGGL(k_jacobian_parameters, dim3((s->d.N + 255) / 256, s->d.np), dim3(256), 0, s->stream, s->d, s->lgp, s->gp, s->hp,
                       s->jacobian_parameters);
}

}  // namespace calipso
