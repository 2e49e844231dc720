// solvek.hip — the condensed solve  K [dx; dy; dz] = b  with the factors of schur.hip / ldl.hip (the counterpart of
// linear_solve!/QDLDL_solve!, linear_solver.jl:52-60, qdldl.jl:330-351,592-640, in the order [z | y | x]):
//     dx = S^-1 ( b_x + gx'(omega_y b_y) + hx'(Omega_z b_z) )          forward/backward substitution with L, D of S
//     [dy; dz] = -Omega ( b_m - [gx; hx] dx )                           back-substitution through the constraint pivots
// plus a few O(N) helpers of the solve! driver.
#include "internal.hpp"
#include "device_utils.hpp"

namespace calipso {

__global__ void k_omega_apply(Dims d, Scalars sc, ConeDev cd, const double* __restrict__ in, const double* __restrict__ wz,
                              const double* __restrict__ Wsoc, double* __restrict__ out, double sign) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.m) return;
    if (i < d.ne) {
        const double omega_y = -1.0 / (-1.0 / (sc.rho + sc.ep) + (0.0 - sc.ed));
        out[i] = sign * omega_y * in[i];
    } else {
        const int c = i - d.ne;
        const double* z = in + d.ne;
        if (c < d.q) {
            out[i] = sign * wz[c] * z[c];
        } else {
            const int j = cd.entry_soc[c];
            const int st = cd.soc_start[j], dim = cd.soc_dim[j];
            const double* W = Wsoc + cd.soc_woff[j];
            double v = 0.0;
            for (int b = 0; b < dim; ++b) v += W[(c - st) + b * dim] * z[st + b];
            out[i] = sign * v;
        }
    }
}
void launch_omega_apply(calipso_hip_solver* s, const double* in_m, double* out_m, double sign) {
    if (s->d.m == 0) return;
    hipLaunchKernelGGL(k_omega_apply, dim3((s->d.m + 255) / 256), dim3(256), 0, s->stream, s->d, s->sc, s->cone, in_m, s->wz, s->Wsoc, out_m, sign);
}

__global__ void k_copy_pad(const double* __restrict__ src, int n, double* __restrict__ dst, int npad) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < npad) dst[i] = i < n ? src[i] : 0.0;
}
void launch_copy_pad(calipso_hip_solver* s, const double* src, int n, double* dst, int npad) {
    hipLaunchKernelGGL(k_copy_pad, dim3((npad + 255) / 256), dim3(256), 0, s->stream, src, n, dst, npad);
}

__global__ void k_sub(const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] - b[i];
}
void launch_sub(calipso_hip_solver* s, const double* a, const double* b, double* out, int n) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_sub, dim3((n + 255) / 256), dim3(256), 0, s->stream, a, b, out, n);
}

__global__ void k_negate_copy(const double* __restrict__ src, double* __restrict__ dst, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = -1.0 * src[i];
}
void launch_negate_copy(calipso_hip_solver* s, const double* src, double* dst, int n) {
    hipLaunchKernelGGL(k_negate_copy, dim3((n + 255) / 256), dim3(256), 0, s->stream, src, dst, n);
}

void linear_solve_device(calipso_hip_solver* s) {
    const Dims& d = s->d;
    const double* b = s->residual_symmetric;
    double* out = s->step_symmetric;
    launch_omega_apply(s, b + d.nx, s->t1, 1.0);                                   // t1 = Omega b_m
    launch_copy_pad(s, b, d.nx, s->xbuf, d.NP);
    if (d.ne) gemv_t(s, d.ne, d.nx, s->gx, d.ne, s->t1, s->xbuf, 1.0, 1.0);
    if (d.nc) gemv_t(s, d.nc, d.nx, s->hx, d.nc, s->t1 + d.ne, s->xbuf, 1.0, 1.0);
    launch_trsv(s, s->xbuf);                                                       // xbuf = S^-1 xbuf
    if (d.ne) gemv_n(s, d.ne, d.nx, s->gx, d.ne, s->xbuf, s->t2, 1.0, 0.0);
    if (d.nc) gemv_n(s, d.nc, d.nx, s->hx, d.nc, s->xbuf, s->t2 + d.ne, 1.0, 0.0);
    launch_sub(s, b + d.nx, s->t2, s->t2, d.m);                                    // t2 = b_m - Z dx
    launch_omega_apply(s, s->t2, out + d.nx, -1.0);                                // [dy; dz] = -Omega t2
    launch_copy_pad(s, s->xbuf, d.nx, out, d.nx);
}

// initialize_slacks! / initialize_duals!  initialize.jl:15-36: r = g(x0); nonnegative slacks/duals = 1;
// second-order = [1, .1, .1, ...]; y = z = 0  (cones/nonnegative.jl:2-8, second_order.jl:2-10)
__global__ void k_init_point(Dims d, ConeDev cd, const double* __restrict__ g, double* __restrict__ w) {
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
    hipLaunchKernelGGL(k_init_point, dim3((n + 255) / 256), dim3(256), 0, s->stream, s->d, s->cone, s->g, s->solution);
}

__global__ void k_lambda_update(Dims d, double rho, const double* __restrict__ w, double* __restrict__ lam) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < d.ne) lam[i] = lam[i] + rho * w[d.orr() + i];
}
void launch_lambda_update(calipso_hip_solver* s) {
    if (s->d.ne == 0) return;
    hipLaunchKernelGGL(k_lambda_update, dim3((s->d.ne + 255) / 256), dim3(256), 0, s->stream, s->d, s->sc.rho, s->solution, s->lambda);
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
