// user_device_eval.hip — what a USER of libcalipso_hip.so writes to keep the evaluation of f, g, h and their derivatives on the GPU
// (include/calipso_hip.h: calipso_device_eval_fn).  Test fixture: two evaluators, built into
// tests/device_eval/libuser_device_eval.so by __graft_entry__.build() and registered through calipso_hip_set_device_evaluator.
//   wachter_device_eval   the README / test/solver/wachter.jl:3-15 problem in closed form:
//                         f = x1,  g = [x1^2 - x2 - 1; x1 - x3 - 1/2],  h = [x2; x3]
//   qp_device_eval        min c x'Px + q'x  s.t. Ax - b = 0, h - Gx in K with the problem data in the user's own device buffers
// Both only ENQUEUE kernels on the stream they are given and write straight into the solver's device buffers.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/calipso_hip.h"

__global__ void k_wachter(uint32_t flags, const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ z, calipso_device_problem_data o) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double x1 = x[0], x2 = x[1], x3 = x[2];
    const int64_t ld = o.jacobian_ld;
    if (flags & CALIPSO_EVAL_OBJECTIVE) o.objective[0] = x1;
    if (flags & CALIPSO_EVAL_OBJECTIVE_GRADIENT) { o.objective_gradient_variables[0] = 1.0; o.objective_gradient_variables[1] = 0.0; o.objective_gradient_variables[2] = 0.0; }
    if (flags & CALIPSO_EVAL_EQUALITY) { o.equality_constraint[0] = x1 * x1 - x2 - 1.0; o.equality_constraint[1] = x1 - x3 - 0.5; }
    if (flags & CALIPSO_EVAL_EQUALITY_JACOBIAN) {
        double* J = o.equality_jacobian_variables;         // 2 x 3, leading dimension ld
        J[0 + 0 * ld] = 2.0 * x1; J[0 + 1 * ld] = -1.0; J[0 + 2 * ld] = 0.0;
        J[1 + 0 * ld] = 1.0;      J[1 + 1 * ld] = 0.0;  J[1 + 2 * ld] = -1.0;
    }
    if (flags & CALIPSO_EVAL_EQUALITY_DUAL_GRADIENT) {
        o.equality_dual_jacobian_variables[0] = 2.0 * x1 * y[0] + y[1];
        o.equality_dual_jacobian_variables[1] = -y[0];
        o.equality_dual_jacobian_variables[2] = -y[1];
    }
    if (flags & CALIPSO_EVAL_CONE) { o.cone_constraint[0] = x2; o.cone_constraint[1] = x3; }
    if (flags & CALIPSO_EVAL_CONE_JACOBIAN) {
        double* J = o.cone_jacobian_variables;             // 2 x 3, leading dimension ld
        J[0 + 0 * ld] = 0.0; J[0 + 1 * ld] = 1.0; J[0 + 2 * ld] = 0.0;
        J[1 + 0 * ld] = 0.0; J[1 + 1 * ld] = 0.0; J[1 + 2 * ld] = 1.0;
    }
    if (flags & CALIPSO_EVAL_CONE_DUAL_GRADIENT) { o.cone_dual_jacobian_variables[0] = 0.0; o.cone_dual_jacobian_variables[1] = z[0]; o.cone_dual_jacobian_variables[2] = z[1]; }
    if (flags & (CALIPSO_EVAL_OBJECTIVE_HESSIAN | CALIPSO_EVAL_EQUALITY_DUAL_HESSIAN | CALIPSO_EVAL_CONE_DUAL_HESSIAN)) {
        for (int k = 0; k < 9; ++k) o.lagrangian_hessian[k] = 0.0;
        if (flags & CALIPSO_EVAL_EQUALITY_DUAL_HESSIAN) o.lagrangian_hessian[0] = 2.0 * y[0];      // only (g'y)_xx is non-zero
    }
}

struct QpUser { int nx, ne, nc; double c; double *P, *q, *A, *b, *G, *h; };     // row-major P (nx x nx), A (ne x nx), G (nc x nx)

__global__ void k_qp_user(uint32_t flags, QpUser u, const double* __restrict__ x, const double* __restrict__ y, const double* __restrict__ z, calipso_device_problem_data o) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
    const int nx = u.nx, ne = u.ne, nc = u.nc;
    const int64_t ld = o.jacobian_ld;
    if ((flags & CALIPSO_EVAL_OBJECTIVE) && t == 0) {
        double f = 0.0;
        for (int i = 0; i < nx; ++i) { double r = 0.0; for (int j = 0; j < nx; ++j) r += u.P[i * nx + j] * x[j]; f += u.c * x[i] * r + u.q[i] * x[i]; }
        o.objective[0] = f;
    }
    for (int i = t; i < nx; i += nt) {
        if (flags & CALIPSO_EVAL_OBJECTIVE_GRADIENT) {
            double r = 0.0;
            for (int j = 0; j < nx; ++j) r += (u.P[i * nx + j] + u.P[j * nx + i]) * x[j];
            o.objective_gradient_variables[i] = u.c * r + u.q[i];
        }
        if (flags & CALIPSO_EVAL_EQUALITY_DUAL_GRADIENT) { double r = 0.0; for (int k = 0; k < ne; ++k) r += u.A[k * nx + i] * y[k]; o.equality_dual_jacobian_variables[i] = r; }
        if (flags & CALIPSO_EVAL_CONE_DUAL_GRADIENT) { double r = 0.0; for (int k = 0; k < nc; ++k) r += -u.G[k * nx + i] * z[k]; o.cone_dual_jacobian_variables[i] = r; }
        if (flags & (CALIPSO_EVAL_OBJECTIVE_HESSIAN | CALIPSO_EVAL_EQUALITY_DUAL_HESSIAN | CALIPSO_EVAL_CONE_DUAL_HESSIAN))
            for (int j = 0; j < nx; ++j) o.lagrangian_hessian[i + (size_t)j * nx] = (flags & CALIPSO_EVAL_OBJECTIVE_HESSIAN) ? u.c * (u.P[i * nx + j] + u.P[j * nx + i]) : 0.0;
        if (flags & CALIPSO_EVAL_EQUALITY_JACOBIAN) for (int k = 0; k < ne; ++k) o.equality_jacobian_variables[k + (size_t)i * ld] = u.A[k * nx + i];
        if (flags & CALIPSO_EVAL_CONE_JACOBIAN) for (int k = 0; k < nc; ++k) o.cone_jacobian_variables[k + (size_t)i * ld] = -u.G[k * nx + i];
    }
    for (int k = t; k < ne; k += nt)
        if (flags & CALIPSO_EVAL_EQUALITY) { double r = -u.b[k]; for (int j = 0; j < nx; ++j) r += u.A[k * nx + j] * x[j]; o.equality_constraint[k] = r; }
    for (int k = t; k < nc; k += nt)
        if (flags & CALIPSO_EVAL_CONE) { double r = u.h[k]; for (int j = 0; j < nx; ++j) r -= u.G[k * nx + j] * x[j]; o.cone_constraint[k] = r; }
}

// the same quadratic program for a STRUCTURED handle: the Jacobians and the Hessian go straight into the handle's packed blocks (calipso_device_block_eval_fn) —
// one workgroup per block walks its entries; rows of the stacked matrix [equality; cone]: row r < ne is row r of A, row r >= ne is row r - ne of -G
__global__ void k_qp_user_blocks(uint32_t flags, QpUser u, calipso_device_block_data o) {
    const int nj = (int)o.n_jacobian_blocks, nh = (int)o.n_hessian_blocks;
    const int b = blockIdx.x;
    if (b < nj) {
        const calipso_device_block k = o.jacobian_blocks_device[b];
        for (int64_t idx = threadIdx.x; idx < k.nrows * k.ncols; idx += blockDim.x) {
            const int64_t i = idx % k.nrows, j = idx / k.nrows, r = k.row0 + i, c = k.col0 + j;
            const bool eq = r < u.ne;
            if (eq && !(flags & CALIPSO_EVAL_EQUALITY_JACOBIAN)) continue;
            if (!eq && !(flags & CALIPSO_EVAL_CONE_JACOBIAN)) continue;
            k.values[i + j * k.ld] = eq ? u.A[r * u.nx + c] : -u.G[(r - u.ne) * u.nx + c];
        }
    } else if (b < nj + nh && (flags & (CALIPSO_EVAL_OBJECTIVE_HESSIAN | CALIPSO_EVAL_EQUALITY_DUAL_HESSIAN | CALIPSO_EVAL_CONE_DUAL_HESSIAN))) {
        const calipso_device_block k = o.hessian_blocks_device[b - nj];
        for (int64_t idx = threadIdx.x; idx < k.nrows * k.ncols; idx += blockDim.x) {
            const int64_t i = idx % k.nrows, j = idx / k.nrows, r = k.row0 + i, c = k.col0 + j;
            k.values[i + j * k.ld] = (flags & CALIPSO_EVAL_OBJECTIVE_HESSIAN) ? u.c * (u.P[r * u.nx + c] + u.P[c * u.nx + r]) : 0.0;
        }
    }
}

extern "C" {

int32_t qp_block_device_eval(void* user, uint32_t flags, const double* x, const double* y, const double* z, const double* theta,
                             const calipso_device_block_data* out, void* hip_stream) {
    (void)theta;
    const QpUser* u = (const QpUser*)user;
    if (!u || out->nx != u->nx || out->ne != u->ne || out->nc != u->nc) return 1;
    // the vector fields through the dense-layout kernel with the matrix flags masked off ...
    calipso_device_problem_data v = {};
    v.objective = out->objective; v.objective_gradient_variables = out->objective_gradient_variables; v.equality_constraint = out->equality_constraint;
    v.cone_constraint = out->cone_constraint; v.equality_dual_jacobian_variables = out->equality_dual_jacobian_variables;
    v.cone_dual_jacobian_variables = out->cone_dual_jacobian_variables; v.nx = out->nx; v.np = out->np; v.ne = out->ne; v.nc = out->nc;
    const uint32_t matrices = CALIPSO_EVAL_EQUALITY_JACOBIAN | CALIPSO_EVAL_CONE_JACOBIAN | CALIPSO_EVAL_OBJECTIVE_HESSIAN | CALIPSO_EVAL_EQUALITY_DUAL_HESSIAN | CALIPSO_EVAL_CONE_DUAL_HESSIAN;
    hipLaunchKernelGGL(k_qp_user, dim3(8), dim3(64), 0, (hipStream_t)hip_stream, flags & ~matrices, *u, x, y, z, v);
    // ... the matrices block by block
    const int nb = (int)(out->n_jacobian_blocks + out->n_hessian_blocks);
    if ((flags & matrices) && nb > 0) hipLaunchKernelGGL(k_qp_user_blocks, dim3(nb), dim3(128), 0, (hipStream_t)hip_stream, flags, *u, *out);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

int32_t wachter_device_eval(void* user, uint32_t flags, const double* x, const double* y, const double* z, const double* theta,
                            const calipso_device_problem_data* out, void* hip_stream) {
    (void)user; (void)theta;
    if (out->nx != 3 || out->ne != 2 || out->nc != 2) return 1;
    hipLaunchKernelGGL(k_wachter, dim3(1), dim3(64), 0, (hipStream_t)hip_stream, flags, x, y, z, *out);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

void* qp_user_create(int32_t nx, int32_t ne, int32_t nc, double c, const double* P, const double* q, const double* A, const double* b, const double* G,
                     const double* h) {
    QpUser* u = new QpUser();
    u->nx = nx; u->ne = ne; u->nc = nc; u->c = c;
    auto up = [](const double* src, size_t n) { double* d = nullptr; if (hipMalloc((void**)&d, sizeof(double) * (n ? n : 1)) != hipSuccess) return (double*)nullptr;
                                                if (n) (void)hipMemcpy(d, src, sizeof(double) * n, hipMemcpyHostToDevice); return d; };
    u->P = up(P, (size_t)nx * nx); u->q = up(q, nx); u->A = up(A, (size_t)ne * nx); u->b = up(b, ne); u->G = up(G, (size_t)nc * nx); u->h = up(h, nc);
    return u;
}
void qp_user_destroy(void* p) {
    QpUser* u = (QpUser*)p;
    if (!u) return;
    for (double* d : {u->P, u->q, u->A, u->b, u->G, u->h}) if (d) (void)hipFree(d);
    delete u;
}
int32_t qp_device_eval(void* user, uint32_t flags, const double* x, const double* y, const double* z, const double* theta,
                       const calipso_device_problem_data* out, void* hip_stream) {
    (void)theta;
    const QpUser* u = (const QpUser*)user;
    if (!u || out->nx != u->nx || out->ne != u->ne || out->nc != u->nc) return 1;
    hipLaunchKernelGGL(k_qp_user, dim3(8), dim3(64), 0, (hipStream_t)hip_stream, flags, *u, x, y, z, *out);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}

}  // extern "C"
