// small.hip — batched LDS-resident LDL^T factor + multi-right-hand-side solve for SMALL KKT systems (n <= 128): one workgroup per
// problem instance, the whole condensed matrix in the CU's 160 KiB LDS, the grid runs over the instances.
//
// Use: BASELINE config C5 — the cart-pole MPC auto-tuning loop of the reference (examples/autotuning/cartpole.jl:179-227) needs, per
// MPC step, the solution sensitivities of an n = 89 system for p = 102 parameters: differentiate! (src/solver/differentiate.jl:1-61)
// factors K once and solves one condensed system per parameter column (:29-58).  Thousands of such steps (initial states, tuning
// iterates) are independent; here they are one launch.  Also C2-sized solves (pendulum, n = 56).  The general path (schur.hip + ldl.hip)
// pads nx to 64 * 2^k and spends ~430 launches per step — latency-bound at these sizes.
//
// Semantics are those of the LinearSolver seam (linear_solver.jl:19-60, qdldl.jl:134-188): only triu(K) is read, no pivoting, D is the
// diagonal of the LDL^T in the natural order, X = K^-1 B.  Per instance and per pivot column ONE workgroup barrier: the unscaled pivot
// column travels through a double-buffered LDS vector, the trailing update is spread over 256 threads.
#include <algorithm>
#include <vector>

#include "internal.hpp"
#include "device_utils.hpp"

namespace {

constexpr int SMALL_MAX = 128;
constexpr int SMALL_THREADS = 256;

// K: column-major n x n per instance (upper triangle read); B / X: column-major n x nrhs per instance; D: n per instance.
// LDS (doubles): Ls[n][n+1] | ycol[2][n] | dloc[n] | Bs[n][cw+1]   (cw = right-hand-side columns per pass)
__global__ __launch_bounds__(SMALL_THREADS) void k_small_ldl_solve(int n, int nrhs, int cw, const double* __restrict__ K, const double* __restrict__ B,
                                                                   double* __restrict__ X, double* __restrict__ D) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int ld = n + 1, ldb = cw + 1;
    double* Ls = sm;
    double* ycol = Ls + (size_t)n * ld;
    double* dloc = ycol + 2 * n;
    double* Bs = dloc + n;
    const int tid = threadIdx.x;
    const size_t inst = blockIdx.x;
    const double* Kg = K + inst * (size_t)n * n;
    // lower triangle of the working matrix from the UPPER triangle of K: L[i][j] = K[j][i] for j <= i (triu!, linear_solver.jl:23)
    for (int e = tid; e < n * n; e += SMALL_THREADS) {
        const int r = e % n, c = e / n;                 // K[r + c*n], coalesced
        if (r <= c) Ls[c * ld + r] = Kg[e];
    }
    __syncthreads();
    if (tid < n) ycol[tid] = Ls[tid * ld + 0];           // pivot column 0 (rows >= 0)
    const int ti = tid >> 4, tk = tid & 15;
    for (int j = 0; j < n; ++j) {
        __syncthreads();
        const double* y = ycol + (j & 1) * n;
        double* yn = ycol + ((j + 1) & 1) * n;
        const double d = y[j];
        const double rinv = 1.0 / d;
        if (tid == 0) { D[inst * n + j] = d; dloc[j] = d; }
        // trailing update A[i][k] -= (y_i / d) y_k for j < k <= i; the owners of column j + 1 publish it (unscaled) for the next step and
        // the scaled L[i][j] is stored by the owner of (i, j + 1) (or of the last row's diagonal-less case below)
        for (int i = j + 1 + ti; i < n; i += 16) {
            const double li = y[i] * rinv;
            for (int k = j + 1 + tk; k <= i; k += 16) {
                const double v = Ls[i * ld + k] - li * y[k];
                Ls[i * ld + k] = v;
                if (k == j + 1) { yn[i] = v; Ls[i * ld + j] = li; }
            }
        }
    }
    __syncthreads();
    // right-hand sides, cw columns per pass: forward substitution (unit lower L), diagonal scaling, backward substitution (L')
    const int tc = tid & 31, tr = tid >> 5;
    for (int c0 = 0; c0 < nrhs; c0 += cw) {
        const int w = min(cw, nrhs - c0);
        const double* Bg = B + inst * (size_t)n * nrhs + (size_t)c0 * n;
        for (int e = tid; e < n * w; e += SMALL_THREADS) { const int i = e % n, c = e / n; Bs[i * ldb + c] = Bg[e]; }
        __syncthreads();
        for (int k = 0; k < n - 1; ++k) {                 // forward: B[i][:] -= L[i][k] B[k][:]
            for (int i = k + 1 + tr; i < n; i += 8) {
                const double l = Ls[i * ld + k];
                for (int c = tc; c < w; c += 32) Bs[i * ldb + c] -= l * Bs[k * ldb + c];
            }
            __syncthreads();
        }
        for (int e = tid; e < n * w; e += SMALL_THREADS) { const int c = e % w, i = e / w; Bs[i * ldb + c] *= 1.0 / dloc[i]; }   // x .*= Dinv (qdldl.jl:338)
        __syncthreads();
        for (int k = n - 1; k > 0; --k) {                 // backward: B[i][:] -= L[k][i] B[k][:] for i < k
            for (int i = tr; i < k; i += 8) {
                const double l = Ls[k * ld + i];
                for (int c = tc; c < w; c += 32) Bs[i * ldb + c] -= l * Bs[k * ldb + c];
            }
            __syncthreads();
        }
        double* Xg = X + inst * (size_t)n * nrhs + (size_t)c0 * n;
        for (int e = tid; e < n * w; e += SMALL_THREADS) { const int i = e % n, c = e / n; Xg[e] = Bs[i * ldb + c]; }
        __syncthreads();
    }
}

}  // namespace

struct calipso_hip_small {
    int n = 0, nrhs = 0, batch = 0, device = 0, cw = 0;
    size_t lds_bytes = 0;
    double *K = nullptr, *B = nullptr, *X = nullptr, *D = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float last_ms = 0.f;
    std::string err;
};

static thread_local std::string g_small_err;
#define SK(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) { (s ? s->err : g_small_err) = std::string(#call) + ": " + hipGetErrorString(e__); return CALIPSO_ERR_HIP; } } while (0)

extern "C" {

const char* calipso_hip_small_last_error(calipso_hip_small* s) { return s ? s->err.c_str() : g_small_err.c_str(); }

// `batch` systems of size n (<= 128) with nrhs right-hand sides each, on `device`
int32_t calipso_hip_small_create(int64_t n, int64_t nrhs, int64_t batch, int32_t device, calipso_hip_small** out) {
    calipso_hip_small* s = nullptr;
    if (!out) return CALIPSO_ERR_ARGUMENT;
    *out = nullptr;
    if (n < 1 || n > SMALL_MAX || nrhs < 1 || batch < 1) { g_small_err = "calipso_hip_small_create: need 1 <= n <= 128, nrhs >= 1, batch >= 1"; return CALIPSO_ERR_ARGUMENT; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { g_small_err = "no HIP device available (libcalipso_hip has no CPU path)"; return CALIPSO_ERR_HIP; }
    if (device < 0 || device >= ndev) { g_small_err = "device ordinal out of range"; return CALIPSO_ERR_ARGUMENT; }
    s = new calipso_hip_small();
    s->n = (int)n; s->nrhs = (int)nrhs; s->batch = (int)batch; s->device = device;
    *out = s;
    SK(hipSetDevice(device));
    // right-hand-side columns per pass: whatever fits beside the matrix in 160 KiB of LDS (at most nrhs)
    const size_t fixed = ((size_t)n * (n + 1) + 3 * (size_t)n) * sizeof(double);
    const size_t cap = 160 * 1024 - 1024;
    long cw = (long)((cap - fixed) / (sizeof(double) * (size_t)n)) - 1;
    if (cw > nrhs) cw = (long)nrhs;
    if (cw < 1) { s->err = "system too large for the LDS-resident path"; return CALIPSO_ERR_ARGUMENT; }
    s->cw = (int)cw;
    s->lds_bytes = fixed + (size_t)n * (size_t)(cw + 1) * sizeof(double);
    SK(hipFuncSetAttribute((const void*)k_small_ldl_solve, hipFuncAttributeMaxDynamicSharedMemorySize, (int)s->lds_bytes));
    SK(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    SK(hipEventCreate(&s->e0)); SK(hipEventCreate(&s->e1));
    SK(hipMalloc((void**)&s->K, sizeof(double) * (size_t)batch * n * n));
    SK(hipMalloc((void**)&s->B, sizeof(double) * (size_t)batch * n * nrhs));
    SK(hipMalloc((void**)&s->X, sizeof(double) * (size_t)batch * n * nrhs));
    SK(hipMalloc((void**)&s->D, sizeof(double) * (size_t)batch * n));
    return CALIPSO_OK;
}

int32_t calipso_hip_small_destroy(calipso_hip_small* s) {
    if (!s) return CALIPSO_OK;
    (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    for (double* p : {s->K, s->B, s->X, s->D}) if (p) (void)hipFree(p);
    if (s->e0) (void)hipEventDestroy(s->e0);
    if (s->e1) (void)hipEventDestroy(s->e1);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
    return CALIPSO_OK;
}

// upload: K = batch x (n x n) column-major (only the upper triangles are read), B = batch x (n x nrhs) column-major; either may be NULL (keep)
int32_t calipso_hip_small_set(calipso_hip_small* s, const double* K, const double* B) {
    if (!s) return CALIPSO_ERR_ARGUMENT;
    SK(hipSetDevice(s->device));
    if (K) SK(hipMemcpyAsync(s->K, K, sizeof(double) * (size_t)s->batch * s->n * s->n, hipMemcpyHostToDevice, s->stream));
    if (B) SK(hipMemcpyAsync(s->B, B, sizeof(double) * (size_t)s->batch * s->n * s->nrhs, hipMemcpyHostToDevice, s->stream));
    SK(hipStreamSynchronize(s->stream));
    return CALIPSO_OK;
}

// factorize! + linear_solve! of every instance: ONE launch, matrices and right-hand sides already resident; *ms (may be NULL) = duration
// of that launch from HIP events on its stream
int32_t calipso_hip_small_solve(calipso_hip_small* s, double* ms) {
    if (!s) return CALIPSO_ERR_ARGUMENT;
    SK(hipSetDevice(s->device));
    SK(hipEventRecord(s->e0, s->stream));
    hipLaunchKernelGGL(k_small_ldl_solve, dim3((unsigned)s->batch), dim3(SMALL_THREADS), s->lds_bytes, s->stream, s->n, s->nrhs, s->cw, s->K, s->B, s->X, s->D);
    SK(hipEventRecord(s->e1, s->stream));
    SK(hipStreamSynchronize(s->stream));
    SK(hipGetLastError());
    SK(hipEventElapsedTime(&s->last_ms, s->e0, s->e1));
    if (ms) *ms = (double)s->last_ms;
    return CALIPSO_OK;
}

// download: X = batch x (n x nrhs), inertia = batch x 3 as compute_inertia! reports it (linear_solver.jl:33-44; an exact zero pivot gives
// positive = -1 and counts the rest of D as zeros, qdldl.jl:444,456,579); either may be NULL.  Returns the number of instances that met a
// zero pivot (their X is not meaningful), or a negative status.
int32_t calipso_hip_small_get(calipso_hip_small* s, double* X, int64_t* inertia) {
    if (!s) return CALIPSO_ERR_ARGUMENT;
    SK(hipSetDevice(s->device));
    if (X) SK(hipMemcpyAsync(X, s->X, sizeof(double) * (size_t)s->batch * s->n * s->nrhs, hipMemcpyDeviceToHost, s->stream));
    std::vector<double> D;
    if (inertia) { D.resize((size_t)s->batch * s->n); SK(hipMemcpyAsync(D.data(), s->D, sizeof(double) * D.size(), hipMemcpyDeviceToHost, s->stream)); }
    SK(hipStreamSynchronize(s->stream));
    int32_t bad = 0;
    if (inertia)
        for (int b = 0; b < s->batch; ++b) {
            int64_t pos = 0, nonpos = 0, zero = 0; int k = 0;
            for (; k < s->n; ++k) { const double d = D[(size_t)b * s->n + k]; if (d == 0.0) break; pos += d > 0.0; nonpos += d <= 0.0; }
            if (k < s->n) { zero = s->n - k; nonpos += s->n - k; pos = -1; bad += 1; }
            inertia[3 * b] = pos; inertia[3 * b + 1] = nonpos; inertia[3 * b + 2] = zero;
        }
    return bad;
}

}  // extern "C"
