// overlap_probe4.hip — an fp64-MFMA-saturating kernel A (fixed amount of work: iters x 4 independent accumulators per wave) with workgroups of 256 / 512 / 1024 threads at
// full occupancy, and the streaming kernel B on a second stream: how long does each take alone and together?  (overlap_probe3: a spinning MFMA kernel of 1024-thread
// workgroups stops other streams' kernels almost completely, one of 256-thread workgroups does not.)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int THREADS>
__global__ __launch_bounds__(THREADS) void mfmaA(int iters, double* out) {
    v4d acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (v4d){0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + blockIdx.x * 1e-6;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 1.2345e-300) out[0] = s;
}
__global__ __launch_bounds__(256) void kB(const double* __restrict__ src, size_t n, double* out) {
    const size_t stride = (size_t)gridDim.x * 256;
    double s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i + 15 * stride < n; i += 16 * stride) {
        double v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = __builtin_nontemporal_load(src + i + k * stride);
#pragma unroll
        for (int k = 0; k < 16; ++k) s += v[k];
    }
    if (s == 1.2345e-300) out[0] = s;
}
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static hipStream_t sa, sb;
static const double* src; static size_t n; static double* out;
template <int THREADS> void run(int waves_per_cu, int iters, int nB) {
    const int grid = 256 * waves_per_cu * 64 / THREADS;
    auto A = [&] { hipLaunchKernelGGL((mfmaA<THREADS>), dim3(grid), dim3(THREADS), 0, sa, iters, out); };
    auto B = [&] { for (int r = 0; r < nB; ++r) hipLaunchKernelGGL(kB, dim3(512), dim3(256), 0, sb, src, n, out + 8); };
    A(); B(); hipDeviceSynchronize();
    double t0 = now(); A(); hipStreamSynchronize(sa); const double a = now() - t0;
    t0 = now(); B(); hipStreamSynchronize(sb); const double b = now() - t0;
    t0 = now(); A(); B(); hipStreamSynchronize(sb); const double bdone = now() - t0; hipStreamSynchronize(sa); const double both = now() - t0;
    const double tf = 2.0 * 16 * 16 * 4 * 4.0 * iters * (double)grid * (THREADS / 64) * 1e-9;
    printf("A: %4d-thread workgroups, %2d waves per CU: alone %.2f ms (%.1f TFLOP/s), B (%d passes over 1 GiB) alone %.2f ms (%.2f TB/s) | together: B done %.2f ms, all %.2f ms = %.2f x (A + B), %.2f x max\n",
           THREADS, waves_per_cu, a, tf / a, nB, b, nB * n * 8 / b * 1e-9, bdone, both, both / (a + b), both / (a > b ? a : b));
}
int main() {
    n = (size_t)1 << 27;
    double* s_; hipMalloc(&s_, n * 8); hipMemset(s_, 0, n * 8); src = s_;
    hipMalloc(&out, 1024);
    int least, greatest; hipDeviceGetStreamPriorityRange(&least, &greatest);
    hipStreamCreateWithPriority(&sa, hipStreamNonBlocking, greatest); hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, least);
    for (int w : {4, 8, 16}) {
        const int iters = 40000 * 4 / w;      // the same total work whatever the occupancy (~3-4 ms)
        run<256>(w, iters, 24); run<512>(w, iters, 24); run<1024>(w < 16 ? 16 : w, 40000 * 4 / (w < 16 ? 16 : w), 24);
    }
    return 0;
}
