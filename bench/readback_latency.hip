// What one blocking scalar read-back costs between two dependent kernels of a stream (the Newton loop has ~14 per step: refinement norms, merit / step-length
// decisions).  Variants: (a) kernels only, no read-back; (b) hipMemcpyAsync of 8 bytes to pinned memory + hipStreamSynchronize (what api.hip: read_scalars
// does); (c) the kernel stores the scalar and a sequence number straight into mapped pinned host memory (system-scope release) and the host spins on the
// sequence number — no copy engine, no stream synchronisation.
//   hipcc -O3 --offload-arch=gfx950 bench/readback_latency.hip -o /tmp/readback && /tmp/readback
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_work(double* d, int n) {                     // a few microseconds of dependent work
    double v = d[0];
    for (int i = 0; i < n; ++i) v = v * 1.0000001 + 1e-9;
    if (threadIdx.x == 0 && blockIdx.x == 0) d[0] = v;
}
__global__ void k_publish(const double* d, volatile double* hval, volatile unsigned long long* hseq, unsigned long long seq) {
    hval[0] = d[0];
    __atomic_thread_fence(__ATOMIC_RELEASE);                    // system scope by default on a host-visible store? make it explicit:
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    hseq[0] = seq;
}

int main() {
    double* d; CK(hipMalloc(&d, 64)); CK(hipMemset(d, 0, 64));
    double* hpin; CK(hipHostMalloc((void**)&hpin, 64, hipHostMallocDefault));
    double* hmap; unsigned long long* hseq;
    CK(hipHostMalloc((void**)&hmap, 64, hipHostMallocMapped | hipHostMallocCoherent)); CK(hipHostMalloc((void**)&hseq, 64, hipHostMallocMapped | hipHostMallocCoherent));
    double* dmap; unsigned long long* dseq;
    CK(hipHostGetDevicePointer((void**)&dmap, hmap, 0)); CK(hipHostGetDevicePointer((void**)&dseq, hseq, 0));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    const int N = 2000, W = 200;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    for (int rep = 0; rep < 2; ++rep) {
        // (a)
        for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, st, d, W);
        CK(hipStreamSynchronize(st));
        auto t0 = now();
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, st, d, W);
        CK(hipStreamSynchronize(st));
        const double ta = us(t0, now()) / N;
        // (b)
        t0 = now();
        for (int i = 0; i < N; ++i) {
            hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, st, d, W);
            CK(hipMemcpyAsync(hpin, d, 8, hipMemcpyDeviceToHost, st));
            CK(hipStreamSynchronize(st));
        }
        const double tb = us(t0, now()) / N;
        // (c)
        hseq[0] = 0;
        t0 = now();
        for (int i = 0; i < N; ++i) {
            hipLaunchKernelGGL(k_work, dim3(1), dim3(64), 0, st, d, W);
            hipLaunchKernelGGL(k_publish, dim3(1), dim3(1), 0, st, d, dmap, dseq, (unsigned long long)(i + 1));
            while (*(volatile unsigned long long*)hseq != (unsigned long long)(i + 1)) { }
        }
        const double tc = us(t0, now()) / N;
        CK(hipStreamSynchronize(st));
        printf("rep %d: kernel only %.2f us | + memcpyAsync(8 B) + streamSynchronize %.2f us (read-back costs %.2f) | + publish kernel to mapped host memory, host spins %.2f us (costs %.2f)\n",
               rep, ta, tb, tb - ta, tc, tc - ta);
    }
    return 0;
}
