// Does a hipGraph with two dependent parallel branches overlap them on gfx950 / ROCm 7.2, and what does a cross-branch edge cost?
// Shape (the look-ahead schedule considered for ldl.hip): per step k   A(k): one workgroup, ~tA us (the pivot chain);   B(k) then C(k): many
// workgroups, ~tB + tC us (panel + trailing update).  Edges: A(k-1) -> A(k), A(k-1) -> B(k), C(k-1) -> A(k), C(k-1) -> B(k), B(k) -> C(k).
// If the branches overlap, a step costs max(tA, tB + tC) + edge overhead; on one stream it costs tA + tB + tC.
//   hipcc -O3 --offload-arch=gfx950 bench/graph_fork_join.hip -o /tmp/gfj && /tmp/gfj
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_spin(double* d, long long cycles) {          // busy for `cycles` clock ticks (s_memtime), then one store
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) { }
    if (threadIdx.x == 0) d[blockIdx.x] += 1.0;
}

int main() {
    double* d; CK(hipMalloc(&d, 8 * 4096)); CK(hipMemset(d, 0, 8 * 4096));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    const int K = 39;
    // calibrate: cycles per microsecond of the counter
    const long long per_us = 100;                               // s_memtime runs at 100 MHz on CDNA
    const long long tA = 20 * per_us, tB = 6 * per_us, tC = 9 * per_us;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    // (1) one stream, everything in sequence, captured as a graph
    hipGraph_t g; hipGraphExec_t ge1, ge2;
    CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < K; ++k) {
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, d, tA);
        hipLaunchKernelGGL(k_spin, dim3(38), dim3(256), 0, s1, d + 64, tB);
        hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s1, d + 128, tC);
    }
    CK(hipStreamEndCapture(s1, &g)); CK(hipGraphInstantiate(&ge1, g, nullptr, nullptr, 0)); CK(hipGraphDestroy(g));
    // (2) two branches
    hipEvent_t eA[K + 1], eC[K + 1], fork;
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
    for (int k = 0; k <= K; ++k) { CK(hipEventCreateWithFlags(&eA[k], hipEventDisableTiming)); CK(hipEventCreateWithFlags(&eC[k], hipEventDisableTiming)); }
    CK(hipStreamBeginCapture(s1, hipStreamCaptureModeThreadLocal));
    CK(hipEventRecord(fork, s1)); CK(hipStreamWaitEvent(s2, fork, 0));
    for (int k = 0; k < K; ++k) {
        if (k > 0) { CK(hipStreamWaitEvent(s1, eC[k - 1], 0)); CK(hipStreamWaitEvent(s2, eA[k - 1], 0)); }
        hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, d, tA);
        CK(hipEventRecord(eA[k], s1));
        hipLaunchKernelGGL(k_spin, dim3(38), dim3(256), 0, s2, d + 64, tB);
        hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s2, d + 128, tC);
        CK(hipEventRecord(eC[k], s2));
    }
    CK(hipStreamWaitEvent(s1, eC[K - 1], 0));
    CK(hipStreamEndCapture(s1, &g)); CK(hipGraphInstantiate(&ge2, g, nullptr, nullptr, 0)); CK(hipGraphDestroy(g));
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipGraphLaunch(ge1, s1)); CK(hipStreamSynchronize(s1));
        auto t0 = now(); for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge1, s1)); CK(hipStreamSynchronize(s1));
        const double t1 = us(t0, now()) / 20 / K;
        CK(hipGraphLaunch(ge2, s1)); CK(hipStreamSynchronize(s1));
        t0 = now(); for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge2, s1)); CK(hipStreamSynchronize(s1));
        const double t2 = us(t0, now()) / 20 / K;
        // (3) the same two-branch schedule issued live on two streams with events (no graph)
        t0 = now();
        for (int i = 0; i < 20; ++i) {
            for (int k = 0; k < K; ++k) {
                if (k > 0 || i > 0) { CK(hipStreamWaitEvent(s1, eC[(k + K - 1) % K], 0)); CK(hipStreamWaitEvent(s2, eA[(k + K - 1) % K], 0)); }
                hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, s1, d, tA);
                CK(hipEventRecord(eA[k], s1));
                hipLaunchKernelGGL(k_spin, dim3(38), dim3(256), 0, s2, d + 64, tB);
                hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, s2, d + 128, tC);
                CK(hipEventRecord(eC[k], s2));
            }
        }
        CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
        const double t3 = us(t0, now()) / 20 / K;
        printf("per step (A 20 us | B 6 + C 9 us): one-stream graph %.1f us   two-branch graph %.1f us   two streams + events, live %.1f us\n", t1, t2, t3);
    }
    return 0;
}
