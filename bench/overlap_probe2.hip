// overlap_probe2.hip — follow-up of overlap_probe.hip: many SHORT one-workgroup kernels on two streams (the shape of the library's pivot chain and of its second stream) —
// do the two sequences run side by side, or one kernel at a time?  Prints the wall time of each sequence alone and of both queued together.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ __launch_bounds__(256) void spin(long long ticks, double* out) {      // ~ticks x 10 ns on ONE workgroup
    const long long t0 = wall_clock64();
    double s = 0;
    while (wall_clock64() - t0 < ticks) s += 1.0;
    if (s < 0) out[0] = s;
}
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    double* out; hipMalloc(&out, 64);
    int least, greatest; hipDeviceGetStreamPriorityRange(&least, &greatest);
    hipStream_t s[3];
    for (int i = 0; i < 3; ++i) hipStreamCreateWithPriority(&s[i], hipStreamNonBlocking, greatest + i > least ? least : greatest + i);
    for (int grid : {1, 64, 256}) for (int us : {20, 200, 2000}) {
        const int n = 4000 / us > 1 ? 4000 / us : 2;      // ~4 ms per sequence
        auto seq = [&](hipStream_t st) { for (int k = 0; k < n; ++k) hipLaunchKernelGGL(spin, dim3(grid), dim3(256), 0, st, (long long)us * 100, out); };
        seq(s[0]); seq(s[1]); hipDeviceSynchronize();
        double t0 = now(); seq(s[0]); hipStreamSynchronize(s[0]); const double a = now() - t0;
        t0 = now(); seq(s[1]); hipStreamSynchronize(s[1]); const double b = now() - t0;
        t0 = now(); seq(s[0]); seq(s[1]); hipStreamSynchronize(s[0]); hipStreamSynchronize(s[1]); const double both = now() - t0;
        t0 = now(); seq(s[0]); seq(s[2]); hipStreamSynchronize(s[0]); hipStreamSynchronize(s[2]); const double both2 = now() - t0;
        printf("%4d kernels of %4d us on %3d workgroup(s): alone %.2f / %.2f ms, two streams (priorities -1 / 0) %.2f ms, (-1 / 1) %.2f ms = %.2f x one sequence\n", n, us, grid, a, b, both, both2, both / a);
    }
    return 0;
}
