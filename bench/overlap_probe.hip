// overlap_probe.hip — can an HBM-bound kernel of one stream run INSIDE the compute units an fp64-MFMA-bound kernel of another stream occupies?
// The dense batched path alternates MFMA-bound phases (k_schur, k_ldl_step: 1024-thread workgroups, ~136 KB of LDS, one per compute unit) and HBM-bound ones
// (the mat-vecs of the solves and refinement residuals); with several groups in flight the phases of different groups could overlap if the hardware co-schedules
// them on a compute unit.  This probe measures it with stand-ins whose resources are set by template parameters:
//   A<WAVES_PER_EU>: 1024 threads, LDS_A bytes of dynamic LDS, back-to-back v_mfma_f64_16x16x4 on 8 accumulators, register budget from amdgpu_waves_per_eu
//   B: 256 threads, streams a buffer (NLOAD independent 8-byte loads per thread in flight), register budget <= 128
// and reports A alone, B alone, A and B queued on two streams at once (wall time of both), against the sum and the maximum.
//   hipcc --offload-arch=gfx950 -O3 bench/overlap_probe.hip -o /tmp/overlap_probe && /tmp/overlap_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int WPE>
__global__ __attribute__((amdgpu_flat_work_group_size(1024, 1024), amdgpu_waves_per_eu(WPE, WPE))) void kA(int iters, double* out) {
    extern __shared__ double lds[];
    v4d acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (v4d){0, 0, 0, 0};
    lds[threadIdx.x] = threadIdx.x * 1e-3;
    __syncthreads();
    double a = lds[(threadIdx.x * 7) & 1023], b = blockIdx.x * 1e-3 + 1.0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NLOAD>
__global__ __launch_bounds__(256) void kB(const double* __restrict__ src, size_t n, double* out) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    double s = 0;
    for (; i + (NLOAD - 1) * stride < n; i += NLOAD * stride) {
        double v[NLOAD];
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) v[k] = __builtin_nontemporal_load(src + i + k * stride);
#pragma unroll
        for (int k = 0; k < NLOAD; ++k) s += v[k];
    }
    if (s == 1.2345e-300) out[0] = s;
}

static float ms_between(hipEvent_t a, hipEvent_t b) { float ms; hipEventElapsedTime(&ms, a, b); return ms; }

static hipStream_t g_streams[8];
static int g_sa = 0, g_sb = 1, g_gridA = 256;
template <int WPE>
void probe(int ldsA, int itersA, const double* src, size_t n, double* out, int gridB) {
    hipStream_t sa = g_streams[g_sa], sb = g_streams[g_sb];
    hipFuncSetAttribute((const void*)kA<WPE>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsA);
    hipEvent_t e[6];
    for (auto& x : e) hipEventCreate(&x);
    auto runA = [&](hipStream_t st) { hipLaunchKernelGGL(kA<WPE>, dim3(g_gridA), dim3(1024), ldsA, st, itersA, out); };
    auto runB = [&](hipStream_t st) { for (int r = 0; r < 8; ++r) hipLaunchKernelGGL(kB<16>, dim3(gridB), dim3(256), 0, st, src, n, out + (1 << 20)); };
    runA(sa); runB(sb); hipDeviceSynchronize();
    hipEventRecord(e[0], sa); runA(sa); hipEventRecord(e[1], sa); hipDeviceSynchronize();
    hipEventRecord(e[2], sb); runB(sb); hipEventRecord(e[3], sb); hipDeviceSynchronize();
    const float a = ms_between(e[0], e[1]), b = ms_between(e[2], e[3]);
    // both at once: host wall clock around the two queues (no events between the launches: an event record is a packet of its own in the queue), then the per-stream
    // event times of a second run
    hipDeviceSynchronize();
    const auto h0 = std::chrono::steady_clock::now();
    runA(sa); runB(sb);
    hipStreamSynchronize(sa); hipStreamSynchronize(sb);
    const double wall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h0).count();
    const auto h1 = std::chrono::steady_clock::now();
    runA(sa); hipStreamSynchronize(sa); runB(sb); hipStreamSynchronize(sb);
    const double wall_seq = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - h1).count();
    hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void*)kA<WPE>);
    hipFuncAttributes fb; hipFuncGetAttributes(&fb, (const void*)kB<16>);
    printf("streams %d/%d gridA %3d | A: waves/EU %d (%3d VGPRs), LDS %6d B | B: %d VGPRs, grid %5d | A alone %.3f ms, B alone %.3f ms (%.2f TB/s) | one after the other %.3f ms, both queued at once %.3f ms = %.2f x max, %.2f x sum\n",
           g_sa, g_sb, g_gridA, WPE, fa.numRegs, ldsA, fb.numRegs, gridB, a, b, 8.0 * n * 8 / b * 1e-9, wall_seq, wall, wall / (a > b ? a : b), wall / (a + b));
}

int main() {
    const size_t n = (size_t)1 << 27;      // 1 GiB of doubles: far beyond the Infinity Cache
    double *src, *out;
    hipMalloc(&src, n * 8); hipMalloc(&out, sizeof(double) * ((1 << 20) + 1024));
    hipMemset(src, 0, n * 8);
    // (streams of ONE priority share a hardware queue on this system: kernels of two such streams never overlap — the first runs of this probe, and why the library's
    // lanes are one per priority class; streams 0..2 = the three classes, 3..7 = default priority)
    int least = 0, greatest = 0;
    hipDeviceGetStreamPriorityRange(&least, &greatest);
    printf("stream priorities: least %d greatest %d\n", least, greatest);
    for (int i = 0; i < 8; ++i) {
        if (i < 3) hipStreamCreateWithPriority(&g_streams[i], hipStreamNonBlocking, greatest + i > least ? least : greatest + i);
        else hipStreamCreateWithFlags(&g_streams[i], hipStreamNonBlocking);
    }
    // (1) which pairs of streams sit on different hardware queues?  A on half the compute units: B must overlap if the queues differ
    // (0) A on ONE, 8, 64 compute units: does B get the rest of the chip?
    for (int ga : {1, 8, 64}) { g_gridA = ga; g_sa = 0; g_sb = 1; probe<5>(64 * 1024, 5000, src, n, out, 512); g_sa = 0; g_sb = 2; probe<5>(64 * 1024, 5000, src, n, out, 512); }
    g_gridA = 128;
    for (int a : {0, 1, 2}) for (int b : {0, 1, 2, 3}) { if (a == b) continue; g_sa = a; g_sb = b; probe<5>(64 * 1024, 5000, src, n, out, 512); }
    // (2) A on every compute unit
    g_gridA = 256;
    for (int a : {0, 2}) for (int b : {0, 1, 2}) {
        if (a == b) continue;
        g_sa = a; g_sb = b;
        probe<4>(136 * 1024, 5000, src, n, out, 512);     // A fills the registers of a SIMD: 4 waves x 128
        probe<5>(136 * 1024, 5000, src, n, out, 512);     // A capped at 96 VGPRs: 128 left per SIMD lane
        probe<5>(64 * 1024, 5000, src, n, out, 512);
    }
    return 0;
}
