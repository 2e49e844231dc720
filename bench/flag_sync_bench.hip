// Cross-workgroup hand-off latency inside ONE kernel on gfx950 (8 XCDs, one L2 each): a chain of workgroups, each waits for its predecessor's
// flag (agent-scope acquire), reads the predecessor's 512-double payload, adds 1, publishes its own payload and flag (agent-scope release).
// Reports microseconds per hop and verifies the payload (a stale L2 line would show up as a wrong sum).  Variants: dirty = bytes of unrelated
// global stores each workgroup does before its release (what a release has to write back from the XCD's L2).
//   hipcc -O3 --offload-arch=gfx950 bench/flag_sync_bench.hip -o /tmp/flag_sync && /tmp/flag_sync
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_chain(int* flags, double* payload, double* scratch, int hops_per_wg, int dirty_doubles, int nwg) {
    const int w = blockIdx.x, t = threadIdx.x;
    for (int it = 0; it < hops_per_wg; ++it) {
        const int hop = it * nwg + w;                       // global hop index; predecessor is hop-1
        if (hop > 0) {
            const int pw = (w + nwg - 1) % nwg;
            if (t == 0) { while (__hip_atomic_load(&flags[pw], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < hop) { __builtin_amdgcn_s_sleep(1); } }
            __syncthreads();
            __atomic_thread_fence(__ATOMIC_ACQUIRE);        // every wave of the workgroup needs the invalidate, not just wave 0
            double a = payload[(size_t)pw * 512 + t], b = payload[(size_t)pw * 512 + 256 + t];
            payload[(size_t)w * 512 + t] = a + 1.0; payload[(size_t)w * 512 + 256 + t] = b + 1.0;
        } else { payload[(size_t)w * 512 + t] = 1.0; payload[(size_t)w * 512 + 256 + t] = 1.0; }
        for (int i = t; i < dirty_doubles; i += 256) scratch[(size_t)w * dirty_doubles + i] = (double)hop;
        __atomic_thread_fence(__ATOMIC_RELEASE);            // (agent scope by default for __atomic_thread_fence in HIP device code)
        __syncthreads();
        if (t == 0) __hip_atomic_store(&flags[w], hop + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main() {
    int* flags; double *payload, *scratch;
    const int NWG = 256;
    CK(hipMalloc(&flags, NWG * sizeof(int))); CK(hipMalloc(&payload, NWG * 512 * sizeof(double)));
    CK(hipMalloc(&scratch, (size_t)NWG * (1 << 17) * sizeof(double)));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int nwgs[] = {2, 8, 64, 256}; const int dirties[] = {0, 1 << 12, 1 << 17};
    for (int nwg : nwgs) for (int dirty : dirties) {
        const int hops = nwg <= 8 ? 2000 : (nwg == 64 ? 100 : 25);
        float best = 1e30f; bool ok = true;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(flags, 0, NWG * sizeof(int))); CK(hipMemset(payload, 0, NWG * 512 * sizeof(double)));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_chain, dim3(nwg), dim3(256), 0, 0, flags, payload, scratch, hops, dirty, nwg);
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
            std::vector<double> h(512); CK(hipMemcpy(h.data(), payload + (size_t)(nwg - 1) * 512, 512 * sizeof(double), hipMemcpyDeviceToHost));
            for (double v : h) if (v != (double)(hops * nwg)) ok = false;
        }
        printf("workgroups %3d  dirty %7d B/hop : %.2f us per hop   payload %s\n", nwg, dirty * 8, best * 1e3 / (hops * nwg), ok ? "ok" : "WRONG");
    }
    return 0;
}
