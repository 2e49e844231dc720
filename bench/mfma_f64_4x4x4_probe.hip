// mfma_f64_4x4x4_probe.hip — finds the lane layout of v_mfma_f64_4x4x4_f64 (4 blocks of 4x4x4) empirically: A one-hot in lane la, B one-hot in
// lane lb, prints every (la, lb, ld) with a non-zero result in lane ld.  (The CDNA guides give the layouts of the 16x16x4 form only.)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int la, int lb, double* out) {
    const int l = threadIdx.x;
    const double a = (l == la) ? 1.0 : 0.0, b = (l == lb) ? 1.0 : 0.0;
    out[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
}
int main() {
    double* d; hipMalloc(&d, 64 * sizeof(double));
    double h[64];
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            k<<<1, 64>>>(la, lb, d);
            hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
            for (int l = 0; l < 64; ++l) if (h[l] != 0.0) printf("%d %d %d\n", la, lb, l);
        }
    return 0;
}
