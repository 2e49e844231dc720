// lfac_item_bench.hip — how should one workgroup form a 64 x 64 tile of [gx; hx]' Omega [gx; hx] (a SCHUR item of csrc/lfac.hip)?  One tile per workgroup, 255 workgroups, 79 stages of
// 32 constraint rows, random operands (the matrix cores' clock depends on the data: profiles/r02_schur_loop_bench.txt), operands through LDS as in the library.
// Variants of who computes: MW wavefronts of the 16 hold ACC 16 x 16 accumulators each (MW * ACC = 16); every wavefront stages operands.
//   hipcc -O3 --offload-arch=gfx950 bench/lfac_item_bench.hip -o /tmp/lfac_item_bench && /tmp/lfac_item_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
constexpr int KT = 32, LDK = 34, TT = 64;
__device__ __forceinline__ void lds_barrier() { __builtin_amdgcn_s_waitcnt(0xc07f); __builtin_amdgcn_s_barrier(); }

// MW = 16: wave (wr, wc) = (w >> 2, w & 3), one tile.  MW = 8: wave w < 8: row strip w >> 1, column tiles 2 (w & 1) .. + 1.  MW = 4: wave w < 4: row strip w, all four column tiles.
template <int MW, bool PIPE>
__global__ __launch_bounds__(1024) void k_tile(int stages, const double* __restrict__ Zm, int ld, int ncol, double* __restrict__ out, int skew = 0, int nstall = 78) {
    __shared__ double smem[4 * TT * LDK];
    constexpr int ACC = 16 / MW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fk = lane >> 4;
    const int i0 = (int)((blockIdx.x * 64) % (ncol - 64)), j0 = (int)((blockIdx.x * 192 + 64) % (ncol - 64));
    const int k = tid & 31, c = tid >> 5;
    double ra[2], rb[2];
    const int sk = skew ? (int)((blockIdx.x * 7u) % (unsigned)nstall) : 0;
    auto fetch = [&](int st0) {
        const int st = (st0 + sk) % nstall;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            ra[q] = Zm[(st * KT + k) + (size_t)(i0 + c + 32 * q) * ld];
            rb[q] = 0.37 * Zm[(st * KT + k) + (size_t)(j0 + c + 32 * q) * ld];
        }
    };
    auto park = [&](int buf) {
        double* As = smem + (size_t)buf * 2 * TT * LDK;
        double* Bs = As + TT * LDK;
#pragma unroll
        for (int q = 0; q < 2; ++q) { As[(c + 32 * q) * LDK + k] = ra[q]; Bs[(c + 32 * q) * LDK + k] = rb[q]; }
    };
    int wr, wc0;
    if (MW == 16) { wr = wave >> 2; wc0 = wave & 3; }
    else if (MW == 8) { wr = wave >> 1; wc0 = 2 * (wave & 1); }
    else { wr = wave; wc0 = 0; }
    const bool mf = wave < MW;
    v4d acc[ACC];
#pragma unroll
    for (int n = 0; n < ACC; ++n) acc[n] = (v4d){0.0, 0.0, 0.0, 0.0};
    fetch(0); park(0); fetch(1);
    lds_barrier();
#pragma unroll 1
    for (int st = 0; st < stages; ++st) {
        const int cur = st & 1;
        const double* As = smem + (size_t)cur * 2 * TT * LDK;
        const double* Bs = As + TT * LDK;
        if (st + 1 < stages) park(cur ^ 1);
        if (st + 2 < stages) fetch(st + 2);
        if (mf) {
            const unsigned ab = (unsigned)(uintptr_t)(As + (wr * 16 + fr) * LDK + fk);
            const unsigned bb = (unsigned)(uintptr_t)(Bs + (wc0 * 16 + fr) * LDK + fk);
            if (!PIPE) {
                double fa[8], fb[8][ACC];
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(fa[kk]) : "v"(ab), "n"(kk * 32) : "memory");
#pragma unroll
                    for (int n = 0; n < ACC; ++n) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(fb[kk][n]) : "v"(bb + n * 16 * LDK * 8), "n"(kk * 32) : "memory");
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int kk = 0; kk < 8; ++kk)
#pragma unroll
                    for (int n = 0; n < ACC; ++n) acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[kk][n], fa[kk], acc[n], 0, 0, 0);
            } else {
                // reads one k-step ahead of the matrix instructions
                double fa[2], fb[2][ACC];
                auto issue = [&](int kk, int buf) {
                    asm volatile("ds_read_b64 %0, %1" : "=v"(fa[buf]) : "v"(ab + kk * 32u) : "memory");
#pragma unroll
                    for (int n = 0; n < ACC; ++n) asm volatile("ds_read_b64 %0, %1" : "=v"(fb[buf][n]) : "v"(bb + n * 16 * LDK * 8 + kk * 32u) : "memory");
                };
                issue(0, 0);
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    const int cb = kk & 1;
                    if (kk + 1 < 8) {
                        issue(kk + 1, cb ^ 1);
                        if (ACC == 1) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fa[cb]), "+v"(fb[cb][0]) :: "memory");
                        else if (ACC == 2) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fa[cb]), "+v"(fb[cb][0]), "+v"(fb[cb][ACC - 1]) :: "memory");
                        else asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(fa[cb]), "+v"(fb[cb][0]), "+v"(fb[cb][1]), "+v"(fb[cb][ACC - 2]), "+v"(fb[cb][ACC - 1]) :: "memory");
                    } else {
                        if (ACC == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[cb]), "+v"(fb[cb][0]) :: "memory");
                        else if (ACC == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[cb]), "+v"(fb[cb][0]), "+v"(fb[cb][ACC - 1]) :: "memory");
                        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[cb]), "+v"(fb[cb][0]), "+v"(fb[cb][1]), "+v"(fb[cb][ACC - 2]), "+v"(fb[cb][ACC - 1]) :: "memory");
                    }
#pragma unroll
                    for (int n = 0; n < ACC; ++n) acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[cb][n], fa[cb], acc[n], 0, 0, 0);
                }
            }
        }
        lds_barrier();
    }
    if (mf) {
        double s = 0.0;
#pragma unroll
        for (int n = 0; n < ACC; ++n) for (int r = 0; r < 4; ++r) s += acc[n][r];
        out[(size_t)blockIdx.x * 1024 + tid] = s;
    }
}


// Ring of THREE LDS stage buffers: the operands of stage s + 2 are parked during stage s, so the fragments of stage s + 1 (parked during stage s - 1, visible since the
// barrier that ended it) can be read BEFORE the barrier that ends stage s — the matrix instructions of the next stage start right behind the barrier instead of behind a
// round of LDS reads.  16 wavefronts x 1 accumulator, the arithmetic per entry unchanged.
__global__ __launch_bounds__(1024) void k_tile_ring3(int stages, const double* __restrict__ Zm, int ld, int ncol, double* __restrict__ out) {
    __shared__ double smem[6 * TT * LDK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, fr = lane & 15, fk = lane >> 4;
    const int i0 = (int)((blockIdx.x * 64) % (ncol - 64)), j0 = (int)((blockIdx.x * 192 + 64) % (ncol - 64));
    const int k = tid & 31, c = tid >> 5;
    double ra[2][2], rb[2][2];
    auto fetch = [&](int st, int set) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            ra[set][q] = Zm[(st * KT + k) + (size_t)(i0 + c + 32 * q) * ld];
            rb[set][q] = Zm[(st * KT + k) + (size_t)(j0 + c + 32 * q) * ld];
        }
    };
    auto park = [&](int buf, int set) {
        double* As = smem + (size_t)buf * 2 * TT * LDK;
        double* Bs = As + TT * LDK;
#pragma unroll
        for (int q = 0; q < 2; ++q) { As[(c + 32 * q) * LDK + k] = ra[set][q]; Bs[(c + 32 * q) * LDK + k] = 0.37 * rb[set][q]; }
    };
    const int wr = wave >> 2, wc = wave & 3;
    v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
    // prologue: stages 0, 1 parked, stages 2, 3 in flight
    fetch(0, 0); fetch(1, 1);
    park(0, 0); park(1, 1);
    fetch(2, 0); fetch(3, 1);
    lds_barrier();
    double fa[8], fb[8];
    auto reads = [&](int buf, int k0, int k1) {
        const unsigned ab = (unsigned)(uintptr_t)(smem + (size_t)buf * 2 * TT * LDK + (wr * 16 + fr) * LDK + fk);
        const unsigned bb = (unsigned)(uintptr_t)(smem + (size_t)buf * 2 * TT * LDK + TT * LDK + (wc * 16 + fr) * LDK + fk);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) if (kk >= k0 && kk < k1) {
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(fa[kk]) : "v"(ab), "n"(kk * 32) : "memory");
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(fb[kk]) : "v"(bb), "n"(kk * 32) : "memory");
        }
    };
    reads(0, 0, 4);
#pragma unroll 1
    for (int st = 0; st < stages; st += 6) {          // unrolled by 6: buffer (mod 3) and register set (mod 2) static
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int s = st + u;
            if (s >= stages) break;
            const int buf = u % 3, set = u & 1;
            // the second half of this stage's fragments; the first half arrived before the barrier
            reads(buf, 4, 8);
            asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(fa[0]), "+v"(fb[0]), "+v"(fa[1]), "+v"(fb[1]), "+v"(fa[2]), "+v"(fb[2]), "+v"(fa[3]), "+v"(fb[3]) :: "memory");
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[kk], fa[kk], acc, 0, 0, 0);
            if (s + 2 < stages) park((u + 2) % 3, set);                 // stage s + 2 (fetched two stages ago) into the buffer stage s - 1 used
            if (s + 4 < stages) fetch(s + 4, set);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[4]), "+v"(fb[4]), "+v"(fa[5]), "+v"(fb[5]), "+v"(fa[6]), "+v"(fb[6]), "+v"(fa[7]), "+v"(fb[7]) :: "memory");
            if (s + 1 < stages) reads((u + 1) % 3, 0, 4);               // next stage's first fragments: its buffer is complete since the previous barrier
#pragma unroll
            for (int kk = 4; kk < 8; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[kk], fa[kk], acc, 0, 0, 0);
            lds_barrier();
        }
    }
    double s = 0.0;
    for (int r = 0; r < 4; ++r) s += acc[r];
    out[(size_t)blockIdx.x * 1024 + tid] = s;
}

template <int MW, bool PIPE> void run(const char* name, const double* Zm, int ld, int ncol, double* out) {
    const int blocks = 255, stages = 78;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_tile<MW, PIPE><<<blocks, 1024>>>(stages, Zm, ld, ncol, out); hipDeviceSynchronize();
    const int reps = 5;
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) k_tile<MW, PIPE><<<blocks, 1024>>>(stages, Zm, ld, ncol, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0.f; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double flops = (double)blocks * stages * 2.0 * 64 * 64 * 32;
    printf("%-58s %8.1f us/launch  %6.3f us/stage  %6.2f TFLOP/s\n", name, ms * 1e3, ms * 1e3 / stages, flops / (ms * 1e-3) / 1e12);
}

int main() {
    const int m = 2500, ncol = 2500;
    std::vector<double> h((size_t)m * ncol);
    unsigned long long x = 88172645463325252ull;
    for (auto& v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (double)(x >> 11) / 9007199254740992.0 * 2.0 - 1.0; }
    double *Zm, *out;
    hipMalloc(&Zm, h.size() * 8); hipMemcpy(Zm, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipMalloc(&out, 255 * 1024 * 8);
    for (int rep = 0; rep < 2; ++rep) {
        run<16, false>("16 wavefronts x 1 accumulator, reads up front (lfac v1)", Zm, m, ncol, out);
        run<16, true>("16 wavefronts x 1 accumulator, reads one k-step ahead", Zm, m, ncol, out);
        run<8, false>(" 8 wavefronts x 2 accumulators, reads up front", Zm, m, ncol, out);
        run<8, true>(" 8 wavefronts x 2 accumulators, reads one k-step ahead", Zm, m, ncol, out);
        run<4, false>(" 4 wavefronts x 4 accumulators, reads up front", Zm, m, ncol, out);
        run<4, true>(" 4 wavefronts x 4 accumulators, reads one k-step ahead", Zm, m, ncol, out);
    }
    {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int stages : {78, 12}) {
            k_tile_ring3<<<255, 1024>>>(stages, Zm, m, ncol, out); hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int r = 0; r < 5; ++r) k_tile_ring3<<<255, 1024>>>(stages, Zm, m, ncol, out);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = 0.f; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
            printf("ring of three stage buffers, next stage's fragments read before the barrier, %2d stages: %7.1f us/launch  %6.3f us/stage\n", stages, ms * 1e3, ms * 1e3 / stages);
        }
    }
    // every workgroup at a stage of its own (what slices of tiles that started at different times look like): the operand rows are no longer shared through the L2s
    {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int skew = 0; skew < 2; ++skew) for (int stages : {78, 10}) {
            k_tile<16, false><<<255, 1024>>>(stages, Zm, m, ncol, out, skew, 78); hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int r = 0; r < 5; ++r) k_tile<16, false><<<255, 1024>>>(stages, Zm, m, ncol, out, skew, 78);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms = 0.f; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
            printf("16 x 1, %2d stages per launch, %s: %7.1f us/launch  %6.3f us/stage\n", stages, skew ? "every workgroup at its own stage" : "all workgroups in step           ", ms * 1e3, ms * 1e3 / stages);
        }
    }
    return 0;
}
