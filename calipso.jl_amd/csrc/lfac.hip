// lfac.hip — the factorisation of ONE dense system with the Schur complement's products folded into the panel launches ("left-looking" schedule).
//
// What it replaces: schur.hip's k_schur (S = Lxx + ep I + [gx; hx]' Omega [gx; hx], 0.42 ms at C3 with nothing else running) followed by ldl.hip's right-looking
// panel steps (a pivot chain of 40 launches, 0.83 ms, during which 255 of the 256 compute units have little to do).  Together they are factorize! of
// linear_solver.jl:19-31 / QDLDL_factor! of qdldl.jl:400-589 for the condensed matrix of residual_jacobian_variables.jl:110-167 in the order [z | y | x].
//
// Every 64 x 64 tile (i, j) of the lower triangle of S is the end of ONE sequence of operations that no schedule may reorder (the bits of a group member, which
// takes k_schur + k_ldl_step, must be the bits of the same handle stepped alone):
//     acc = 0;  acc += [gx; hx](:, i)' Omega [gx; hx](:, j)  stage by stage (32 constraint rows per stage, equality rows first: k_schur's order);
//     S(i, j) = acc + Lsym(i, j) (+ ep on the diagonal);
//     S(i, j) -= Z_p(i) A(j, p)'  for the panels p = 0, 1, ..., j - 1 in turn, Z_p(i) = A(i, p) M_p (ldl.hip), each product rounded on its own.
// WHEN a segment of that sequence runs is free as long as the order per tile is kept and a panel is applied after it has been factored.  Here:
//   * launch k (one per panel, as before) has ONE workgroup that carries the pivot chain — tile (k+1, k+1) -= Z_k A', then the 64 pivots of diagonal block k + 1,
//     X, M_{k+1}: ldl_device.hpp, the code of k_ldl_step's tile 0 — and 255 workers that walk item lists the host planned once per shape:
//       ROW   (row i > k + 1): Z_k(i) = A(i, k) M_k, kept in Zbuf for later; tile (i, k+1) gets its last panel; tile (i, k+2) whatever panels it still lacks, then k
//       FAR   (tile (i, j), j > k + 2): the panels [a, b), b <= k, from Zbuf(i, p) and the raw columns A(j, p) — no M, no second product
//       SCHUR (tile (i, j), stages [a, b)): a slice of the constraint products; the accumulator rests in S between slices (a store and a load change no bit)
//   * the planner (earliest deadline first: column j is consumed by launch j - 2) fills every launch up to the duration of the chain's workgroup, so the
//     1.56e10 flop of the Schur complement and the deferred trailing updates run UNDER the pivot chain instead of in front of it;
//   * launch -2 (the head) holds the slices that must precede the first pivot, launch -1 factors diagonal block 0.
// The raw panel columns stay in S to the end (deferred updates read them), so the factor columns L = A X' D^-1 go to a buffer of their own (Lf), which is what the
// finish (k_ldl_scale, the inverse merges, the W-form products: ldl.hip) and the solves read; the finish of completed column ranges still runs on the second
// stream beside the chain.
#include "internal.hpp"
#include "ldl_device.hpp"

#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <map>
#include <memory>
#include <mutex>
#include <queue>
#include <thread>
#include <type_traits>

namespace calipso {

constexpr int LKT = 32;            // constraint rows per Schur stage (schur.hip: KT — the stage boundaries are part of the arithmetic)
constexpr int LLDK = LKT + 2;
constexpr int LFAC_LDS_DOUBLES = 4 * TT * LDT > DIAG_LDS_DOUBLES ? 4 * TT * LDT : DIAG_LDS_DOUBLES;     // two operand pairs of the deferred updates | the diagonal block | four Schur stage buffers
static_assert(6 * TT * LLDK <= LFAC_LDS_DOUBLES, "Schur stage buffers (a ring of three)");

enum { LI_SCHUR = 0, LI_FAR = 1, LI_ROW = 2 };
struct LItem { short kind, i, j, a, b, c, pad0, pad1; };     // SCHUR: tile (i, j), stages [a, b), c bit 0: first slice, bit 1: last (epilogue); FAR: tile (i, j), panels [a, b); ROW: row i, panel j, pending panels of tile (i, j + 2) from a

struct LfacArgs {
    int NP, nx, m, ne, nc, tb;
    int k;                  // launch: -2 head, -1 diagonal block 0, k >= 0 panel k
    int nst0, nst;          // Schur stages: equality rows, all
    double *S, *Zbuf, *Minv, *Dx, *Tinv;
    const double *Lsym, *Zj, *WH;      // Zj = [gx; hx] (leading dimension m), WH = Omega_z hx (leading dimension nc)
    int* icount;
    const LItem* items; const int* wfirst;
    unsigned long long* hprog; unsigned long long ptag;
    Scalars sc;
};

// ---- worker items ----------------------------------------------------------------------------------------------------------------------
struct LLane { int tid, lane, wave, wr, wc, fr, fk, row, cb; };

// a slice of the constraint products of tile (i, j): k_schur's arithmetic per entry (one accumulator, stages in order, 8 matrix instructions of 4 rows per stage,
// the column side scaled on the way to LDS), a 64 x 64 tile per workgroup (one 16 x 16 MFMA tile per wavefront).  it.pad0 = P > 1: the workgroup forms only the
// 64 / P rows [it.pad1 * 64 / P, ...) of the tile with its first 16 / P wavefronts (one or two per SIMD instead of four: a tile that the pivot chain is waiting
// for walks its 79 stages two to four times faster on P compute units).  Addresses: a wave-uniform base per stage + a per-thread 32-bit offset that is constant over
// the stages; the bounds predicates only where a stage or the tile touches an edge (the vector ALU work per stage must stay a handful of instructions: 16 wavefronts
// share four SIMDs with the matrix instructions).
__device__ __forceinline__ void item_schur(const LfacArgs& a, const LItem it, const LLane& L, double* __restrict__ smem) {
    const int i0 = it.i * TT, j0 = it.j * TT;
    const int P = it.pad0 > 1 ? it.pad0 : 1;
    const bool mf = __builtin_amdgcn_readfirstlane(L.wave) < 16 / P;      // this wavefront holds an accumulator tile (wave-uniform: a scalar branch, not an exec mask around the matrix instructions)
    const int wr = (4 / P) * it.pad1 + (L.wave >> 2);             // its row tile
    const int offC = (wr * 16 + L.fr) + (L.wc * 16 + L.fk) * a.NP;
    double* T = a.S + (i0 + (size_t)j0 * a.NP);
    v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
    if (mf && !(it.c & 1)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = T[offC + 4 * r * a.NP];
    }
    // the accumulator has ARRIVED before the stage loop starts: left pending, its first use inside the loop makes the compiler wait for every outstanding load in every
    // stage (s_waitcnt vmcnt(0) in front of the first matrix instruction) — the operands just requested for two stages ahead included, 1.4 us per stage
    asm volatile("" : "+v"(acc));
    const double omega_y = -1.0 / (-1.0 / (a.sc.rho + a.sc.ep) + (0.0 - a.sc.ed));
    const int k = L.tid & (LKT - 1), c = L.tid >> 5;          // lanes along k (contiguous in the column-major Jacobians), 32 columns per pass, two passes
    // BRANCH-FREE requests: rows beyond the constraint block and columns beyond nx are clamped to an address inside the matrix and replaced by 0.0 when the stage is
    // parked (a select, not a product: the operands are those of k_schur to the bit).  With branches around the loads the compiler waited for every outstanding
    // load wherever paths met — the operands requested for two stages ahead included, which put the memory latency into every stage.
    const bool vA0 = i0 + c < a.nx, vA1 = i0 + c + 32 < a.nx, vB0 = j0 + c < a.nx, vB1 = j0 + c + 32 < a.nx;
    const unsigned cA0 = (unsigned)min(i0 + c, a.nx - 1), cA1 = (unsigned)min(i0 + c + 32, a.nx - 1), cB0 = (unsigned)min(j0 + c, a.nx - 1), cB1 = (unsigned)min(j0 + c + 32, a.nx - 1);
    const unsigned pA0 = cA0 * (unsigned)a.m, pA1 = cA1 * (unsigned)a.m, pBm0 = cB0 * (unsigned)a.m, pBm1 = cB1 * (unsigned)a.m, pBc0 = cB0 * (unsigned)a.nc, pBc1 = cB1 * (unsigned)a.nc;
    double ra[2][2], rb[2][2];        // operands in flight for TWO stages ahead (two register sets, by the parity of the stage relative to the slice's first)
    auto fetch = [&](int st, int set) {
        const bool eq = st < a.nst0;
        const int kl = (eq ? st : st - a.nst0) * LKT, klim = (eq ? a.ne : a.nc) - 1 - kl;
        const unsigned kk = (unsigned)min(k, klim);
        const double* bA = (eq ? a.Zj : a.Zj + a.ne) + kl;
        const double* bB = (eq ? a.Zj : a.WH) + kl;
        ra[set][0] = bA[kk + pA0]; ra[set][1] = bA[kk + pA1];
        rb[set][0] = bB[kk + (eq ? pBm0 : pBc0)]; rb[set][1] = bB[kk + (eq ? pBm1 : pBc1)];
    };
    auto park = [&](int buf, int set, int st) {      // st: the stage the set holds (its row bound and the scale of the column side)
        const bool eq = st < a.nst0;
        const int klim = (eq ? a.ne : a.nc) - 1 - (eq ? st : st - a.nst0) * LKT;
        const bool kin = k <= klim;
        const double scale = eq ? omega_y : 1.0;
        double* As = smem + (size_t)buf * 2 * TT * LLDK;
        double* Bs = As + TT * LLDK;
        As[c * LLDK + k] = (kin && vA0) ? 1.0 * ra[set][0] : 0.0;
        As[(c + 32) * LLDK + k] = (kin && vA1) ? 1.0 * ra[set][1] : 0.0;
        Bs[c * LLDK + k] = (kin && vB0) ? scale * rb[set][0] : 0.0;
        Bs[(c + 32) * LLDK + k] = (kin && vB1) ? scale * rb[set][1] : 0.0;
    };
    // A ring of THREE stage buffers in LDS and two register sets: stage s + 2 is parked while stage s is computed (its operands were requested at stage s - 2), so the
    // first half of the fragments of stage s + 1 — parked during stage s - 1, visible since the barrier that ended it — is read BEFORE the barrier that ends stage s,
    // under the last matrix instructions of the stage: the next stage's matrix instructions start right behind the barrier instead of behind a round of LDS reads
    // (bench/lfac_item_bench.hip: 1.63 -> 1.43 us per stage).  The arithmetic per entry is untouched.
    const int s0 = it.a, s1 = it.b;
    double fa[8], fb[8];
    auto reads = [&](int buf, int half) {          // fragments 4 half .. 4 half + 3 of the stage in buffer buf
        const unsigned ab = (unsigned)(uintptr_t)(smem + (size_t)buf * 2 * TT * LLDK + (wr * 16 + L.fr) * LLDK + L.fk);
        const unsigned bb = (unsigned)(uintptr_t)(smem + (size_t)buf * 2 * TT * LLDK + TT * LLDK + (L.wc * 16 + L.fr) * LLDK + L.fk);
        if (half == 0) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(fa[kk]) : "v"(ab), "n"(kk * 32) : "memory");
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(fb[kk]) : "v"(bb), "n"(kk * 32) : "memory");
            }
        } else {
#pragma unroll
            for (int kk = 4; kk < 8; ++kk) {
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(fa[kk]) : "v"(ab), "n"(kk * 32) : "memory");
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(fb[kk]) : "v"(bb), "n"(kk * 32) : "memory");
            }
        }
    };
    if (s1 > s0) {
        fetch(s0, 0);
        if (s0 + 1 < s1) fetch(s0 + 1, 1);
        park(0, 0, s0);
        if (s0 + 1 < s1) park(1, 1, s0 + 1);
        if (s0 + 2 < s1) fetch(s0 + 2, 0);
        if (s0 + 3 < s1) fetch(s0 + 3, 1);
    }
    lds_barrier();
    if (mf && s1 > s0) reads(0, 0);
    // one stage; u = (stage - s0) mod 6 picks the buffer (u mod 3) and the register set (u mod 2): the loop is unrolled by six so that both are static.  inner: at
    // least four more stages behind it — no conditions around its requests (the bulk of the loop is straight-line code)
    auto stage = [&](int st, auto uc, auto inner) {
        constexpr int u = decltype(uc)::value, buf = u % 3, set = u & 1;
        if (mf) {
            reads(buf, 1);
            asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(fa[0]), "+v"(fb[0]), "+v"(fa[1]), "+v"(fb[1]), "+v"(fa[2]), "+v"(fb[2]), "+v"(fa[3]), "+v"(fb[3]) :: "memory");
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[kk], fa[kk], acc, 0, 0, 0);
        }
        if (decltype(inner)::value || st + 2 < s1) park((u + 2) % 3, set, st + 2);      // (its buffer held stage st - 1: every wavefront is past the barrier behind that stage)
        if (decltype(inner)::value || st + 4 < s1) fetch(st + 4, set);
        if (mf) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[4]), "+v"(fb[4]), "+v"(fa[5]), "+v"(fb[5]), "+v"(fa[6]), "+v"(fb[6]), "+v"(fa[7]), "+v"(fb[7]) :: "memory");
            // (the first half's registers are free: the matrix instructions that read them issued long ago — the pattern of frag_product's ring)
            if (decltype(inner)::value || st + 1 < s1) reads((u + 1) % 3, 0);
#pragma unroll
            for (int kk = 4; kk < 8; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[kk], fa[kk], acc, 0, 0, 0);
        }
        lds_barrier();
    };
    int st = s0;
#pragma unroll 1
    for (; st + 9 < s1; st += 6) {
        stage(st, std::integral_constant<int, 0>(), std::true_type()); stage(st + 1, std::integral_constant<int, 1>(), std::true_type());
        stage(st + 2, std::integral_constant<int, 2>(), std::true_type()); stage(st + 3, std::integral_constant<int, 3>(), std::true_type());
        stage(st + 4, std::integral_constant<int, 4>(), std::true_type()); stage(st + 5, std::integral_constant<int, 5>(), std::true_type());
    }
#pragma unroll 1
    for (; st < s1; st += 6) {
        stage(st, std::integral_constant<int, 0>(), std::false_type());
        if (st + 1 < s1) stage(st + 1, std::integral_constant<int, 1>(), std::false_type());
        if (st + 2 < s1) stage(st + 2, std::integral_constant<int, 2>(), std::false_type());
        if (st + 3 < s1) stage(st + 3, std::integral_constant<int, 3>(), std::false_type());
        if (st + 4 < s1) stage(st + 4, std::integral_constant<int, 4>(), std::false_type());
        if (st + 5 < s1) stage(st + 5, std::integral_constant<int, 5>(), std::false_type());
    }
    if (!mf) return;
    if (it.c & 2) {
        // + Lxx through its upper triangle (triu(K), schur.hip: k_symmetrize_upper) + ep on the diagonal; unit pivots in the padding
        const int gi = i0 + wr * 16 + L.fr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gj = j0 + L.wc * 16 + L.fk + 4 * r;
            double v;
            if (gi < a.nx && gj < a.nx) {
                v = acc[r] + a.Lsym[gi + (size_t)gj * a.nx];
                if (gi == gj) v += a.sc.ep;
            } else {
                v = (gi == gj) ? 1.0 : 0.0;
            }
            acc[r] = v;
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) T[offC + 4 * r * a.NP] = acc[r];
}

// tile (ti, tj) -= sum over the panels [p0, p1) of Zbuf(ti, p) A(tj, p)', panel by panel (each product rounded on its own, subtracted in panel order); cS in registers
__device__ __forceinline__ void apply_panels(const LfacArgs& a, double (&cS)[4], int ti, int tj, int p0, int p1, const LLane& L, double* __restrict__ Zb0, double* __restrict__ Yb0,
                                             double* __restrict__ Zb1, double* __restrict__ Yb1) {
    if (p1 <= p0) return;
    const int offY = L.row + L.cb * a.NP;
    const double* Zg = a.Zbuf + ((size_t)ti * TT + (size_t)p0 * NB * a.NP);
    const double* Ag = a.S + ((size_t)tj * TT + (size_t)p0 * NB * a.NP);
    double zv[4], yv[4];
    asm volatile("" : "+v"(cS[0]), "+v"(cS[1]), "+v"(cS[2]), "+v"(cS[3]));      // (the tile has arrived: its first use inside the loop would wait for the operands in flight as well)
#pragma unroll
    for (int q = 0; q < 4; ++q) { zv[q] = Zg[offY + q * 16 * a.NP]; yv[q] = Ag[offY + q * 16 * a.NP]; }
#pragma unroll 1
    for (int p = p0; p < p1; ++p) {
        double* Zs = ((p - p0) & 1) ? Zb1 : Zb0;
        double* Ys = ((p - p0) & 1) ? Yb1 : Yb0;
#pragma unroll
        for (int q = 0; q < 4; ++q) { Zs[L.row * LDT + L.cb + q * 16] = zv[q]; Ys[L.row * LDT + L.cb + q * 16] = yv[q]; }
        lds_barrier();          // (the other pair is still being read by wavefronts that lag: that is what the second pair is for — one barrier per panel)
        if (p + 1 < p1) {
            Zg += (size_t)NB * a.NP; Ag += (size_t)NB * a.NP;
#pragma unroll
            for (int q = 0; q < 4; ++q) { zv[q] = Zg[offY + q * 16 * a.NP]; yv[q] = Ag[offY + q * 16 * a.NP]; }
        }
        const v4d acc = frag_product((unsigned)(uintptr_t)(Zs + (L.wr * 16 + L.fr) * LDT + L.fk), (unsigned)(uintptr_t)(Ys + (L.wc * 16 + L.fr) * LDT + L.fk));
#pragma unroll
        for (int r = 0; r < 4; ++r) cS[r] -= acc[r];
        if (Zb0 == Zb1 && p + 1 < p1) lds_barrier();      // a single pair: its readers are done before the next panel is parked
    }
    lds_barrier();              // every read of the operand pairs is done before the caller refills LDS
}

__device__ __forceinline__ void item_far(const LfacArgs& a, const LItem it, const LLane& L, double* __restrict__ smem) {
    const int offC = (L.wr * 16 + L.fr) + (L.wc * 16 + L.fk) * a.NP;
    double* T = a.S + ((size_t)it.i * TT + (size_t)it.j * TT * a.NP);
    double cS[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) cS[r] = T[offC + 4 * r * a.NP];
    apply_panels(a, cS, it.i, it.j, it.a, it.b, L, smem, smem + TT * LDT, smem + 2 * TT * LDT, smem + 3 * TT * LDT);
#pragma unroll
    for (int r = 0; r < 4; ++r) T[offC + 4 * r * a.NP] = cS[r];
}

// row i of panel k: Z_k(i) -> Zbuf; tile (i, k+1) -= Z_k(i) A(k+1, k)' (its last panel); tile (i, k+2): the pending panels [it.a, k) from Zbuf, then panel k
__device__ __forceinline__ void item_row(const LfacArgs& a, const LItem it, const LLane& L, double* __restrict__ smem) {
    double* Zs = smem;                     // Z_k(i): stays for the whole item
    double* Ys = smem + TT * LDT;
    double* Ms = smem + 2 * TT * LDT;
    double* Xs = smem + 3 * TT * LDT;
    const int i = it.i, k = it.j, k0 = k * NB;
    const int offC = (L.wr * 16 + L.fr) + (L.wc * 16 + L.fk) * a.NP, offY = L.row + L.cb * a.NP;
    // the tile that takes its last panel and the column operand of that product travel with the operands of Z
    double* T1 = a.S + ((size_t)i * TT + (size_t)(k + 1) * TT * a.NP);
    const double* A1 = a.S + ((size_t)(k + 1) * TT + (size_t)k0 * a.NP);
    // it.c: bit 0 — tile (i, k+2) too; bit 1 — ONLY that tile (the other half of a row item the planner has cut in two where workers idle: Z is formed by both
    // halves — the same operands, the same bits — and kept by the half that owns tile (i, k+1))
    const bool second = (it.c & 1) != 0, first = (it.c & 2) == 0;
    double c1[4], y1[4];
    if (first) {
#pragma unroll
        for (int r = 0; r < 4; ++r) c1[r] = T1[offC + 4 * r * a.NP];
#pragma unroll
        for (int q = 0; q < 4; ++q) y1[q] = A1[offY + q * 16 * a.NP];
    }
    v4d z;
    form_Z(a.S + ((size_t)i * TT + (size_t)k0 * a.NP), a.NP, a.Minv + (size_t)k * NB * NB, Ys, Ms, Zs, L.row, L.cb, L.wr, L.wc, L.fr, L.fk, &z);
    if (first) {
        double* Zg = a.Zbuf + ((size_t)i * TT + (size_t)k0 * a.NP);
#pragma unroll
        for (int r = 0; r < 4; ++r) Zg[offC + 4 * r * a.NP] = z[r];
#pragma unroll
        for (int q = 0; q < 4; ++q) Ys[L.row * LDT + L.cb + q * 16] = y1[q];
        lds_barrier();
    }
    double* T2 = a.S + ((size_t)i * TT + (size_t)(k + 2) * TT * a.NP);
    double c2[4], y2[4];
    if (second) {
        const double* A2 = a.S + ((size_t)(k + 2) * TT + (size_t)k0 * a.NP);
#pragma unroll
        for (int r = 0; r < 4; ++r) c2[r] = T2[offC + 4 * r * a.NP];
#pragma unroll
        for (int q = 0; q < 4; ++q) y2[q] = A2[offY + q * 16 * a.NP];
    }
    if (first) {
        const v4d acc = frag_product((unsigned)(uintptr_t)(Zs + (L.wr * 16 + L.fr) * LDT + L.fk), (unsigned)(uintptr_t)(Ys + (L.wc * 16 + L.fr) * LDT + L.fk));
#pragma unroll
        for (int r = 0; r < 4; ++r) T1[offC + 4 * r * a.NP] = c1[r] - acc[r];
    }
    if (!second) { lds_barrier(); return; }
    // the pending panels go through the pair (Ms, Xs), which nothing has read since form_Z's last barrier
    if (it.a < k) apply_panels(a, c2, i, k + 2, it.a, k, L, Ms, Xs, Ms, Xs);     // (ONE pair — Zs must survive: apply_panels then closes every panel with a second barrier)
    else lds_barrier();                                                            // either way: every wavefront is past its reads of Ys
#pragma unroll
    for (int q = 0; q < 4; ++q) Ys[L.row * LDT + L.cb + q * 16] = y2[q];
    lds_barrier();
    {
        const v4d acc = frag_product((unsigned)(uintptr_t)(Zs + (L.wr * 16 + L.fr) * LDT + L.fk), (unsigned)(uintptr_t)(Ys + (L.wc * 16 + L.fr) * LDT + L.fk));
#pragma unroll
        for (int r = 0; r < 4; ++r) T2[offC + 4 * r * a.NP] = c2[r] - acc[r];
    }
    lds_barrier();
}

__global__ __launch_bounds__(TR_THREADS) void k_lfac(const LfacArgs a) {
    __shared__ double smem[LFAC_LDS_DOUBLES];
    if (a.hprog && blockIdx.x == 0 && threadIdx.x == 0) __hip_atomic_store(a.hprog, a.ptag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    auto lanes = [] {
        LLane L;
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));          // (opaque: what is derived from the lane index is formed where it is used, not held in registers across the items)
        L.tid = tid; L.lane = L.tid & 63; L.wave = L.tid >> 6;
        L.wr = L.wave >> 2; L.wc = L.wave & 3; L.fr = L.lane & 15; L.fk = L.lane >> 4;
        L.row = (L.lane & 7) + 8 * ((L.lane >> 4) & 1) + 16 * (L.wave & 3);          // staging map of ldl.hip: k_ldl_step
        L.cb = ((L.lane >> 3) & 1) + 2 * ((L.lane >> 5) & 1) + 4 * (L.wave >> 2);
        return L;
    };
    if (blockIdx.x == 0) {
        const LLane L = lanes();
        // the pivot chain: k_ldl_step's tile 0 / k_ldl_diag
        if (a.k < -1) return;
        const int offC = (L.wr * 16 + L.fr) + (L.wc * 16 + L.fk) * a.NP;
        if (a.k == -1) {
            v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
            if (L.wr >= L.wc) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q] = a.S[offC + 4 * q * a.NP];
            }
            diag_block(smem, acc, a.NP, a.nx, 0, a.tb, a.S, a.Dx, a.Tinv, a.Minv, a.icount);
            return;
        }
        const int k0 = a.k * NB, r0 = k0 + NB;
        double* Zs = smem;
        double* Ys = smem + TT * LDT;
        double* Ms = Ys + TT * LDT;
        double cS[4];
        const double* Q = a.S + ((size_t)r0 + (size_t)r0 * a.NP);
#pragma unroll
        for (int r = 0; r < 4; ++r) cS[r] = Q[offC + 4 * r * a.NP];
        form_Z(a.S + ((size_t)r0 + (size_t)k0 * a.NP), a.NP, a.Minv + (size_t)a.k * NB * NB, Ys, Ms, Zs, L.row, L.cb, L.wr, L.wc, L.fr, L.fk);
        const v4d acc = frag_product((unsigned)(uintptr_t)(Zs + (L.wr * 16 + L.fr) * LDT + L.fk), (unsigned)(uintptr_t)(Ys + (L.wc * 16 + L.fr) * LDT + L.fk));
#pragma unroll
        for (int r = 0; r < 4; ++r) cS[r] -= acc[r];
        lds_barrier();
        diag_block(smem, (v4d){cS[0], cS[1], cS[2], cS[3]}, a.NP, a.nx, r0, a.tb, a.S, a.Dx, a.Tinv, a.Minv, a.icount);
        return;
    }
    const int w = (int)blockIdx.x - 1;
    const int first = a.wfirst[w], last = a.wfirst[w + 1];
#pragma unroll 1
    for (int idx = first; idx < last; ++idx) {
        const LItem it = a.items[idx];
        const LLane L = lanes();
        if (it.kind == LI_SCHUR) item_schur(a, it, L, smem);
        else if (it.kind == LI_FAR) item_far(a, it, L, smem);
        else item_row(a, it, L, smem);
    }
}

// ---- the plan (host, once per shape) -------------------------------------------------------------------------------------------------
struct LfacPlan {
    int nblk = 0, nst0 = 0, nst = 0, W = 0;
    std::vector<int> item0;                 // per launch (index = k + 2): offset of its (W + 1) worker offsets in wfirst
    std::vector<LItem> items;
    std::vector<int> wfirst;
    double budget = 0.0; int margin = 0;
    std::vector<double> load;               // per launch: the longest worker (cost units) — diagnostics
    std::vector<int> counts;                // per launch: items
    std::vector<double> mean;               // per launch: mean worker load
    std::vector<int> busy;                  // per launch: workers that have items (the launch's grid: the others would leave at once)
};
struct LfacAux {
    LfacPlan plan;
    int NP = 0, ne = 0, nc = 0;
    double *Zbuf = nullptr, *Lfac = nullptr;
    LItem* d_items = nullptr; int* d_wfirst = nullptr;
    double plan_ms = 0.0;                   // host time of the plan scan this handle paid (0: the shape's plan was in the process-wide cache)
};

// cost model (microseconds on one compute unit while the whole chip is busy: the fp64 matrix cores sustain ~41 TFLOP/s on real data, 161 GFLOP/s per unit — a
// 64 x 64 x 64 product 3.3 us, a Schur stage 1.64 us; bench/lfac_item_bench.hip): what balances the workers of a launch against the chain's workgroup
constexpr double LFAC_CHAIN_US = 19.4;      // what the chain's workgroup makes a panel launch last at least (C3: profiles/r05_lfac_timeline.txt)
constexpr double LFAC_LAUNCH_US = 0.0;      // (a launch whose workers all carry items lasts its longest worker + ~2.8 us of boundary and dispatch skew; pricing it — 1.4, 2.8 — made the scan choose plans that MEASURE 1-2 % slower: 1.105 / 1.12 ms against 1.095, so the scan keeps the bare model)
static double cost_schur(int stages, int P = 1) { return (P == 1 ? 1.52 : P == 2 ? 1.05 : 0.8) * stages + 3.3; }      // (measured: bench/lfac_items.py, less the launch boundary)
static double cost_far(int panels) { return 3.1 * panels + 2.5; }
static double cost_row(int pending) { return 15.0 + 2.2 * pending; }
static double cost_row_half(int pending, bool second) { return second ? 11.0 + 2.2 * pending : 10.0; }      // a row item cut in two (item_row: it.c)
static int stages_within(double us, int P = 1) { const int n = (int)((us - 3.3) / (P == 1 ? 1.52 : P == 2 ? 1.05 : 0.8)); return n < 1 ? 1 : n; }

// budget: the duration (us) a panel launch should have — the chain's workgroup takes ~19; head: the duration of launch -2 (0: long enough for whole tiles)
static bool lfac_make_plan(LfacPlan& P, int nblk, int nx, int ne, int nc, int W, double budget, double head, int margin = 1 << 20) {
    P = LfacPlan();
    P.nblk = nblk; P.W = W; P.budget = budget;
    P.nst0 = (ne + LKT - 1) / LKT; P.nst = P.nst0 + (nc + LKT - 1) / LKT;
    const int nst = P.nst;
    if (head <= 0.0) head = std::min(cost_schur(nst), 120.0);
    struct Tile { int st = 0, pn = 0; };
    std::vector<Tile> tiles((size_t)nblk * nblk);
    auto T = [&](int i, int j) -> Tile& { return tiles[(size_t)i * nblk + j]; };
    auto deadline = [&](int i, int j) { return (i == 0 && j == 0) ? -1 : std::max(0, j - 2); };   // the launch that consumes the tile: everything it is owed must be there before
    std::vector<std::pair<int, int>> order;         // candidates in the order of their deadlines (column-major)
    for (int j = 0; j < nblk; ++j) for (int i = j; i < nblk; ++i) {
        order.push_back({i, j});
        if (i * TT >= nx) T(i, j).st = nst;         // rows of the padding only: launch_scale_rows / launch_pad_identity wrote the unit pivots, nothing to accumulate
    }
    for (int ell = -2; ell <= nblk - 2; ++ell) {
        std::vector<std::pair<double, LItem>> its;
        const double target = ell == -2 ? head : budget;
        double used = 0.0;
        if (ell >= 0) {
            if (T(ell + 1, ell + 1).st != nst || T(ell + 1, ell + 1).pn != ell) return false;
            T(ell + 1, ell + 1).pn = ell + 1;                                         // the chain's tile
            for (int i = ell + 2; i < nblk; ++i) {
                Tile& t1 = T(i, ell + 1);
                if (t1.st != nst || t1.pn != ell) return false;
                t1.pn = ell + 1;
                LItem it{}; it.kind = LI_ROW; it.i = (short)i; it.j = (short)ell; it.c = 1;
                Tile& t2 = T(i, ell + 2);
                if (t2.st != nst) return false;
                it.a = (short)t2.pn;
                const double c = cost_row(ell - t2.pn);
                t2.pn = ell + 1;
                its.push_back({c, it}); used += c;
            }
        }
        // Deferred panels: a tile takes at most per_item a launch; `behind` ones whatever the load, the others where workers are left over (below)
        const int per_item = std::max(1, (int)((LFAC_CHAIN_US - LFAC_LAUNCH_US - 2.5) / 3.1));     // (sized for the chain's workgroup, not for the launch: behind the constraint products the launches are as short as the chain)
        std::vector<std::pair<int, int>> far_optional;
        for (const auto& ij : order) {
            const int i = ij.first, j = ij.second, D = deadline(i, j);
            if (D <= ell) continue;                                                   // consumed already (its remaining panels belong to the row items / the chain)
            Tile& t = T(i, j);
            if (t.st < nst) continue;
            const int n_avail = std::min(ell, j - 2) - t.pn;                          // panels < ell have their Z in Zbuf; the row item of launch j - 2 takes the tile from there
            if (n_avail <= 0) continue;
            const int owed = (D - 1) - t.pn, later_far = (D - ell - 1) * per_item;    // panels it must have taken before its row item; what the later launches can still give it
            if (!(owed > later_far || n_avail >= per_item)) { if (n_avail >= 2) far_optional.push_back(ij); continue; }
            const int n = std::min(n_avail, per_item);
            LItem it{}; it.kind = LI_FAR; it.i = (short)i; it.j = (short)j; it.a = (short)t.pn; it.b = (short)(t.pn + n);
            t.pn += n;
            its.push_back({cost_far(n), it}); used += cost_far(n);
        }
        double launch_len = target;
        std::vector<std::pair<int, int>> cand_all;      // the tiles the constraint products did not reach in this launch, in their order
        size_t fill_from = 0;
        // Constraint products: ONE slice per tile and launch, one tile per worker that is left — the tiles furthest behind their own deadline first, and every
        // slice as long as the launch is going to be anyway (the launch stretches to the slice the most urgent served tile needs: under overload the launches get
        // longer for all workers alike instead of some workers taking two items).  The tile's deadline for the products is ~0.8 D: behind them it has D - 1 panels
        // to take at per_item a launch while one more arrives with every launch: (1 - 1 / per_item) D.
        {
            const int spl_later = stages_within(budget);
            struct Cand { int needed, i, j, hard; };      // hard: no later launch before the tile's own deadline
            std::vector<Cand> cand;
            for (const auto& ij : order) {
                const int i = ij.first, j = ij.second, D = deadline(i, j);
                if (D <= ell) continue;
                const Tile& t = T(i, j);
                if (t.st >= nst) continue;
                const int Ds = D <= 2 ? D : std::max(2, (int)((1.0 - 1.0 / per_item) * (D - 1)) + 1);
                cand.push_back({(nst - t.st) - std::max(0, Ds - ell - 1) * spl_later, i, j, Ds - ell - 1 <= 0 ? 1 : 0});
            }
            std::stable_sort(cand.begin(), cand.end(), [](const Cand& x, const Cand& y) { return x.hard != y.hard ? x.hard > y.hard : x.needed > y.needed; });
            int slots = std::max(W - (int)its.size(), W / 4);
            // the length of the launch: what the tile furthest behind needs, between the chain's own duration (nothing is gained below it) and the target
            double len = target;
            if (ell != -2) {
                int most = 0;
                for (const Cand& c : cand) most = std::max(most, c.needed);
                len = std::min(target, std::max(LFAC_CHAIN_US - LFAC_LAUNCH_US, cost_schur(most)));
            }
            // a tile that needs more stages than one workgroup gives it in a launch of this length is split over 2 or 4 workgroups (rows of the tile; the kernel's
            // it.pad0 / it.pad1) — a few urgent tiles do not stretch the launch for everybody; only beyond that the launch gets longer
            for (const Cand& c : cand) if (c.hard && c.needed > stages_within(len, 4)) len = std::max(len, cost_schur(c.needed, 4));
            size_t served = 0;
            for (const Cand& c : cand) {
                if (slots <= 0) break;
                if (!c.hard && c.needed <= -margin * spl_later && ell != -2) break;      // (margin: how many launches ahead of its need a tile may be served — what waits leaves its worker's time to the tail of the chain)
                ++served;
                Tile& t = T(c.i, c.j);
                const int R = nst - t.st;
                int split = c.needed <= stages_within(len, 1) ? 1 : c.needed <= stages_within(len, 2) ? 2 : 4;
                if (split > slots) split = 1;
                const int n = std::min(R, stages_within(len, split));
                for (int part = 0; part < split; ++part) {
                    LItem it{}; it.kind = LI_SCHUR; it.i = (short)c.i; it.j = (short)c.j; it.a = (short)t.st; it.b = (short)(t.st + n);
                    it.c = (short)((t.st == 0 ? 1 : 0) | (t.st + n == nst ? 2 : 0));
                    it.pad0 = (short)split; it.pad1 = (short)part;
                    its.push_back({cost_schur(n, split), it}); used += cost_schur(n, split);
                }
                t.st += n;
                slots -= split;
            }
            launch_len = len;
            for (size_t q = served; q < cand.size(); ++q) cand_all.push_back({cand[q].i, cand[q].j});
        }
        // the workers that carry a row item or deferred panels are done before the launch is: a short slice of the next tiles in line fills their time (the longest-
        // first packing below puts the short slices on the least loaded workers)
        if (fill_from < cand_all.size() && ell != -2) {
            std::vector<double> spare;
            for (const auto& ci : its) if (ci.second.kind != LI_SCHUR && launch_len - ci.first >= cost_schur(4)) spare.push_back(launch_len - ci.first);
            std::sort(spare.begin(), spare.end(), std::greater<double>());
            for (double room : spare) {
                if (fill_from >= cand_all.size()) break;
                const auto c = cand_all[fill_from++];
                Tile& t = T(c.first, c.second);
                const int n = std::min(nst - t.st, stages_within(room));
                LItem it{}; it.kind = LI_SCHUR; it.i = (short)c.first; it.j = (short)c.second; it.a = (short)t.st; it.b = (short)(t.st + n);
                it.c = (short)((t.st == 0 ? 1 : 0) | (t.st + n == nst ? 2 : 0));
                it.pad0 = 1; it.pad1 = 0;
                t.st += n;
                its.push_back({cost_schur(n), it}); used += cost_schur(n);
            }
        }
        for (const auto& ij : far_optional) {
            if ((int)its.size() >= W) break;
            Tile& t = T(ij.first, ij.second);
            const int n = std::min(std::min(ell, ij.second - 2) - t.pn, per_item);
            LItem it{}; it.kind = LI_FAR; it.i = (short)ij.first; it.j = (short)ij.second; it.a = (short)t.pn; it.b = (short)(t.pn + n);
            t.pn += n;
            its.push_back({cost_far(n), it}); used += cost_far(n);
        }
        // where workers idle (the tail of the chain: a launch is as long as its longest item, and that is a row item) every row item is cut in two halves on two
        // workers: Z is formed twice, the two tiles of the row are updated side by side
        if (ell >= 0) {
            int rows = 0;
            for (const auto& ci : its) rows += ci.second.kind == LI_ROW;
            if (rows > 0 && (int)its.size() + rows <= W) {
                const size_t n0 = its.size();
                for (size_t q = 0; q < n0; ++q) {
                    if (its[q].second.kind != LI_ROW) continue;
                    LItem b = its[q].second;
                    const int pend = b.j - b.a;
                    its[q].second.c = 0; its[q].first = cost_row_half(0, false);
                    b.c = 3;
                    its.push_back({cost_row_half(pend, true), b});
                    used += cost_row_half(0, false) + cost_row_half(pend, true) - cost_row(pend);
                }
            }
        }
        // longest item first onto the least loaded worker
        std::stable_sort(its.begin(), its.end(), [](const std::pair<double, LItem>& x, const std::pair<double, LItem>& y) { return x.first > y.first; });
        std::vector<std::vector<LItem>> per(W);
        typedef std::pair<double, int> LW;
        std::priority_queue<LW, std::vector<LW>, std::greater<LW>> pq;
        for (int w = 0; w < W; ++w) pq.push({0.0, w});
        double longest = 0.0;
        for (const auto& ci : its) {
            LW lw = pq.top(); pq.pop();
            per[lw.second].push_back(ci.second);
            lw.first += ci.first; longest = std::max(longest, lw.first);
            pq.push(lw);
        }
        // the two tiles the NEXT launch's chain workgroup starts with — (ell + 2, ell + 1) and (ell + 2, ell + 2), the row items of row ell + 2 — are written on the
        // XCD the chain's workgroup runs on (block 0: XCD 0; worker w is block w + 1): its loads then hit that XCD's L2 instead of going to memory (0.3 - 0.6 us per launch)
        if (ell >= 0) {
            int slot = 7;
            for (int w = 0; w < W && slot < W; ++w) {
                bool has = false;
                for (const LItem& it : per[w]) has = has || (it.kind == LI_ROW && it.i == ell + 2);
                if (!has || (w + 1) % 8 == 0) continue;
                while (slot < W) {
                    bool busy_slot = false;
                    for (const LItem& it : per[slot]) busy_slot = busy_slot || (it.kind == LI_ROW && it.i == ell + 2);
                    if (!busy_slot) break;
                    slot += 8;
                }
                if (slot < W) { std::swap(per[w], per[slot]); slot += 8; }
            }
        }
        P.item0.push_back((int)P.wfirst.size());
        for (int w = 0; w < W; ++w) { P.wfirst.push_back((int)P.items.size()); for (const LItem& it : per[w]) P.items.push_back(it); }
        P.wfirst.push_back((int)P.items.size());
        P.load.push_back(longest); P.counts.push_back((int)its.size()); P.mean.push_back(used / W);
        { int busy = 0; for (int w = 0; w < W; ++w) if (!per[w].empty()) busy = w + 1; P.busy.push_back(busy); }
    }
    for (int j = 0; j < nblk; ++j) for (int i = j; i < nblk; ++i) if (T(i, j).st != nst || T(i, j).pn != j) return false;      // every tile complete?
    return true;
}


// what the plan's own cost model says the launches take: the head as long as its longest worker, a panel launch at least as long as the chain's workgroup
static double lfac_plan_estimate(const LfacPlan& P) {
    double e = P.load.empty() ? 0.0 : P.load[0];
    for (size_t l = 1; l < P.load.size(); ++l) e += std::max(P.load[l] + LFAC_LAUNCH_US, LFAC_CHAIN_US);
    return e;
}
// The launch length (budget) and the head are chosen by the model (it reproduces the measured timeline within a few percent: profiles/r05_lfac_timeline.txt): a scan
// of both (a fixed pair can be tried through calipso_hip_debug_lfac_plan / bench/lfac_plan.py).  The planner costs a few milliseconds per candidate, once per handle.
static bool lfac_scan_plans(LfacPlan& best, int nblk, int nx, int ne, int nc, int W) {
    // (a candidate costs ~ nblk^2 microseconds of host time: 630 of them are 0.4 s at C3's 40 panels and 8 s at 128 on ONE core — a coarser grid beyond 48 panels, and
    // the candidates are independent: they are dealt over the host's cores, the winner is the first of the cheapest in candidate order whatever the thread count)
    struct Cand { double b, h; int margin; };
    std::vector<Cand> cands;
    const bool coarse = nblk > 48;
    for (double b = 19.0; b <= 40.0; b += coarse ? 3.0 : 1.5)
        for (double h = 50.0; h <= 130.0; h += coarse ? 20.0 : 10.0)
            for (int margin : {1 << 20, 3, 2, 1, 0}) {
                if (coarse && margin != (1 << 20) && margin != 1) continue;
                cands.push_back({b, h, margin});
            }
    static const int env_threads = [] { const char* e = getenv("CALIPSO_HIP_LFAC_PLAN_THREADS"); return e ? atoi(e) : 0; }();
    unsigned hw = std::thread::hardware_concurrency();
    const int nthreads = std::max(1, std::min<int>({env_threads > 0 ? env_threads : 64, hw ? (int)hw : 1, (int)cands.size()}));      // (the host has nothing else to do: the first factorisation of the shape waits for this)
    std::vector<double> est(cands.size(), -1.0);            // the model's estimate of every feasible candidate (-1: infeasible)
    std::atomic<size_t> next{0};
    auto work = [&] {
        for (size_t c; (c = next.fetch_add(1)) < cands.size();) {
            LfacPlan P;
            if (lfac_make_plan(P, nblk, nx, ne, nc, W, cands[c].b, cands[c].h, cands[c].margin)) est[c] = lfac_plan_estimate(P);
        }
    };
    {
        std::vector<std::thread> pool;
        for (int t = 1; t < nthreads; ++t) pool.emplace_back(work);
        work();
        for (auto& t : pool) t.join();
    }
    int win = -1;
    for (size_t c = 0; c < cands.size(); ++c) if (est[c] >= 0.0 && (win < 0 || est[c] < est[(size_t)win])) win = (int)c;
    if (win >= 0) {
        if (!lfac_make_plan(best, nblk, nx, ne, nc, W, cands[(size_t)win].b, cands[(size_t)win].h, cands[(size_t)win].margin)) return false;     // (deterministic: the same plan again)
        best.margin = cands[(size_t)win].margin;
        return true;
    }
    // Shapes the grid does not cover — a tile's constraint products alone outlast the grid's longest head (cost_schur(nst, 4) > 130: m > ~5000), or the work per
    // launch its longest budget (nx = 6000, m = 4900: 94 panels of >= 80 us of work each): longer launches, heads in multiples of the shortest one that can
    // finish a tile (split four ways; from twice that a tile takes ONE worker and the first two columns fit the 255 slots), the cheapest by the model
    {
        const int nst = (ne + LKT - 1) / LKT + (nc + LKT - 1) / LKT;
        const double h0 = std::max(50.0, cost_schur(nst, 4));
        double best_e = -1.0;
        for (double b = 40.0; b <= 640.0; b *= 1.25)
            for (double hm : {1.0, 2.0, 3.0}) {
                LfacPlan P;
                if (!lfac_make_plan(P, nblk, nx, ne, nc, W, b, hm * h0)) continue;
                const double e = lfac_plan_estimate(P);
                if (best_e < 0.0 || e < best_e) { best_e = e; best = P; }
            }
        return best_e >= 0.0;
    }
}
// One plan per SHAPE and process: handles of one shape (the lanes of bench.py, the members of a batch stepped alone, a handle re-created per MPC step) share it — the
// scan runs once, every later handle of the shape finds its plan here (a shape that cannot be planned is remembered too).
static std::shared_ptr<const LfacPlan> lfac_cached_plan(int nblk, int nx, int ne, int nc, int W, double* host_ms = nullptr) {
    static std::mutex mu;
    static std::map<std::array<int, 5>, std::shared_ptr<const LfacPlan>> cache;
    const std::array<int, 5> key{nblk, nx, ne, nc, W};
    std::lock_guard<std::mutex> lock(mu);      // (a second handle of the shape arriving during the scan waits for it instead of scanning again)
    if (host_ms) *host_ms = 0.0;
    const auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    const auto t0 = std::chrono::steady_clock::now();
    auto P = std::make_shared<LfacPlan>();
    std::shared_ptr<const LfacPlan> out;
    if (lfac_scan_plans(*P, nblk, nx, ne, nc, W)) out = P;
    cache[key] = out;
    if (host_ms) *host_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return out;
}
static bool lfac_best_plan(LfacPlan& best, int nblk, int nx, int ne, int nc, int W) {
    const auto P = lfac_cached_plan(nblk, nx, ne, nc, W);
    if (!P) return false;
    best = *P;
    return true;
}

bool lfac_on(const calipso_hip_solver* s) {
    static const int env = [] { const char* e = getenv("CALIPSO_HIP_LFAC"); return e ? atoi(e) : 1; }();
    // CALIPSO_HIP_GRAPH_LDL=1 (ldl.hip: the panel steps as a captured graph on one stream) has no left-looking schedule: k_lfac takes the handle's scalars (rho, the
    // regularisation) BY VALUE, so a replayed capture would rebuild S with the scalars of the first factorisation
    static const bool graph_ldl = [] { const char* e = getenv("CALIPSO_HIP_GRAPH_LDL"); return e && atoi(e) != 0; }();
    if (!env || graph_ldl || s->cur || s->band64 > 0 || s->compact || s->blocks.on || (s->stage_parallel && s->spS) || s->lfac_failed) return false;
    return s->d.NP >= 1024 && s->d.NP <= 8192 && s->d.m > 0;
}

void lfac_release(calipso_hip_solver* s) {
    LfacAux* A = static_cast<LfacAux*>(s->lfac_aux);
    if (!A) return;
    if (A->Zbuf) (void)hipFree(A->Zbuf);
    if (A->Lfac) (void)hipFree(A->Lfac);
    if (A->d_items) (void)hipFree(A->d_items);
    if (A->d_wfirst) (void)hipFree(A->d_wfirst);
    delete A;
    s->lfac_aux = nullptr;
}


// plan + buffers of the handle's shape (made on first use, remade when the shape's constraint counts change — they do not for a live handle)
static LfacAux* lfac_prepare(calipso_hip_solver* s) {
    LfacAux* A = static_cast<LfacAux*>(s->lfac_aux);
    const int NP = s->d.NP, nblk = NP / NB;
    if (A && A->NP == NP && A->ne == s->d.ne && A->nc == s->d.nc) return A;
    lfac_release(s);
    A = new LfacAux();
    A->NP = NP; A->ne = s->d.ne; A->nc = s->d.nc;
    bool ok = false;
    {
        double ms = 0.0;
        const auto P = lfac_cached_plan(nblk, s->d.nx, s->d.ne, s->d.nc, 255, &ms);
        A->plan_ms = ms;
        if (P) { A->plan = *P; ok = true; }
        else {      // (not an error: the handle keeps k_schur + the right-looking panel steps — but say so: calipso_hip_last_error, calipso_hip_kernel_times[6] = -1)
            char buf[256];
            snprintf(buf, sizeof buf, "note: no left-looking plan meets its deadlines for nx = %d, ne = %d, nc = %d (NP = %d): the right-looking schedule is kept", s->d.nx, s->d.ne, s->d.nc, NP);
            s->err = buf;
        }
    }
    const size_t nn = (size_t)NP * NP;
    const bool planned = ok;
    ok = ok && hipMalloc((void**)&A->Zbuf, nn * sizeof(double)) == hipSuccess && hipMalloc((void**)&A->Lfac, nn * sizeof(double)) == hipSuccess &&
         hipMalloc((void**)&A->d_items, std::max<size_t>(1, A->plan.items.size()) * sizeof(LItem)) == hipSuccess &&
         hipMalloc((void**)&A->d_wfirst, A->plan.wfirst.size() * sizeof(int)) == hipSuccess &&
         hipMemcpy(A->d_items, A->plan.items.data(), A->plan.items.size() * sizeof(LItem), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemcpy(A->d_wfirst, A->plan.wfirst.data(), A->plan.wfirst.size() * sizeof(int), hipMemcpyHostToDevice) == hipSuccess &&
         hipMemset(A->Lfac, 0, nn * sizeof(double)) == hipSuccess;
    s->lfac_aux = A;
    if (!ok) {
        if (planned) { (void)hipGetLastError(); s->err = "note: the buffers of the left-looking schedule (2 NP^2 doubles) could not be allocated: the right-looking schedule is kept"; }
        lfac_release(s); s->lfac_failed = true; return nullptr;
    }
    return A;
}

// lfac_on + plan and buffers in place (made on first use; a handle that cannot have them keeps the right-looking schedule from then on)
bool lfac_ready(calipso_hip_solver* s) { return lfac_on(s) && lfac_prepare(s) != nullptr; }

// where the factor columns of this handle live: its own buffer under this schedule, S otherwise (ldl.hip scales in place)
double* lfac_factor_buffer(calipso_hip_solver* s) {
    if (!lfac_ready(s)) return s->S;
    return static_cast<LfacAux*>(s->lfac_aux)->Lfac;
}

// the launches -2 .. nblk - 2 on the handle's stream; the panel launches carry the progress tags the second stream's feeds wait for (ldl.hip: launch_ldl)
int lfac_enqueue(calipso_hip_solver* s, unsigned long long* hprog, unsigned long long epoch) {
    LfacAux* A = lfac_prepare(s);
    if (!A) return 0;
    const int NP = s->d.NP, nblk = NP / NB;
    LfacArgs a;
    a.NP = NP; a.nx = s->d.nx; a.m = s->d.m; a.ne = s->d.ne; a.nc = s->d.nc; a.tb = trsv_block(NP, (int)s->solve_block);
    a.nst0 = A->plan.nst0; a.nst = A->plan.nst;
    a.S = s->S; a.Zbuf = A->Zbuf; a.Minv = s->Ypanel; a.Dx = s->Dx; a.Tinv = s->Tinv;
    a.Lsym = s->Lsym; a.Zj = s->Z; a.WH = s->WH; a.icount = s->icount;
    a.sc = s->sc;
    int launches = 0;
    for (int ell = -2; ell <= nblk - 2; ++ell) {
        a.k = ell;
        a.items = A->d_items;
        a.wfirst = A->d_wfirst + A->plan.item0[ell + 2];
        a.hprog = ell >= 0 ? hprog : (unsigned long long*)nullptr;
        a.ptag = epoch | (unsigned long long)(ell >= 0 ? ell : 0);
        hipLaunchKernelGGL(k_lfac, dim3(1 + A->plan.busy[ell + 2]), dim3(TR_THREADS), 0, s->stream, a);
        ++launches;
    }
    return launches;
}

// diagnostics for bench.py / tests: [0] launches, [1] items, [2] budget, [3] longest worker of the head, [4] mean longest worker of the panel launches, [5] bytes of the
// schedule's buffers, [6] budget (us per panel launch) the scan chose, [7] the plan's own estimate of the launches' total (us)
void lfac_describe(calipso_hip_solver* s, double out[8]) {
    for (int i = 0; i < 8; ++i) out[i] = 0.0;
    LfacAux* A = static_cast<LfacAux*>(s->lfac_aux);
    if (!A) return;
    out[0] = (double)A->plan.item0.size(); out[1] = (double)A->plan.items.size(); out[2] = A->plan.budget; out[3] = A->plan.load.empty() ? 0.0 : A->plan.load[0];
    double sum = 0.0; int n = 0;
    for (size_t l = 2; l < A->plan.load.size(); ++l) { sum += A->plan.load[l]; ++n; }
    out[4] = n ? sum / n : 0.0;
    out[5] = 2.0 * (double)A->NP * A->NP * sizeof(double) + (double)A->plan.items.size() * sizeof(LItem) + (double)A->plan.wfirst.size() * sizeof(int);      // bytes outside the slab
    out[6] = A->plan_ms; out[7] = lfac_plan_estimate(A->plan);      // [6]: host milliseconds this handle spent scanning plans (0: cache hit)
}

}  // namespace calipso

// the handle's schedule (bench/lfac_sizes.py): lfac_describe's eight numbers — [6] = host milliseconds of the plan scan this handle paid (0: cache hit)
extern "C" int32_t calipso_hip_debug_lfac_describe(calipso_hip_solver* s, double out[8]) {
    if (!s || !out) return CALIPSO_ERR_ARGUMENT;
    calipso::lfac_describe(s, out);
    return CALIPSO_OK;
}

// plan diagnostics without a device (tests, bench): per launch [longest worker (cost units), items, SCHUR items, FAR items, ROW items, longest ROW item]; returns the number of
// launches, 0 if the shape cannot be planned within the budget
extern "C" int32_t calipso_hip_debug_lfac_plan(int32_t nblk, int32_t nx, int32_t ne, int32_t nc, double budget, double head, double* out, int32_t max_launches) {
    calipso::LfacPlan P;
    if (budget <= 0.0 ? !calipso::lfac_best_plan(P, nblk, nx, ne, nc, 255) : !calipso::lfac_make_plan(P, nblk, nx, ne, nc, 255, budget, head)) return 0;
    if (max_launches > 0) { out[6 * (max_launches - 1)] = P.budget; out[6 * (max_launches - 1) + 1] = P.margin; out[6 * (max_launches - 1) + 2] = P.load.empty() ? 0.0 : P.load[0]; }
    const int nl = (int)P.item0.size();
    for (int l = 0; l < nl && l < max_launches; ++l) {
        const int w0 = P.item0[l];
        const int i0 = P.wfirst[w0], i1 = P.wfirst[w0 + P.W];
        double kinds[3] = {0, 0, 0}, longrow = 0;
        for (int i = i0; i < i1; ++i) {
            const calipso::LItem& it = P.items[i];
            kinds[it.kind] += 1;
            if (it.kind == calipso::LI_ROW) longrow = std::max(longrow, calipso::cost_row(it.j - it.a));
        }
        out[6 * l + 0] = P.load[l]; out[6 * l + 1] = i1 - i0; out[6 * l + 2] = kinds[0]; out[6 * l + 3] = kinds[1]; out[6 * l + 4] = kinds[2]; out[6 * l + 5] = P.mean[l]; (void)longrow;
    }
    return nl;
}

// Timing of synthetic item lists on the real kernel (bench/lfac_items.py; the handle's factor is garbage afterwards): every worker gets `per_worker` items of one kind —
// kind 0: SCHUR slices of n stages split P ways (skew: every worker at a stage range of its own), 1: FAR items of n panels, 2: ROW items with n pending panels.  Returns the
// mean duration of the launch in microseconds (5 launches), < 0 on failure.
extern "C" double calipso_hip_debug_lfac_items(calipso_hip_solver* s, int32_t kind, int32_t n, int32_t P, int32_t per_worker, int32_t skew, int32_t active) {
    using namespace calipso;
    if (!s || !lfac_ready(s)) return -1.0;
    LfacAux* A = static_cast<LfacAux*>(s->lfac_aux);
    const int NP = s->d.NP, nblk = NP / NB, W = A->plan.W, nst = A->plan.nst;
    std::vector<LItem> items; std::vector<int> wfirst;
    std::vector<std::pair<int, int>> tiles;
    for (int j = 0; j < nblk; ++j) for (int i = j; i < nblk; ++i) tiles.push_back({i, j});
    size_t next = 0;
    for (int w = 0; w < W; ++w) {
        wfirst.push_back((int)items.size());
        for (int q = 0; q < per_worker && (active <= 0 || (w * active) / W != ((w + 1) * active) / W); ++q) {      // active > 0: only that many workers (spread evenly) get items
            LItem it{};
            if (kind == 0) {
                const auto ij = tiles[(next++) % tiles.size()];
                const int a0 = skew ? (w * 7) % std::max(1, nst - n) : 0;
                it.kind = LI_SCHUR; it.i = (short)ij.first; it.j = (short)ij.second; it.a = (short)a0; it.b = (short)std::min(nst, a0 + n); it.c = 1; it.pad0 = (short)P; it.pad1 = (short)(w % std::max(1, P));
            } else if (kind == 1) {
                std::pair<int, int> ij;
                do { ij = tiles[(next++) % tiles.size()]; } while (ij.second < n + 3);
                it.kind = LI_FAR; it.i = (short)ij.first; it.j = (short)ij.second; it.a = 0; it.b = (short)n;
            } else {
                const int k = n + (w % 3), i = k + 2 + (int)((next++) % (size_t)(nblk - k - 2));
                it.kind = LI_ROW; it.i = (short)i; it.j = (short)k; it.a = (short)(k - n); it.c = (short)(kind == 3 ? 0 : kind == 4 ? 3 : 1);      // kind 3 / 4: the two halves of a row item
            }
            items.push_back(it);
        }
    }
    wfirst.push_back((int)items.size());
    LItem* d_items = nullptr; int* d_wfirst = nullptr;
    if (hipMalloc((void**)&d_items, items.size() * sizeof(LItem)) != hipSuccess || hipMalloc((void**)&d_wfirst, wfirst.size() * sizeof(int)) != hipSuccess) return -1.0;
    (void)hipMemcpy(d_items, items.data(), items.size() * sizeof(LItem), hipMemcpyHostToDevice);
    (void)hipMemcpy(d_wfirst, wfirst.data(), wfirst.size() * sizeof(int), hipMemcpyHostToDevice);
    LfacArgs a;
    a.NP = NP; a.nx = s->d.nx; a.m = s->d.m; a.ne = s->d.ne; a.nc = s->d.nc; a.tb = trsv_block(NP, (int)s->solve_block);
    a.nst0 = A->plan.nst0; a.nst = nst; a.k = -2;
    a.S = s->S; a.Zbuf = A->Zbuf; a.Minv = s->Ypanel; a.Dx = s->Dx; a.Tinv = s->Tinv; a.Lsym = s->Lsym; a.Zj = s->Z; a.WH = s->WH; a.icount = s->icount;
    a.items = d_items; a.wfirst = d_wfirst; a.hprog = nullptr; a.ptag = 0; a.sc = s->sc;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_lfac, dim3(1 + W), dim3(TR_THREADS), 0, s->stream, a);
    (void)hipStreamSynchronize(s->stream);
    (void)hipEventRecord(e0, s->stream);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_lfac, dim3(1 + W), dim3(TR_THREADS), 0, s->stream, a);
    (void)hipEventRecord(e1, s->stream);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(d_items); (void)hipFree(d_wfirst);
    return (double)ms * 1e3 / 5.0;
}
