// internal.hpp — device state and kernel launchers shared by the translation units of libcalipso_hip.so.
// Product code (gfx950 only).  Nothing in here touches oracle/.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/calipso_hip.h"

namespace calipso {

typedef int64_t i64;

constexpr int TILE = 128;   // Schur-complement (SYRK) workgroup tile
constexpr int NB = 64;      // LDL^T panel width
constexpr int MAX_SOC_DIM = 1024;  // widest second-order cone (soc_wide.hip: up to sixteen elements per lane of the cone's wavefront)
constexpr int TRSV_BLOCK = 2048;   // widest diagonal block of L whose inverse is assembled for the triangular solves (ldl.hip)
// limit: the handle's "opt.solve_block" (512, 1024 or 2048).  NP is a power of two up to 512 and a multiple of 512 beyond; the inverse of a block is assembled
// by pairwise merges of equal halves, so every block of the layout — the last, narrower one included — must be 64 * 2^k wide: the widest power of two
// <= min(limit, NP) whose remainder NP mod tb is one as well (NP = 1536: 1024 + 512 also under limit 2048; NP = 2560: 2048 + 512).
inline int trsv_block(int NP, int limit = TRSV_BLOCK) {
    if (NP <= 512) return NP;
    int tb = 512;
    while (2 * tb <= limit && 2 * tb <= NP) tb *= 2;
    for (;; tb /= 2) { const int rem = NP % tb; if (tb == 512 || (rem & (rem - 1)) == 0) return tb; }
}
inline size_t tinv_doubles(int NP) {      // every block is stored with the leading dimension tb of its layout; sized for the largest of the three layouts
    size_t most = 0;
    for (int limit : {512, 1024, 2048}) { const size_t tb = (size_t)trsv_block(NP, limit); const size_t n = ((size_t)NP + tb - 1) / tb * tb * tb; most = n > most ? n : most; }
    return most;
}
// W-form of the triangular solves (ldl.hip): W_b = L[rows below block b, block b] * Tinv_b for every solve block b that has rows below it, stored block after
// block with leading dimension rb = NP - k0 - w (the rows below).  With it a forward step  u_b = Tinv_b b_b ; b_below -= L[below, b] u_b  is ONE mat-vec with
// the stacked matrix [Tinv_b; W_b] on b_b, and a backward step  v_b = Tinv_b'(z_b - L[below, b]' v_below) = [Tinv_b; W_b]'[z_b; -v_below]  as well: a solve is
// 2 nb dependent launches instead of 4 nb - 2.
inline bool wform_layout_ok(int NP, int tb) { return NP > tb && NP <= 4096; }
inline size_t wform_offset(int NP, int tb, int kb) {
    size_t off = 0;
    for (int b = 0; b < kb; ++b) { const int k0 = b * tb, w = tb < NP - k0 ? tb : NP - k0; off += (size_t)(NP - k0 - w) * (size_t)w; }
    return off;
}
constexpr int CONE_MASK_WORDS = 26;                     // icount[6..31] (slack) and icount[32..57] (slack dual): one bit per trial step size
constexpr int CONE_MASK_TRIALS = 32 * CONE_MASK_WORDS;  // => max_cone_line_search <= 831

// options.jl:6-59 (hot-path relevant subset + the rest for API parity)
struct Options {
    double residual_norm = 1.0, constraint_norm = 1.0;
    i64 max_outer_iterations = 10, max_residual_iterations = 100;
    double scaling_line_search = 0.5;
    i64 max_residual_line_search = 25, max_cone_line_search = 25;
    i64 iterative_refinement = 1, max_iterative_refinement = 10, min_iterative_refinement = 1;
    double iterative_refinement_tolerance = 1.0e-10;
    double central_path_initial = 1.0, central_path_update_tolerance = 10.0, central_path_scaling = 0.2, central_path_exponent = 1.5;
    double penalty_initial = 1.0, penalty_scaling = 10.0, dual_initial = 0.0;
    double residual_tolerance = 1.0e-4, optimality_tolerance = 1.0e-4, slack_tolerance = 1.0e-4, equality_tolerance = 1.0e-4,
           complementarity_tolerance = 1.0e-4;
    double min_regularization = 1.0e-20, primal_regularization_initial = 1.0e-7, dual_regularization_initial = 1.0e-7,
           max_regularization = 1.0e40, dual_regularization = 1.0e-8, dual_regularization_exponent = 0.25,
           scaling_regularization_initial = 100.0, scaling_regularization = 8.0, scaling_regularization_last = 1.0 / 3.0;
    double min_central_path = 1.0e-8, max_penalty = 1.0e8;
    double constraint_tensor = 1.0, update_factorization = 1.0;
    double violation_tolerance = 1.0e-5, violation_exponent = 1.1, merit_tolerance = 1.0e-5, merit_exponent = 2.3,
           armijo_tolerance = 1.0e-4, machine_tolerance = 1.0e-16;
    double max_filter = 1000, differentiate = 1.0, warmstart = 0.0;
};

// scalars the host owns and passes to kernels by value (solver.jl:81-127)
struct Scalars {
    double kappa = 0.1, tau = 0.99, rho = 10.0, ep = 0.0, ep_last = 0.0, ed = 0.0;
};

// Which problem instances a launch covers.  Every handle carves all its device buffers out of ONE slab with the same layout, so
// "buffer X of instance k" = "buffer X of the base handle" + delta[k] doubles, for every X.  Kernels take this by value, use
// blockIdx.z as the instance slot and shift their pointers (device_utils.hpp: inst_shift).  A single handle is a batch of one
// with delta 0; a group (group.hip) steps several same-shape handles in lockstep through the same launches.
constexpr int MAX_BATCH = 128;
struct Batch {
    int n = 1;                       // gridDim.z
    long long delta[MAX_BATCH] = {0};
    int slot[MAX_BATCH] = {0};       // the member's index in its group (0 for a handle stepped alone): addresses per-member storage outside the slabs
};
// + the per-instance scalars.  A handle stepped alone carries its scalars by value (sc1); a group's (up to 128 x 48 bytes: more than a kernel may take as arguments
// beside the instance list) sit in a device table the group driver uploads in stream order whenever the active set or a member's scalars change (group.hip).
struct BatchSc {
    Batch b;
    const Scalars* sctab = nullptr;
    Scalars sc1;
    __host__ __device__ const Scalars& scal(int z) const { return sctab ? sctab[z] : sc1; }
};

// Structure of the matrix a mat-vec works on (structure.hip): the loads of entries that are structurally zero are predicated off;
// the arithmetic (which lane adds what, in which order) is that of the dense kernels, so results do not change by a bit.
enum { SP_DENSE = 0, SP_Z = 1 /* [gx; hx], m rows */, SP_GX = 2 /* gx, ne rows */, SP_HX = 3 /* hx, nc rows */, SP_LXX = 4 /* banded square */ };
struct Sparsity {
    int kind = SP_DENSE;
    int hb = 0;                      // SP_LXX: half bandwidth
    int ne = 0;                      // SP_Z: offset of the cone rows
    const int* kr = nullptr;         // per 16-column group: [eq_lo, eq_hi, cone_lo, cone_hi) rows that touch it (per instance, in the slab)
    const int* rowrange = nullptr;   // per row of [gx; hx]: [first, last + 1) non-zero column (per instance, in the slab)
};

struct Dims {
    int nx, np, ne, nc, n, N, m;  // m = ne + nc
    int q;                        // number of nonnegative entries (cone-local 0..q-1)
    int n_soc;                    // number of (non-empty) second-order cones
    int max_dim;                  // largest SOC dimension
    int n_wide;                   // second-order cones of dimension > 4
    int NP;                       // nx padded to a multiple of TILE
    // offsets into a Point (point.jl:13-22)
    __host__ __device__ int orr() const { return nx; }
    __host__ __device__ int os() const { return nx + ne; }
    __host__ __device__ int oy() const { return nx + ne + nc; }
    __host__ __device__ int oz() const { return nx + ne + nc + ne; }
    __host__ __device__ int ot() const { return nx + ne + nc + ne + nc; }
};

// device-side description of the cone layout (every cone is a contiguous range; validated at create)
struct ConeDev {
    int* soc_start = nullptr;   // [n_soc] cone-local start of each SOC
    int* soc_dim = nullptr;     // [n_soc]
    int* soc_woff = nullptr;    // [n_soc] offset of its d x d weight block in Wsoc
    int* entry_soc = nullptr;   // [nc] SOC id of an entry, -1 for nonnegative entries
    int* wide = nullptr;        // [n_wide] ids of the cones of dimension > 4 (soc_wide.hip: one wavefront per cone)
};

// Stage blocks (blocks.hip): packed copies of the blocks of [gx; hx] and Lxx of a stage-structured problem, and the tables the block kernels walk.
// Offsets are in doubles from the start of the slab region of Lsym (unused in this mode), the same for every handle of one structure.
struct ZBlock { int row0, nrows, col0, ncols; long long off_c, off_r; };   // rows [row0, row0 + nrows) x columns [col0, col0 + ncols): column-major (ld nrows) and row-major (ld ncols) copies
struct LBlock { int c0, n; long long off_c, off_r; };                      // diagonal block of Lxx
struct Segment { int c0, nc, first, count; };                              // <= 64 columns; covering Z blocks: segblk[first .. first + count), ascending
struct SegPair { int a, b, first, count, lblock; long long soff; };        // tile (segment a >= segment b) of S: the Z blocks covering both, the Hessian block containing both (-1: none); its offset in the packed S of a structured handle
struct StageBlocks {
    bool on = false;
    int nblk = 0, nlb = 0, nseg = 0, npairs = 0, max_lb = 0;
    size_t packed = 0;                  // doubles per instance
    unsigned long long signature = 0;   // of the block structure (members of a group must share it)
    ZBlock* d_blk = nullptr; LBlock* d_lblk = nullptr; Segment* d_seg = nullptr; int* d_segblk = nullptr; SegPair* d_pairs = nullptr; int* d_pairblk = nullptr;
    int* d_colrange = nullptr;          // per column of Lxx: [first, last + 1) row of its Hessian block (uploads are re-checked against it)
    std::vector<calipso_device_block> h_jdesc, h_hdesc;   // descriptors of the blocks for a calipso_device_block_eval_fn (host copies; values = device addresses of the base handle)
    calipso_device_block *d_jdesc = nullptr, *d_hdesc = nullptr;
    int* d_rowcov = nullptr;            // per row of [gx; hx]: covered by some block (blocks_pack_from; made on first use)
    double schur_flops = 0.0;           // multiply-adds x 2 of one k_schur_blocks launch per instance (sum over the segment pairs of the covering blocks' rows x columns x columns)
    std::vector<ZBlock> h_blk; std::vector<LBlock> h_lblk; std::vector<SegPair> h_pairs; std::vector<Segment> h_seg; std::vector<int> h_seg_of_col;   // host copies (uploads of structured handles are packed on the host)
};
struct BlockPlan {                      // blocks.hip: blocks_plan (host only)
    std::vector<ZBlock> zb; std::vector<LBlock> lb; std::vector<Segment> seg; std::vector<int> segblk, pairblk, colrange, seg_of_col; std::vector<SegPair> pairs;
    size_t packed = 0, spacked = 0; int max_lb = 0; unsigned long long signature = 0;
};

struct QpEval {
    bool attached = false;
    double scale = 0.5;
    double *q = nullptr, *bh = nullptr;   // device: q[nx], bh = [-b; h] (m)
};

struct Stats {
    i64 total_iterations = 0, outer = 0, factorizations = 0, refine_fail = 0, refine_max = 0, fallbacks = 0, last_refine = 0, newton_steps = 0;
};

}  // namespace calipso

struct calipso_hip_solver {
    calipso::Dims d;
    calipso::Options opt;
    calipso::Scalars sc;
    calipso::ConeDev cone;
    calipso::QpEval qp;
    calipso::Stats stats;
    struct calipso_hip_group* owner = nullptr;   // the group this handle is a member of (group.hip), if any
    const calipso::BatchSc* cur = nullptr;   // set by the group driver on its base handle: launches cover these instances
    double* slab = nullptr; size_t slab_doubles = 0;   // all per-instance device buffers live here (see calipso::Batch)
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    calipso_callback_fn cb_inner = nullptr, cb_outer = nullptr;   // options.callback_inner / callback_outer (solver.jl:183,193)
    void* cb_user = nullptr;
    calipso_device_eval_fn dev_eval = nullptr;   // user evaluation on the device (include/calipso_hip.h): enqueues on `stream`, never syncs
    void* dev_eval_user = nullptr;
    calipso_device_block_eval_fn dev_block_eval = nullptr;   // structured handles: the evaluator writes the packed blocks (no dense scratch)
    void* dev_block_eval_user = nullptr;
    size_t scratch_bytes = 0;                    // dense scratch of a structured handle with a dense-layout device evaluator (evalL / evalZ)
    bool rhs_ahead = false, rhs_joined = false;   // the operands of the first condensed solve were queued on the second stream during this factorisation (ldl.hip: ldl_rhs_stream) / the main stream has joined it
    double *evalL = nullptr, *evalZ = nullptr;   // structured handle with a device evaluator: dense scratch (nx^2, m nx) the evaluator writes; packed into the blocks behind it
    std::map<std::string, double*> optd;
    // host copies of the layout
    std::vector<int> h_soc_start, h_soc_dim, h_soc_woff;
    std::vector<int64_t> h_nonneg, h_soc_ptr, h_soc_idx;
    // ---- device buffers -------------------------------------------------------------------------------------------
    // ProblemData
    double* Lxx = nullptr;      // nx*nx
    double* Z = nullptr;        // (ne+nc) x nx, ld = m: the equality Jacobian stacked on the cone Jacobian, so that every
                                // mat-vec with [gx; hx] is ONE launch (y and z are adjacent in a Point, point.jl:13-22)
    double *gx = nullptr, *hx = nullptr;   // = Z, Z + ne (leading dimension m)
    double* Lsym = nullptr;     // nx*nx: Lxx mirrored from its upper triangle (what triu(K) sees)
    bool hessian_dirty = true;
    double *fx = nullptr, *gyx = nullptr, *hzx = nullptr;   // nx each
    double* gh = nullptr;       // m: [equality_constraint; cone_constraint] contiguous
    double *g = nullptr, *hc = nullptr;    // = gh, gh + ne
    double *cone_product = nullptr, *cone_target = nullptr, *barrier_gradient = nullptr;  // nc
    double* dscal = nullptr;   // device scalars: [0] objective [1] barrier  [2..] reduction outputs (see kernels)
    double* hscal = nullptr;   // pinned host mirror of dscal
    double* hscal_dev = nullptr;            // its device-side address (the publish kernel of api.hip stores into it)
    unsigned long long* hseq = nullptr;     // pinned: sequence number the publish kernel writes after the values
    unsigned long long* hseq_dev = nullptr;
    unsigned long long pub_seq = 0;
    double *jacobian_parameters = nullptr, *solution_sensitivity = nullptr;   // N*np
    // points
    double *solution = nullptr, *candidate = nullptr, *lambda = nullptr, *parameters = nullptr;
    double *residual = nullptr, *residual_error = nullptr, *step = nullptr, *step_correction = nullptr, *saved_point = nullptr;
    double *residual_symmetric = nullptr, *step_symmetric = nullptr, *merit_gradient = nullptr;
    bool pad_done = false;               // launch_scale_rows wrote the unit pivots of the padded rows of S for the launch_schur that follows
    double* Kdense = nullptr;  // n*n, allocated on first request
    // factorisation
    double* S = nullptr;        // NP*NP: Schur complement onto x, then L (unit lower) in place
    double* Dx = nullptr;       // NP: pivots of S
    double* refpart = nullptr;  // per workgroup of k_refine_local / k_solve_tail: its part of ||residual_error||_inf
    int refparts = 0;           // how many of them the last producer wrote (k_refine_x combines them)
    int* zgrp = nullptr; int n_zgrp = 0;   // k_solve_tail (vectors.hip): first row of every group of whole constraints (<= 16 rows of [gx; hx]); shape-only, outside the slab
    // speculative refinement rounds (api.hip: do_refinement): while gate_epoch != 0 the launchers of a round's kernels pass (gate, gate_epoch) and the kernels leave at
    // once when gate[0] == gate_epoch — "this refinement has converged" (set by the residual kernel that saw it), so rounds queued ahead of the host's knowledge cost nothing
    int* gate = nullptr; int gate_epoch = 0; int gate_counter = 0;
    bool refine_defer = false, refine_pending = false;   // api.hip: the speculative rounds of a refinement are queued, their report is read with a later read-back of the caller
    bool spec_ahead_ok = false;            // api.hip: inner_iteration may queue IC-1 ahead of its exit tests (the last step went on to a search direction, far from the thresholds)
    bool refine_local_done = false;        // the solve tail just queued also formed the local rows of the refinement residual: the next launch_refine_local is a no-op
    double* Ypanel = nullptr;   // NP*NB: M_k = (L_kk D_k L_kk')^-1 of every 64-column panel (ldl.hip: what the trailing update multiplies the raw panel with)
    double* Tinv = nullptr;     // tinv_doubles(NP): inverses of the unit-lower diagonal blocks of L (up to 1024 x 1024, the last one may be 512 wide)
    double* Ttmp = nullptr;     // NP*1024 scratch of the inverse assembly (NP*512: one 1024 x 1024 product at the top level; as much again so that every merge level has an area of its own, ldl.hip: merge_scratch)
    double* zf2 = nullptr;      // NP
    double* Wfac = nullptr;     // NP*NP/2: the W-form blocks of the triangular solves (wform_offset)
    double* WH = nullptr;       // nc*nx: Omega_z * hx
    double* wz = nullptr;       // nc: Omega for nonnegative entries (-1/K_zz)
    double* kzz = nullptr;      // nc: K_zz diagonal for nonnegative entries
    double* Wsoc = nullptr;     // sum d^2: (-B_sym)^-1 per SOC
    double* Bsoc = nullptr;     // sum d^2: the reference's (non-symmetric) K_zz SOC block
    double* socwork = nullptr;  // 2 * sum d^2 scratch
    int schur_nj = 8;           // Schur tile = 128 x 16*schur_nj for single-instance launches (schur.hip: schur_plan)
    // stage-banded structure (structure.hip); band64 = 0: dense
    int half_bandwidth = 0, band64 = 0;
    // stage-parallel factorisation of S (calipso_hip_set_stage_parallel): the skyline pattern of S found by the structure analysis, a multifrontal
    // sparse LDL^T over its nested-dissection tree (sparse.hip) instead of the blocked LDL^T of ldl.hip, for this handle or the group it leads
    std::vector<int> h_reach;                 // per column j of S: last row that can be non-zero (analysis)
    std::vector<int> h_zrow, h_lreach;        // analysis, host copies: per row of [gx; hx] its [first, last + 1) column; per column of Lxx the last row its entries reach
    calipso::StageBlocks blocks;              // calipso_hip_set_stage_blocks (blocks.hip)
    bool blocks_effective = true;             // false while a group launch covers members whose block structures differ
    size_t blocks_zero_cell = 0;              // structured handle: offset (doubles, in S) of a cell that stays zero
    bool compact = false;                     // structured handle (calipso_hip_create_structured): no dense Lxx / [gx; hx] / S / Tinv exist, only the blocks
    calipso_hip_sparse* spS = nullptr;
    long long* spS_src = nullptr;             // device: offset (row + col * NP) in S of every pattern entry
    bool spS_values_current = false;          // k_schur_blocks has just written the multifrontal values of the active instances (consumed by launch_ldl)
    int* spS_inv = nullptr;                   // structured handle: for every cell of the packed S the pattern entry it is (-1: none) — k_schur_blocks writes the multifrontal values itself
    int* d_reach = nullptr;                   // device copy of h_reach while the stage-parallel factorisation is on: uploads are re-checked against the skyline
    bool stage_parallel = false;
    calipso::i64 structure_resets = 0;   // uploads that broke an analysed structure (the handle went back to dense)
    int* krange = nullptr;      // (in the slab) per 16-column group: [eq_lo, eq_hi, cone_lo, cone_hi) constraint rows that touch it
    int* zrow = nullptr;        // (in the slab) per row of [gx; hx]: [first, last + 1) non-zero column
    int* icount = nullptr;      // device ints: [0] pos [1] nonpos [2] zero (constraint part), [3..5] same for S, [6..] cone-search masks
    int* hicount = nullptr;     // pinned host mirror
    int* hicount_dev = nullptr;
    double* gemv_partial = nullptr;   // partial sums for column-split mat-vecs
    double* vtmp = nullptr;     // 4*N scratch vectors
    double *xbuf = nullptr, *zf = nullptr, *t1 = nullptr, *t2 = nullptr;   // NP, NP, m, m: work vectors of the condensed solve
    double *zsx = nullptr;                                                  // m: [gx; hx] step_x, kept up to date across refinement rounds (vectors.hip)
    double *w1 = nullptr, *w2 = nullptr, *lxv = nullptr;                    // nx each: [gx; hx]'step_yz, [gx; hx]'(Omega b_m), Lxx step_x
    double *saved_g = nullptr, *saved_h = nullptr;                          // ne, nc (benchmark-mode restore)
    double *lgp = nullptr, *gp = nullptr, *hp = nullptr;                    // nx*np, ne*np, nc*np parameter Jacobians
    std::vector<double> hparams;
    double* multi_rhs = nullptr;  // workspace of the multi-right-hand-side solve of differentiate! (allocated on demand)
    double* dsym_multi = nullptr; // n * np
    void* scatter_aux = nullptr;  // scatter.hip: registered sparsity patterns of the evaluate! scatter
    void* ldl_aux = nullptr;      // ldlsolver.hip: staging of the caller's CSC matrix (handles made by calipso_hip_ldl_create)
    void* lfac_aux = nullptr;     // lfac.hip: plan and buffers of the left-looking schedule (one dense system alone)
    bool lfac_last = false;       // the last blocked factorisation took it
    bool lfac_failed = false;     // its buffers could not be had: the handle keeps k_schur + the right-looking panel steps
    double* Lf = nullptr;         // where the factor columns L of the last blocked factorisation live: S (scaled in place) or lfac.hip's buffer
    double* Hdense = nullptr; int* lu_ipiv = nullptr;   // fallback.hip: N x N unreduced matrix and pivots (allocated on first use)
    hipEvent_t ev[16];
    hipStream_t stream2 = nullptr;       // second stream of the handle: the finish of completed solve blocks while the pivot chain runs (ldl.hip)
    bool ldl_publish = false;            // launch_ldl: the last diagonal block may publish the inertia counts ...
    unsigned long long ldl_pub_seq = 0;  // ... and did, under this sequence number (0: it did not; read them back)
    bool factor_times_pending = false;   // the events of the last factorisation have not been read yet (api.hip: factor_times)
    bool ldl_overlap_on = false;         // the decision enqueue_ldl_steps took for the factorisation in progress (second stream ready, ranges planned): enqueue_ldl_finish follows it
    bool ldl_failed = false;             // launch_ldl could not factor (a structured handle whose multifrontal path refused): do_factorize reports it
    int ldl_step_launches = 0;           // panel-step launches (k_ldl_diag + k_ldl_step) of the last blocked factorisation
    int ldl_forks = 0;                   // feeds the last enqueue_ldl_steps left to the second stream (the first ldl_forks of ldl_feeds); the ranges of columns they refer to: ...
    std::vector<int> ldl_ranges;         // ... (first column, width) pairs, in order; range f may be finished once panel step (c0 + w) / 64 has STARTED
    std::vector<int> ldl_feeds;          // (panel step that must have started, kind, argument) triples in hand-over order; kind 0: finish of range `argument`, 1: W-form product of solve block `argument`
    unsigned long long ldl_epoch = 0;    // factorisations so far (tags the progress word)
    unsigned long long *hprog = nullptr, *hprog_dev = nullptr;   // mapped host word: epoch << 16 | index of the last panel step that started
    hipEvent_t ev_side[8] = {};          // [7]: the join (second stream -> main)
    hipGraphExec_t graph_ldl = nullptr, graph_ldl_fin = nullptr, graph_trsv = nullptr;   // captured once per handle (fixed launch sequences): panel steps, factor columns + block inverses, one triangular solve
    bool graph_ldl_tried = false, graph_ldl_fin_tried = false, graph_trsv_tried = false, use_graphs = true;
    calipso::i64 solve_block = 1024;   // "opt.solve_block": widest diagonal block of L whose inverse is assembled (1024: fewest launches per solve, what one system wants; 512: a quarter of the
                                                        // inverse-assembly flops, what a group wants — its solves are bandwidth-bound).  Members of a group use the leader's.
    calipso::i64 solve_wform = 1;      // "opt.solve_wform": the triangular solves through the stacked [Tinv_b; W_b] blocks (internal.hpp: wform_offset): 2 launches per solve block instead of 4.
                                       // ONE system wants it (its solves are chains of latency-bound launches); a group, whose solves are bandwidth-bound, does not need the extra products.
    bool time_matvec = false, matvec_timed = false;   // the next gemv_refine_pair brackets its mat-vec launch with ev[5] / ev[6] (once per Newton step: calipso_hip_kernel_times [4])
    double kernel_ms[4] = {0};   // [0] the panel-step launches (k_ldl_diag + k_ldl_step) of the last factorisation
    double phase_ms[9] = {0};
    // filter (filter.jl:1-13), host side
    std::vector<double> filter_theta, filter_merit, cache_theta, cache_merit;
    calipso::i64 filter_index = 0;
    // host staging
    std::vector<double> hpoint, hstage;
};

namespace calipso {

// ---- launchers (each enqueues on s->stream; no host synchronisation unless stated) -------------------------------
// cones.hip
void launch_cone(calipso_hip_solver* s, const double* point, int flags);
// fills icount[6..] violation bit masks for alpha = scaling_line_search^k (publish_seq != 0, a single handle: the kernel itself publishes the masks + this sequence number)
void launch_cone_search(calipso_hip_solver* s, unsigned long long publish_seq = 0);
void launch_cone_candidate(calipso_hip_solver* s, double a_s, double a_t);
void launch_cone_candidate_batch(calipso_hip_solver* s, const double* a_s, const double* a_t);   // one pair per covered instance
void launch_cone_violation_host(calipso_hip_solver* s, const double* xhat_dev, const double* x_dev, double tau);
// vectors.hip
void launch_residual(calipso_hip_solver* s);
void launch_violations(calipso_hip_solver* s, int pub_first = 0, int pub_count = 0);   // pub_count > 0 (single handle): the kernel also publishes dscal[pub_first ..) for the read-back behind it                  // -> dscal[8..]
void launch_residual_symmetric(calipso_hip_solver* s, const double* res);   // also fills xbuf (b_x, zero padded) and t1 = Omega b_m
// back-substitution + recovery (+ accumulate += step); zsx_mode 0: leave zsx, 1: zsx = [gx; hx] dx (= t2), 2: zsx += t2
void launch_recover(calipso_hip_solver* s, double* step, const double* res, double* accumulate, int zsx_mode = 0);
// one refinement residual in two kernels around the mat-vecs (see vectors.hip): rows r, s, y, z, t of residual_error = residual - H step from
// zsx, the condensed b_m and t1 = Omega b_m, partial norm -> dscal[18]; then the x rows, dscal[7] = ||residual_error||_inf, xbuf = [b_x + w2; 0]
void launch_refine_local(calipso_hip_solver* s);
// t2 = [gx; hx] dx + launch_recover (+ launch_refine_local when with_refine) in ONE launch; false: not available for this handle (the caller takes the separate launches)
bool launch_solve_tail(calipso_hip_solver* s, int which, bool accumulate, bool with_refine);
bool solve_tail_available(const calipso_hip_solver* s);
void solve_tail_plan(const Dims& d, const std::vector<int>& soc_start, const std::vector<int>& soc_dim, std::vector<int>& grp);
void launch_refine_x(calipso_hip_solver* s, bool publish = false);
void launch_refine_x_fused(calipso_hip_solver* s, bool publish, int nchunk, int it = -1, bool last_queued = false);   // it >= 0: speculative round index (see gate)
void launch_trsv_direct(calipso_hip_solver* s, double* x);      // launch_trsv without the captured graph (its kernels then carry the handle's current gate)   // the same with the reduction of the nchunk partial sums of Lxx step_x (gemv_refine_pair) folded in   // publish: dscal[7] also to the handle's mapped host mirror + sequence number
void launch_axpy_points(calipso_hip_solver* s, double step_size, int with_s);
void launch_accept(calipso_hip_solver* s, double step_size);
void launch_axpy_points_batch(calipso_hip_solver* s, const double* step_size, int with_s);
void launch_accept_batch(calipso_hip_solver* s, const double* step_size);
void launch_merit(calipso_hip_solver* s, const double* point);  // -> dscal[4] (M), uses dscal[0], dscal[1]
void launch_merit_gradient(calipso_hip_solver* s);
void launch_merit_and_gradient(calipso_hip_solver* s);      // merit at the current point + its gradient, one launch
void launch_first_candidate(calipso_hip_solver* s, double a_s, double a_t);                        // first candidate of the line search + directional derivative, one launch
void launch_first_candidate_batch(calipso_hip_solver* s, const double* a_s, const double* a_t);
void launch_first_candidate_from_masks(calipso_hip_solver* s);                                     // the step sizes from the cone-search masks on the device (no host round trip)
void launch_constraint_violation(calipso_hip_solver* s, const double* point, int pub_first = 0, int pub_count = 0);   // -> dscal[5]
void launch_merit_and_constraint(calipso_hip_solver* s, const double* point, int pub_first, int pub_count);   // k_merit + k_constraint_violation in one launch
void launch_violations_and_constraint(calipso_hip_solver* s, int pub_first = 0, int pub_count = 0);   // both at the current point, one launch
void launch_dot_merit(calipso_hip_solver* s);                   // -> dscal[6]
void launch_Hmul(calipso_hip_solver* s, const double* v, double* out);   // out = H v
void launch_residual_error(calipso_hip_solver* s, const double* step);   // residual_error = residual - H step ; dscal[7] = inf-norm
void launch_add(calipso_hip_solver* s, double* y, const double* x, int len);    // y += x
void launch_assemble_K(calipso_hip_solver* s);
// gemv.hip
void gemv_n(calipso_hip_solver* s, int rows, int cols, const double* A, int ld, const double* x, double* y, double alpha, double beta, int kind = SP_DENSE,
            const double* add = nullptr);      // add: y = alpha A x + add (beta must be 0): the vector a caller would add in a launch of its own
void gemv_t(calipso_hip_solver* s, int rows, int cols, const double* A, int ld, const double* x, double* y, double alpha, double beta, int kind = SP_DENSE,
            const double* add = nullptr);      // (as for gemv_n)
void gemv_t2(calipso_hip_solver* s, int rows, int cols, const double* A, int ld, const double* x1, const double* x2, double* y1, double* y2, int kind = SP_DENSE);   // y1 = A'x1, y2 = A'x2, one pass
// gemv_t2 on [gx; hx] and gemv_n on Lxx in one launch.  defer_reduce: the column-chunk partial sums of Lxx xl stay in gemv_partial and their number is returned (> 0) for
// launch_refine_x_fused to combine; 0: yl holds the product
int gemv_refine_pair(calipso_hip_solver* s, const double* x1, const double* x2, double* y1, double* y2, const double* xl, double* yl, bool defer_reduce = false);
void gemv_both(calipso_hip_solver* s, int rows, int cols, const double* A, int ld, const double* x, const double* u, double* yn, double* yt, double beta_t,
               int kind = SP_DENSE);   // yn = A x, yt = A'u + beta_t*yt, one pass over A
// `kind` names the block (SP_Z, SP_GX, SP_HX, SP_LXX) so that a handle with an analysed structure skips its structural zeros
// schur.hip
void launch_cone_weights(calipso_hip_solver* s);
// soc_wide.hip: cones of dimension > 4, one wavefront per cone (no-ops on handles without such cones)
void launch_cone_weights_wide(calipso_hip_solver* s);
void launch_residual_symmetric_wide(calipso_hip_solver* s, const double* res, int p, double* rsym, double* t1);
void launch_recover_wide(calipso_hip_solver* s, const double* res, int p, const double* rsym, const double* t2, double* dsym, double* step, double* accumulate, int zsx_mode);
void launch_refine_local_wide(calipso_hip_solver* s, int part0);
void launch_scale_rows(calipso_hip_solver* s);
void launch_schur(calipso_hip_solver* s);
void launch_symmetrize(calipso_hip_solver* s);          // Lsym from the upper triangle of Lxx (for the covered instances)
void schur_plan(calipso_hip_solver* s);   // host: choose the tile shape of single-instance launches
// ldl.hip
void launch_pad_identity(calipso_hip_solver* s);
// blocks.hip: stage blocks.  The blocks_* launchers return false when the handle (or the group launch in progress) does not use blocks: the caller
// then takes the dense-layout kernel.
void blocks_release(calipso_hip_solver* s);
void blocks_pack(calipso_hip_solver* s, bool z, bool l);
int blocks_pack_from(calipso_hip_solver* s, const double* L, const double* Z, bool l, bool zg, bool zh);
int blocks_descriptors(calipso_hip_solver* s);                                   // (once) the calipso_device_block arrays of the handle's blocks, host and device
void blocks_mirror(calipso_hip_solver* s, bool l, bool zg, bool zh);             // second orientation of the blocks a block evaluator has just written   // zg / zh: the equality / cone Jacobian was written   // dense arrays -> blocks, with the check that nothing lies outside the structure (synchronises)
bool blocks_gemv_n(calipso_hip_solver* s, int kind, const double* x, double* y, double alpha, double beta);
bool blocks_gemv_t(calipso_hip_solver* s, int kind, const double* u1, const double* u2, double* y1, double* y2, double alpha, double beta);
bool blocks_schur(calipso_hip_solver* s);
bool blocks_gemm_n(calipso_hip_solver* s, const double* X, long long ldx, double* Y, long long ldy, int p);                  // Y(:, c) = [gx; hx] X(:, c), p columns
bool blocks_gemm_t(calipso_hip_solver* s, const double* U, long long ldu, double* Y, long long ldy, int p, double beta);     // Y(:, c) = [gx; hx]' U(:, c) + beta Y(:, c)
bool blocks_plan(const Dims& d, const std::vector<int>& zrow, const std::vector<int>& lreach, BlockPlan& P, std::string& err);
int blocks_install(calipso_hip_solver* s, const BlockPlan& P);
int blocks_unpack_dense(calipso_hip_solver* s, double** Lxx, double** Z);
int blocks_upload_dense(calipso_hip_solver* s, int which, const double* data, double scale);        // structured handles: dense host array -> packed blocks
int blocks_download_dense(calipso_hip_solver* s, int which, double* data);
bool blocks_entry_offsets(const calipso_hip_solver* s, int which, int row, int col, long long* off_c, long long* off_r);
void launch_ldl(calipso_hip_solver* s);
void ldl_drop_graphs(calipso_hip_solver* s);
void launch_trsv(calipso_hip_solver* s, double* x);            // x (length NP) <- S^-1 x using L, D
bool lastblock_sym_on(const calipso_hip_solver* s);   // the last solve block as one symmetric mat-vec (ldl.hip)
bool wform_on(const calipso_hip_solver* s);                    // the solves of this handle (or of the group launch in progress) go through the W-form blocks
// lfac.hip: the left-looking schedule of one dense system (the Schur complement's products under the pivot chain)
bool lfac_on(const calipso_hip_solver* s);
bool lfac_ready(calipso_hip_solver* s);          // lfac_on, and the plan / buffers exist
int lfac_enqueue(calipso_hip_solver* s, unsigned long long* hprog, unsigned long long epoch);   // returns the launches queued (0: not available)
double* lfac_factor_buffer(calipso_hip_solver* s);
void lfac_release(calipso_hip_solver* s);
void lfac_describe(calipso_hip_solver* s, double out[8]);
// solvek.hip
void launch_copy_pad(calipso_hip_solver* s, const double* src, int n, double* dst, int npad);
void launch_init_point(calipso_hip_solver* s);                 // initialize_slacks!/duals! (initialize.jl:15-36), r <- g
void launch_lambda_update(calipso_hip_solver* s);              // lambda += rho * r (solve.jl:362-364)
void launch_jacobian_parameters(calipso_hip_solver* s);        // residual_jacobian_parameters.jl:1-40
void launch_negate_copy(calipso_hip_solver* s, const double* src, double* dst, int n);
void linear_solve_device(calipso_hip_solver* s, bool with_t2 = true, bool rhs_ready = false);   // rhs_ready: xbuf already holds b_x + [gx; hx]'(Omega b_m)
hipStream_t ldl_rhs_stream(calipso_hip_solver* s);   // middle of the condensed solve (operands from k_residual_symmetric); with_t2 = false: the caller's k_solve_tail forms t2 = [gx; hx] dx
void launch_solve_from_b(calipso_hip_solver* s);               // step_symmetric = K^-1 residual_symmetric for a caller-provided b
// gemm.hip
void gemm(calipso_hip_solver* s, int M, int N, int K, double alpha, const double* A, int lda, bool transA, const double* B, int ldb, double beta,
          double* C, int ldc);
void trsm_multi(calipso_hip_solver* s, double* X, int p, double* U, double* Zm);
void launch_residual_symmetric_multi(calipso_hip_solver* s, const double* res, int p, double* rsym, double* xbuf, double* t1);
void launch_recover_multi(calipso_hip_solver* s, const double* res, int p, const double* rsym, const double* xbuf, const double* t2, double* step, double scale);
// fallback.hip
int nonsymmetric_solve(calipso_hip_solver* s, const double* res, double* step);   // step = H \\ res (pivoted LU of the unreduced matrix)
void nonsymmetric_release(calipso_hip_solver* s);
// group.hip
void group_member_destroyed(struct calipso_hip_group* g, calipso_hip_solver* s);   // called by calipso_hip_destroy on a member of a live group
// scatter.hip
void scatter_release(calipso_hip_solver* s);
// ldlsolver.hip
void ldlsolver_release(calipso_hip_solver* s);
// structure.hip
int structure_validate(calipso_hip_solver* s, int which);
bool csc_pattern_ok(i64 n, const i64* colptr, const i64* rowval);   // ordering.hip: colptr[0] == 1, monotone, nnz < 2^31, rows in 1..n
inline bool structure_active(const calipso_hip_solver* s) { return s->band64 > 0 || s->stage_parallel || s->blocks.on; }   // an analysed pattern that uploads must respect      // which: 0 Lxx, 1 gx, 2 hx; clears the structure when the block breaks it
// qp.hip
void launch_qp_evaluate(calipso_hip_solver* s, const double* point, uint32_t flags);

int check(calipso_hip_solver* s, hipError_t e, const char* what);
void inject_refused_launch(hipStream_t stream);      // test hook (CALIPSO_HIP_FAULT_INJECT=launch): a launch the runtime refuses, queued among the factorisation's
// hipFuncAttributeMaxDynamicSharedMemorySize for a kernel that wants more than 64 KB of dynamic LDS: function attributes are PER DEVICE, so once per (kernel, current
// device) — not once per process — and the result is kept: false = the attribute was refused there (the caller takes its fallback)
bool lds_attribute(const void* kernel, int bytes);
// evaluate!(problem, methods, idx, point, parameters; flags) at the current (which = 0) or candidate (1) point: the attached device
// evaluator, else the host callback (api.hip)
int evaluate_point(calipso_hip_solver* s, calipso_eval_fn eval, void* user, int which, uint32_t flags);
// the batch a launch on `s` covers: the group's active set if a group is driving `s`, else `s` alone with its own scalars
inline BatchSc batch_of(const calipso_hip_solver* s) {
    if (s->cur) return *s->cur;
    BatchSc b;
    b.b.n = 1; b.b.delta[0] = 0; b.sctab = nullptr; b.sc1 = s->sc;
    return b;
}
// batched fills / copies (vectors.hip) — replace hipMemsetAsync / hipMemcpyAsync on the hot path so that groups are covered
// sparse.hip hooks used by ldl.hip when a handle factors S through the multifrontal path
bool sparse_is_multifrontal(const calipso_hip_sparse* sp);
void sparse_borrow_stream(calipso_hip_sparse* sp, hipStream_t st);
int sparse_batch(const calipso_hip_sparse* sp);
int sparse_factor_from_dense(calipso_hip_sparse* sp, hipStream_t st, const Batch& bt, const double* S, const long long* src, int* icount, bool values_in_place = false);
void sparse_values(calipso_hip_sparse* sp, double** values, long long* stride);      // the batch x nnz values the factorisation reads (instance-major)
int sparse_solve_inplace(calipso_hip_sparse* sp, hipStream_t st, const Batch& bt, double* x);
int sparse_reserve_solve(calipso_hip_sparse* sp, int batch);
int sparse_solve_inplace_multi(calipso_hip_sparse* sp, hipStream_t st, int slot, double* X, long long ld, int p);
void sparse_describe(const calipso_hip_sparse* sp, int64_t out[4]);
void sparse_work(const calipso_hip_sparse* sp, double out[3]);      // flops of one numeric factorisation, nnz(L) of the fronts, order
int nested_dissection_pieces(i64 n, const i64* colptr, const i64* rowval, i64* perm, std::vector<std::pair<int, int>>& pieces);   // ordering.hip
void fill_d(calipso_hip_solver* s, double* p, size_t n, double v);
void fill_i(calipso_hip_solver* s, int* p, size_t n, int v);
void copy_d(calipso_hip_solver* s, double* dst, const double* src, size_t n);
void copy4_d(calipso_hip_solver* s, double* const dst[4], const double* const src[4], const size_t n[4]);   // four copies, one launch
}  // namespace calipso

#define CK(call)                                                        \
    do {                                                                \
        hipError_t e__ = (call);                                        \
        if (e__ != hipSuccess) return calipso::check(s, e__, #call);    \
    } while (0)
