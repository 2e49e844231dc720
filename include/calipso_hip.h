/*
 * calipso_hip.h — C ABI of libcalipso_hip.so: the MI355X (gfx950) implementation of CALIPSO's
 * per-iteration Newton/KKT hot path.
 *
 * The reference (thowell/CALIPSO.jl v0.1.1) is pure Julia and has no FFI; the seams this ABI replaces are
 * Julia functions and types (paths relative to src/solver/ of the reference):
 *   - exported API            Solver / initialize! / solve!            CALIPSO.jl:48-50, solver.jl:46-173,
 *                                                                      initialize.jl:9-13, solve.jl:8-377
 *   - linear-solver seam      LDLSolver, factorize!, compute_inertia!, linear_solve!   linear_solver.jl:1-60
 *   - per-function seams      cone!, residual!, residual_jacobian_variables(_symmetric)!, residual_symmetric!,
 *                             search_direction(_symmetric)!, iterative_refinement!, inertia_correction!,
 *                             cone_violation, merit, merit_gradient!, constraint_violation!, optimality_error,
 *                             differentiate!      (one entry point each, cited below)
 * INTEGRATION.md shows the Julia `ccall` binding for every entry point.
 *
 * Conventions
 *   - plain C: pointers + sizes, no C++/torch types; every function returns an int32 status.
 *   - all floating point is IEEE fp64; all indices are Int64 and 1-BASED (Julia's), matrices column-major.
 *   - the caller owns every host buffer; the opaque handle owns device memory, one HIP stream and its events (and, created on first use, a second
 *     stream that only ever carries the finish of a factorisation beside its pivot chain and is joined into the first before the call that
 *     queued it returns: everything a caller orders or synchronises against is the first stream).
 *   - one handle = one problem shape (nx, np, ne, nc + cone layout); a handle is used by one host thread at a time;
 *     distinct handles are independent (as distinct `Solver`s are in the reference).
 *   - host arrays are named by the reference's own field names ("equality_jacobian_variables", "solution", ...).
 *   - nothing here falls back to the CPU: without a usable HIP device every call fails with CALIPSO_ERR_HIP.
 */
#ifndef CALIPSO_HIP_H
#define CALIPSO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct calipso_hip_solver calipso_hip_solver;

/* status codes: negative = the reference's error() cases, positive = its @warn cases */
enum {
    CALIPSO_OK = 0,
    CALIPSO_ERR_INERTIA = -1,      /* error("inertia correction failure")  inertia.jl:72 */
    CALIPSO_ERR_CONE_SEARCH = -2,  /* error("cone search failure")         solve.jl:210,220 */
    CALIPSO_ERR_CALLBACK = -3,     /* the evaluation callback returned non-zero */
    CALIPSO_ERR_ARGUMENT = -4,     /* unknown field name, wrong length, null pointer */
    CALIPSO_ERR_HIP = -5,          /* HIP runtime error / no device; text in calipso_hip_last_error */
    CALIPSO_ERR_LAYOUT = -6,       /* cone index sets violate the layout the reference relies on (cones/cone.jl:27-59 vs residual.jl:46-48) */
    CALIPSO_WARN_ZERO_PIVOT = 1,   /* @warn "Zero entry in D (matrix is not quasidefinite)" qdldl.jl:309-311 */
    CALIPSO_WARN_REFINEMENT = 2,   /* @warn "iterative refinement failure" iterative_refinement.jl:50 */
    CALIPSO_WARN_LINE_SEARCH = 3   /* @warn "residual line search failure" solve.jl:301 */
};

/* evaluate! keyword flags (evaluate.jl:1-23), one bit each */
enum {
    CALIPSO_EVAL_OBJECTIVE = 1u << 0,
    CALIPSO_EVAL_OBJECTIVE_GRADIENT = 1u << 1,
    CALIPSO_EVAL_OBJECTIVE_HESSIAN = 1u << 2,
    CALIPSO_EVAL_EQUALITY = 1u << 3,
    CALIPSO_EVAL_EQUALITY_JACOBIAN = 1u << 4,
    CALIPSO_EVAL_EQUALITY_DUAL_GRADIENT = 1u << 5,
    CALIPSO_EVAL_EQUALITY_DUAL_HESSIAN = 1u << 6,
    CALIPSO_EVAL_CONE = 1u << 7,
    CALIPSO_EVAL_CONE_JACOBIAN = 1u << 8,
    CALIPSO_EVAL_CONE_DUAL_GRADIENT = 1u << 9,
    CALIPSO_EVAL_CONE_DUAL_HESSIAN = 1u << 10,
    CALIPSO_EVAL_OBJECTIVE_JACOBIAN_PARAMETERS = 1u << 11,
    CALIPSO_EVAL_EQUALITY_JACOBIAN_PARAMETERS = 1u << 12,
    CALIPSO_EVAL_EQUALITY_DUAL_JACOBIAN_PARAMETERS = 1u << 13,
    CALIPSO_EVAL_CONE_JACOBIAN_PARAMETERS = 1u << 14,
    CALIPSO_EVAL_CONE_DUAL_JACOBIAN_PARAMETERS = 1u << 15
};

/* cone! keyword flags (cones/cone.jl:71-77) */
enum {
    CALIPSO_CONE_BARRIER = 1 << 0,
    CALIPSO_CONE_BARRIER_GRADIENT = 1 << 1,
    CALIPSO_CONE_PRODUCT = 1 << 2,
    CALIPSO_CONE_JACOBIAN = 1 << 3, /* arrow/diagonal Jacobians are implicit in (s, t) on the device; kept for API parity */
    CALIPSO_CONE_TARGET = 1 << 4
};

/* User-evaluation callback: the stand-in for the generated functions evaluate! calls (evaluate.jl:37-121).
 * Called on the host with the point's x, y, z and the parameters; it must evaluate the flagged quantities and hand
 * them to the handle with calipso_hip_set_field (the three Hessian terms as ONE summed "lagrangian_hessian").
 * Return 0 on success. */
typedef int32_t (*calipso_eval_fn)(void* user, uint32_t flags, const double* x, const double* y, const double* z,
                                   const double* theta);

/* DEVICE-side user evaluation: the same contract as calipso_eval_fn, but every pointer is a DEVICE pointer into the
 * handle's own memory and the function only ENQUEUES work (HIP kernels, hipMemcpyAsync, hipBLAS ...) on the given stream — it must not
 * synchronise.  x, y, z point into the current / candidate point, theta to the parameters; `out` names where each flagged ProblemData
 * field has to be written (fields that do not exist for the shape are NULL).  With such an evaluator a whole solve! — including the
 * up to 25 backtracking re-evaluations of the residual line search, solve.jl:254-302 — runs without the point ever visiting the host
 * and without a block upload.  Return 0 on success. */
typedef struct calipso_device_problem_data {
    double* objective;                        /* [1] */
    double* objective_gradient_variables;     /* [nx] */
    double* equality_constraint;              /* [ne] */
    double* cone_constraint;                  /* [nc] */
    double* equality_dual_jacobian_variables; /* [nx]  (g'y)_x */
    double* cone_dual_jacobian_variables;     /* [nx]  (h'z)_x */
    double* lagrangian_hessian;               /* [nx*nx] column-major: objective Hessian + (g'y)_xx + (h'z)_xx summed (the latter two iff
                                                 options.constraint_tensor, residual_jacobian_variables.jl:10-16) */
    double* equality_jacobian_variables;      /* ne x nx column-major with leading dimension jacobian_ld */
    double* cone_jacobian_variables;          /* nc x nx column-major with leading dimension jacobian_ld */
    int64_t jacobian_ld;                      /* = ne + nc: the two Jacobians are stacked in one matrix on the device */
    double* lagrangian_gradient_parameters;   /* [nx*np] objective + equality-dual + cone-dual terms summed (residual_jacobian_parameters.jl:8-14) */
    double* equality_jacobian_parameters;     /* [ne*np] */
    double* cone_jacobian_parameters;         /* [nc*np] */
    int64_t nx, np, ne, nc;
} calipso_device_problem_data;
typedef int32_t (*calipso_device_eval_fn)(void* user, uint32_t flags, const double* x, const double* y, const double* z, const double* theta,
                                          const calipso_device_problem_data* out, void* hip_stream);

/* The same for a STRUCTURED handle (calipso_hip_create_structured), without the dense interchange arrays: the generated functions of evaluate! (evaluate.jl:37-121)
 * write the non-zeros of the Jacobians and of the Hessian where the sparsity lists say (methods.*_sparsity, src/trajectory_optimization/sparsity.jl:28-129) — here
 * straight into the handle's packed blocks.  A block = rows [row0, row0 + nrows) x columns [col0, col0 + ncols) (0-based; rows numbered in the stacked matrix
 * [equality; cone]: cone row k is row ne + k), values column-major with leading dimension ld, a DEVICE pointer into the handle's own storage.  Hessian blocks are the
 * diagonal blocks of the Lagrangian Hessian (row0 = col0, nrows = ncols).  The descriptor arrays exist twice: on the host (`*_blocks`) and on the device
 * (`*_blocks_device`, for kernels that walk them).  An evaluator asked for a Jacobian / the Hessian writes EVERY entry of every block of it (zeros included);
 * nothing outside the blocks exists, so nothing can be written there.  The vector fields are those of calipso_device_problem_data.  With such an evaluator a
 * structured handle never allocates the nx^2 + (ne + nc) nx doubles of dense scratch that calipso_device_eval_fn costs it. */
typedef struct calipso_device_block { int64_t row0, nrows, col0, ncols; double* values; int64_t ld; } calipso_device_block;
typedef struct calipso_device_block_data {
    double* objective; double* objective_gradient_variables; double* equality_constraint; double* cone_constraint;
    double* equality_dual_jacobian_variables; double* cone_dual_jacobian_variables;
    double* lagrangian_gradient_parameters; double* equality_jacobian_parameters; double* cone_jacobian_parameters;
    int64_t nx, np, ne, nc;
    int64_t n_jacobian_blocks; const calipso_device_block* jacobian_blocks; const calipso_device_block* jacobian_blocks_device;
    int64_t n_hessian_blocks; const calipso_device_block* hessian_blocks; const calipso_device_block* hessian_blocks_device;
} calipso_device_block_data;
typedef int32_t (*calipso_device_block_eval_fn)(void* user, uint32_t flags, const double* x, const double* y, const double* z, const double* theta,
                                                const calipso_device_block_data* out, void* hip_stream);

/* callback_inner(custom, solver) / callback_outer(custom, solver)  (solver.jl:183,193; called at solve.jl:350,371) */
typedef void (*calipso_callback_fn)(void* user, calipso_hip_solver* solver);

/* ---- life cycle ------------------------------------------------------------------------------------------------ */
/* Solver(methods, nx, np, ne, nc; nonnegative_indices, second_order_indices)   solver.jl:46-150, indices.jl:20-63.
 * nonneg_idx: n_nonneg cone-local indices (1-based); soc_ptr: n_soc+1 zero-based offsets into soc_idx (1-based,
 * first entry of each cone is its head).  The sets must be [1..q] followed by contiguous SOC blocks in order
 * (the only layout for which the reference is self-consistent), else CALIPSO_ERR_LAYOUT.
 * LIMIT: a second-order cone may have at most 1024 entries (csrc/soc_wide.hip: one wavefront per cone, up to sixteen entries per lane; the reference has none,
 * cones/second_order.jl:1-69 — its largest is 12): wider cones are refused with CALIPSO_ERR_ARGUMENT and a message in calipso_hip_last_error(NULL). */
int32_t calipso_hip_create(int64_t nx, int64_t np, int64_t ne, int64_t nc, int64_t n_nonneg, const int64_t* nonneg_idx,
                           int64_t n_soc, const int64_t* soc_ptr, const int64_t* soc_idx, int32_t device,
                           calipso_hip_solver** out);
/* Solver(...) for a stage-structured problem whose sparsity is known up front — as the reference's are (methods.*_sparsity, src/trajectory_optimization/
 * sparsity.jl:28-129): constraint row k of [equality; cone] touches columns row_first[k]..row_last[k] (1-based, inclusive; first > last: an empty row; the
 * rows of a second-order cone share the union of their ranges), the Lagrangian Hessian is block diagonal with blocks starting at hessian_block_start[0] = 1 <
 * hessian_block_start[1] < ... .  The handle holds ONLY the blocks (packed contiguously, both orientations), the tiles of the Schur complement that the
 * blocks couple, and the multifrontal factor of the Schur complement: O(stages x block^2) device memory — none of the dense nx^2 / (ne + nc) nx / NP^2 buffers of
 * calipso_hip_create exist.  Uploads: calipso_hip_set_field with the dense host arrays of ProblemData (packed on the host; a non-zero outside the declared
 * structure is an error), calipso_hip_set_sparsity + calipso_hip_scatter_field / _hessian (straight into the blocks), calipso_hip_qp_attach.  Everything
 * else of this header works as on a dense handle — calipso_hip_differentiate included (differentiate.jl:1-61: the products with [gx; hx] block by block for all
 * parameter columns at once, the solves through the fronts for all columns together) and device evaluators (they write the dense ProblemData layout: on such a
 * handle into dense scratch arrays — nx^2 + (ne + nc) nx doubles, allocated on the first evaluation — whose entries go into the blocks behind the evaluator; a
 * non-zero outside the declared structure is an error) — except: no calipso_hip_analyze_structure / clear_structure / set_stage_parallel(off) /
 * set_stage_blocks(off) (the structure is fixed); members of a group must share one structure.  The Hessian must have at least two diagonal blocks (dynamics whose y'f Hessian couples x_t with
 * x_{t+1}, e.g. implicit integrators, declare one block and are refused: use calipso_hip_create + calipso_hip_analyze_structure for those). */
int32_t calipso_hip_create_structured(int64_t nx, int64_t np, int64_t ne, int64_t nc, int64_t n_nonneg, const int64_t* nonneg_idx, int64_t n_soc,
                                      const int64_t* soc_ptr, const int64_t* soc_idx, int32_t device, const int64_t* row_first, const int64_t* row_last,
                                      int64_t n_hessian_blocks, const int64_t* hessian_block_start, calipso_hip_solver** out);
int32_t calipso_hip_destroy(calipso_hip_solver*);
const char* calipso_hip_last_error(calipso_hip_solver*); /* never NULL */
const char* calipso_hip_version(void);
int32_t calipso_hip_device_count(void);

/* ---- data movement (host <-> handle), by the reference's field names ----------------------------------------- */
/* ProblemData (problem_data.jl:2-31): "objective"[1] "objective_gradient_variables"[nx] "equality_constraint"[ne]
 *   "equality_jacobian_variables"[ne*nx] "equality_dual_jacobian_variables"[nx] "cone_constraint"[nc]
 *   "cone_jacobian_variables"[nc*nx] "cone_dual_jacobian_variables"[nx] "lagrangian_hessian"[nx*nx]
 *   ( = objective_jacobian_variables_variables + equality_dual_..._variables + cone_dual_..._variables,
 *     residual_jacobian_variables.jl:10-16 )  "cone_product"[nc] "cone_target"[nc] "barrier"[1] "barrier_gradient"[nc]
 *   "jacobian_parameters"[N*np] ( = dR/dtheta, residual_jacobian_parameters.jl:1-40 )
 * Points / SolverData (point.jl, solver_data.jl): "solution"[N] "candidate"[N] "residual"[N] "residual_error"[N]
 *   "step"[N] "step_correction"[N] "residual_symmetric"[n] "step_symmetric"[n] "merit_gradient"[n]
 *   "jacobian_variables_symmetric"[n*n] (dense K, after calipso_hip_residual_jacobian_variables_symmetric)
 *   "solution_sensitivity"[N*np] "parameters"[np] "dual"[ne]
 * Scalars (solver.jl:81-127): "central_path" "fraction_to_boundary" "penalty" "primal_regularization"
 *   "primal_regularization_last" "dual_regularization";  options (options.jl:6-59): "opt.<field>" (as double).  One option has no counterpart in
 *   the reference: "opt.solve_block" (512, 1024 or 2048, default 1024) — the widest diagonal block of the factor of S whose inverse is assembled for the
 *   triangular solves.  It changes the summation order of the solves (not the factor): 1024 suits one system (fewer dependent launches), 512 suits
 *   a group (one merge level less; its solves are bandwidth-bound); 2048 is there for larger systems (at nx = 2432 its extra merge level costs 0.15 ms
 *   and saves 0.07).  Set it on every member of a group (the first member's value governs the group's launches).  And "opt.solve_wform" (0 or 1,
 *   default 1): the solves (linear_solve!, linear_solver.jl:52-60) multiply with the stacked blocks [Tinv_b; W_b], W_b = L[below, b] Tinv_b formed once per
 *   factorisation, so that a solve is two dependent launches per solve block instead of four — and ONE for the last block, which has nothing below it and goes
 *   through the symmetric inverse Tinv' D^-1 Tinv of its Schur complement; same factor, different summation order.  One system wants it; a group
 *   (bandwidth-bound solves) does not need the extra products.  The members of a group must agree on it. */
int32_t calipso_hip_set_field(calipso_hip_solver*, const char* name, const double* data, int64_t len);
int32_t calipso_hip_get_field(calipso_hip_solver*, const char* name, double* data, int64_t len);
/* The scatter of evaluate! on the device (evaluate.jl:37-121): register `methods.<field>_sparsity` once — `count` (row, col)
 * pairs, 1-based, in cache order, duplicates allowed — then hand over only the value caches of an evaluation: O(nnz) over PCIe instead of the
 * dense blocks.  Semantics of the reference: plain assignment in list order, so the LAST writer of a repeated entry wins (the trajectory layer
 * repeats entries across stages); the three Hessian matrices are assigned separately and summed into "lagrangian_hessian"
 * (residual_jacobian_variables.jl:10-16).  Fields: "objective_jacobian_variables_variables", "equality_dual_jacobian_variables_variables",
 * "cone_dual_jacobian_variables_variables" (calipso_hip_scatter_hessian; a NULL part is "not re-evaluated in this call": as in evaluate.jl:37-42
 * it keeps the values of its last scatter — zeros if it never had one; count = 0 in calipso_hip_set_sparsity drops a part),
 * "equality_jacobian_variables", "cone_jacobian_variables" (calipso_hip_scatter_field). */
int32_t calipso_hip_set_sparsity(calipso_hip_solver*, const char* field, int64_t count, const int64_t* rows, const int64_t* cols);
int32_t calipso_hip_scatter_field(calipso_hip_solver*, const char* field, const double* values, int64_t count);
int32_t calipso_hip_scatter_hessian(calipso_hip_solver*, const double* objective_values, int64_t n_objective, const double* equality_dual_values,
                                    int64_t n_equality_dual, const double* cone_dual_values, int64_t n_cone_dual);
/* Indices (indices.jl:1-63) as the handle computed them, 1-based; returns the length or a negative status */
int64_t calipso_hip_get_index(calipso_hip_solver*, const char* name, int64_t* out, int64_t cap);

/* ---- the hot path, one entry point per reference function ---------------------------------------------------- */
/* cone!(problem, methods, idx, point; barrier, barrier_gradient, product, jacobian, target)  cones/cone.jl:71-106
 * which: 0 = solution, 1 = candidate */
int32_t calipso_hip_cone(calipso_hip_solver*, int32_t which, int32_t flags);
/* residual!(data, problem, idx, solution, kappa, rho, lambda)  residual.jl:1-51 */
int32_t calipso_hip_residual(calipso_hip_solver*);
/* the reductions solve! takes of the residual and iterate (solve.jl:130-135, optimality_error.jl:1-27):
 * out[0]=||R||_p/N (p = opt.residual_norm) out[1]=optimality_error out[2]=max(||R_y||inf,||R_z||inf)
 * out[3]=||equality_constraint||inf out[4]=||cone_product||inf */
int32_t calipso_hip_violations(calipso_hip_solver*, double out[5]);
/* residual_jacobian_variables! + residual_jacobian_variables_symmetric!  (residual_jacobian_variables.jl:1-167)
 * materialised as the dense n x n K for inspection ("jacobian_variables_symmetric"); the factorisation below works
 * from the blocks and never needs this. */
int32_t calipso_hip_residual_jacobian_variables_symmetric(calipso_hip_solver*);
/* out = H * v with the unreduced Jacobian H (never materialised): mul! of iterative_refinement.jl:9,39; v, out host[N] */
int32_t calipso_hip_jacobian_variables_mul(calipso_hip_solver*, const double* v, double* out);
/* factorize!(linear_solver, K) + compute_inertia!(linear_solver)   linear_solver.jl:19-44, qdldl.jl:269-317,400-589
 * LDL^T of P K P' in the constraint-first order [z | y | x]; inertia = (positive, negative, zero).
 * Returns CALIPSO_WARN_ZERO_PIVOT if an exact zero pivot was met (positive is then -1, as qdldl.jl:456,579). */
int32_t calipso_hip_factorize(calipso_hip_solver*, int64_t inertia[3]);
/* inertia_correction!(solver)   inertia.jl:30-80  (IC-1..IC-6 incl. the reference's quirks); n_factorizations may be NULL */
int32_t calipso_hip_inertia_correction(calipso_hip_solver*, int64_t* n_factorizations);
/* residual_symmetric!  residual.jl:53-101; which: 0 residual, 1 residual_error */
int32_t calipso_hip_residual_symmetric(calipso_hip_solver*, int32_t which);
/* linear_solve!(solver, x, K, b; fact=false): step_symmetric = K \ residual_symmetric with the current factors
 * (linear_solver.jl:52-60, qdldl.jl:330-351).  The reference's hidden re-factorisation (fact=true) is NOT replicated:
 * the matrix is unchanged between calipso_hip_factorize and the solves, so the factors are reused. */
int32_t calipso_hip_linear_solve(calipso_hip_solver*);
/* search_direction_symmetric!  search_direction.jl:25-104; which: 0 (step <- residual), 1 (step_correction <- residual_error) */
int32_t calipso_hip_search_direction_symmetric(calipso_hip_solver*, int32_t which);
/* iterative_refinement!(step, solver)  iterative_refinement.jl:1-52; returns CALIPSO_WARN_REFINEMENT on failure.
 * rounds / final_norm may be NULL */
int32_t calipso_hip_iterative_refinement(calipso_hip_solver*, int32_t* rounds, double* final_norm);
/* search_direction!(solver)  search_direction.jl:1-23 = inertia correction + condensed solve + refinement (+ pivoted
 * dense fallback on the unreduced system when refinement fails, replacing `H \ R` of :113) */
int32_t calipso_hip_search_direction(calipso_hip_solver*);
/* search_direction_nonsymmetric!(step, H, residual, ...)  search_direction.jl:106-119 (`step .= H \ residual`): partially pivoted
 * dense LU of the unreduced N x N matrix (assembled on the device on demand).  "step" <- H^-1 "residual".  Exception path. */
int32_t calipso_hip_search_direction_nonsymmetric(calipso_hip_solver*);
/* cone fraction-to-boundary search  solve.jl:190-221 with cone_violation (cones/cone.jl:62-68): writes candidate s, t
 * and returns the two step sizes (opt.scaling_line_search^k, k = number of shrinkings).  CALIPSO_ERR_CONE_SEARCH once k would exceed
 * opt.max_cone_line_search (<= 831). */
int32_t calipso_hip_cone_search(calipso_hip_solver*, double* step_size, double* step_size_cone_slack_dual);
/* cone_violation(xhat, x, tau, ...)  cones/cone.jl:62-68 on host vectors of length nc; *violated = 0/1 */
int32_t calipso_hip_cone_violation(calipso_hip_solver*, const double* xhat, const double* x, double tau, int32_t* violated);
/* candidate x, r (and s when with_cone_slack) = solution - step_size * step   solve.jl:224-229, 268-276 */
int32_t calipso_hip_candidate(calipso_hip_solver*, double step_size, int32_t with_cone_slack);
/* merit(f, r, Phi, kappa, lambda, rho)  merit.jl:2-15 on point `which` (uses "objective" and "barrier") */
int32_t calipso_hip_merit(calipso_hip_solver*, int32_t which, double* M);
/* merit_gradient!  merit.jl:17-31 (solution point) */
int32_t calipso_hip_merit_gradient(calipso_hip_solver*);
/* constraint_violation!(c, g, r, h, s, idx; norm_type)  constraint_violation.jl:1-13 on point `which` */
int32_t calipso_hip_constraint_violation(calipso_hip_solver*, int32_t which, double* theta);
/* d = dot(merit_gradient, step.primals) used by switching_condition / armijo (line_search.jl:3,16) */
int32_t calipso_hip_merit_directional(calipso_hip_solver*, double* d);
/* accept the candidate: x,r,s <- candidate; y,z -= step_size*step; t <- candidate t   solve.jl:309-326 */
int32_t calipso_hip_accept(calipso_hip_solver*, double step_size);

/* ---- drivers ----------------------------------------------------------------------------------------------------- */
/* initialize!(solver, guess)  initialize.jl:9-13 */
int32_t calipso_hip_initialize(calipso_hip_solver*, const double* guess);
/* solve!(solver)  solve.jl:8-377: returns 1 (true), 0 (false) or a negative status.  eval may be NULL when a device
 * evaluator is attached (calipso_hip_qp_attach). */
int32_t calipso_hip_solve(calipso_hip_solver*, calipso_eval_fn eval, void* user);
/* differentiate!(solver)  differentiate.jl:1-61: dR/dtheta assembled on the device, one factorisation, then one condensed
 * solve + recovery per parameter column (as differentiate.jl:29-58, without its per-column re-factorisation) */
int32_t calipso_hip_differentiate(calipso_hip_solver*, calipso_eval_fn eval, void* user);
/* install a device-side evaluator (NULL removes it): calipso_hip_solve / calipso_hip_differentiate / the group drivers then call it instead
 * of the host callback (their `eval` argument may be NULL), and calipso_hip_device_evaluate runs it on point `which` (0 solution, 1 candidate) */
int32_t calipso_hip_set_device_evaluator(calipso_hip_solver*, calipso_device_eval_fn fn, void* user);
/* structured handles only (CALIPSO_ERR_ARGUMENT otherwise): the evaluator writes the packed blocks (calipso_device_block_data above); replaces an evaluator set by
 * calipso_hip_set_device_evaluator.  evaluate.jl:37-121 */
int32_t calipso_hip_set_device_block_evaluator(calipso_hip_solver*, calipso_device_block_eval_fn fn, void* user);
int32_t calipso_hip_device_evaluate(calipso_hip_solver*, int32_t which, uint32_t flags);
/* install the per-inner-iteration / per-outer-update callbacks (NULL disables; options.callback_inner/outer) */
int32_t calipso_hip_set_callbacks(calipso_hip_solver*, calipso_callback_fn inner, calipso_callback_fn outer, void* user);
/* statistics of the last solve: [total_iterations, outer, factorizations, refinement_failures, max_refinement_rounds,
 * fallbacks, last_refinement_rounds, newton_steps] */
int32_t calipso_hip_stats(calipso_hip_solver*, int64_t out[8]);

/* ---- device-resident conic QP evaluator (synthetic benchmark inputs) ------------------------------- */
/* min c*x'Px + q'x  s.t. Ax - b = 0, h - Gx in K.  Host arrays are copied once; afterwards evaluate! runs as device
 * mat-vecs and no host callback is needed (eval = NULL).  P must be symmetric. */
int32_t calipso_hip_qp_attach(calipso_hip_solver*, const double* P, const double* q, const double* A, const double* b,
                              const double* G, const double* h, double objective_scale);
/* evaluate!(...; flags) with the attached evaluator on point `which` */
int32_t calipso_hip_qp_evaluate(calipso_hip_solver*, int32_t which, uint32_t flags);
/* One inner Newton iteration of solve! (solve.jl:98-353) at the current iterate with fixed kappa/rho, device evaluator
 * attached; if `advance` is 0 the iterate is restored afterwards (benchmark mode: every step does identical work).
 * info[0]=step_size info[1]=step_size_t info[2]=refinement rounds info[3]=factorizations info[4]=M_candidate info[5]=theta_candidate */
int32_t calipso_hip_newton_step(calipso_hip_solver*, int32_t advance, double info[6]);
/* `count` such steps in one call — the loop a caller would write around calipso_hip_newton_step (solve.jl:107-377 runs its iterations without leaving the
 * solver either): info = count x 6 doubles (may be NULL), status = count return codes; stops at the first failing step (its code is returned).  The stream is
 * synchronised and the phase timers are read after the LAST step only. */
int32_t calipso_hip_newton_steps(calipso_hip_solver*, int32_t count, int32_t advance, double* info, int32_t* status);
/* ---- groups: several handles of one shape stepped in lockstep through the same kernel launches ------------------------------
 * The reference has no batching: distinct `Solver`s are simply independent; BASELINE config C4 runs many of them
 * per GPU.  A group covers up to 128 handles created with identical dimensions and cone layout on one device; every launch of a
 * group step carries all members (the instance is a grid dimension), so the latency-bound parts of the step cost the same for
 * the whole group as for one handle.  Per member the arithmetic is exactly that of calipso_hip_newton_step.  Members stay
 * usable through the single-handle entry points between group calls (not concurrently with them). */
typedef struct calipso_hip_group calipso_hip_group;
int32_t calipso_hip_group_create(calipso_hip_solver** handles, int32_t count, calipso_hip_group** out);
int32_t calipso_hip_group_destroy(calipso_hip_group*);
/* calipso_hip_newton_step for every member: info = count x 6 doubles (row per member, fields as above), status = count codes */
int32_t calipso_hip_group_newton_step(calipso_hip_group*, int32_t advance, double* info, int32_t* status);
/* solve!(solver) (solve.jl:8-377) for every member in lockstep: result[i] = 1 converged, 0 iteration caps reached, < 0 the
 * member's error code.  Per member identical to calipso_hip_solve.  Members with a device evaluator are evaluated inside the
 * group's launches; the others through their host callback (calipso_hip_group_set_evaluators: one calipso_eval_fn / user pointer
 * per member, NULL where a device evaluator is attached), one member after the other. */
int32_t calipso_hip_group_set_evaluators(calipso_hip_group*, const calipso_eval_fn* evals, void* const* users);
int32_t calipso_hip_group_solve(calipso_hip_group*, int32_t* result);

/* ---- stage-banded structure -------------------------------------------------------------------------------
 * Trajectory-optimisation problems order their variables stage by stage (src/trajectory_optimization/indices.jl:41-180), which
 * makes the Schur complement onto x banded.  calipso_hip_analyze_structure reads the non-zero pattern of the blocks the handle
 * currently holds (lagrangian_hessian, equality / cone Jacobians) and from then on the factorisation and the triangular solves
 * skip everything outside the band (the reference obtains the same saving from its sparse LDL^T, qdldl.jl:400-589).  Results are
 * bit-identical to the dense treatment.  Re-run it (or calipso_hip_clear_structure) if the pattern of the blocks changes.
 * out[0] = half bandwidth of S, out[1] = 64-row blocks per panel inside the band (0: dense treatment kept),
 * out[2], out[3] = average number of equality / cone rows a 16-column group visits.  out may be NULL. */
int32_t calipso_hip_analyze_structure(calipso_hip_solver*, int64_t out[4]);
int32_t calipso_hip_clear_structure(calipso_hip_solver*);
/* Stage-parallel factorisation of the Schur complement (the reference gets its parallel-free equivalent from AMD + sparse QDLDL,
 * qdldl.jl:134-188,400-589): after calipso_hip_analyze_structure, S is factored by the multifrontal sparse LDL^T over a nested dissection of its
 * skyline pattern (for a trajectory problem: log2(stages) launches instead of the chain of nx pivots) and solved over the same tree.  `batch` >= the
 * largest group this handle will lead (1 for a handle stepped alone).  Same inertia as the blocked factorisation; values agree to rounding.
 * info (may be NULL) = [tree levels, rows of the largest front, nnz(triu) of the pattern of S, 2].  on = 0 switches back.  Fails (and keeps the
 * blocked factorisation) when a front would exceed one CU's LDS (196 rows). */
int32_t calipso_hip_set_stage_parallel(calipso_hip_solver*, int32_t on, int32_t batch, int64_t info[4]);
/* Stage blocks (after calipso_hip_analyze_structure; SURVEY.md 8(f1): the reference keeps stage-structured problems sparse end to end,
 * src/trajectory_optimization/sparsity.jl:28-129, src/solver/evaluate.jl:37-121): the runs of constraint rows with one column range and the diagonal
 * blocks of the Lagrangian Hessian are packed contiguously and the mat-vecs of the Newton step and the Schur-complement kernel work on the packed
 * blocks (one workgroup per block / per pair of column segments) instead of the dense-layout kernels.  The dense buffers stay the interchange format of
 * the uploads; a pack kernel follows every upload on the handle's stream.  Results agree with the dense treatment to rounding (not bitwise).  on = 0
 * switches back.  Refused when the structure has fewer than two Hessian blocks.  Members of a group must agree (all on with one structure, or all off).
 * info (may be NULL) = [Z blocks, Hessian blocks, column segments, packed doubles per instance]. */
int32_t calipso_hip_set_stage_blocks(calipso_hip_solver*, int32_t on, int64_t info[4]);

/* ---- LinearSolver seam (src/solver/linear_solver.jl:1-60) ----------------------------------------------------------------------
 * A stand-alone device LDL^T for ANY sparse symmetric quasi-definite matrix the caller assembled itself, so that the reference's own
 * search_direction! / iterative_refinement! / differentiate! / inertia_correction! (which hand `data.jacobian_variables_symmetric`
 * to `solver.linear_solver`) run unmodified on the GPU factorisation: `HIPLDLSolver <: LinearSolver` in julia/CalipsoHIP.jl.
 *   calipso_hip_ldl_create         = ldl_solver(A)                               linear_solver.jl:46-50  (destroy: calipso_hip_destroy)
 *   calipso_hip_ldl_factorize_csc  = factorize!(s, A; update) + compute_inertia! linear_solver.jl:19-44, qdldl.jl:134-188,269-317
 *   calipso_hip_ldl_inertia        = compute_inertia!(s) of the last factorisation
 *   calipso_hip_ldl_solve          = linear_solve!(s, x, A, b; fact=false)       linear_solver.jl:52-60,82-99, qdldl.jl:330-351
 * A is Julia's SparseMatrixCSC: colptr[n+1], rowval[nnz] 1-based Int64, nzval[nnz]; only triu(A) is read (linear_solver.jl:23);
 * no pivoting; inertia = (#D>0, #D<=0, #D==0), positive = -1 and CALIPSO_WARN_ZERO_PIVOT on an exact zero pivot (qdldl.jl:456,579).
 * b, x: column-major n x nrhs host arrays (may alias). */
int32_t calipso_hip_ldl_create(int64_t n, int32_t device, calipso_hip_solver** out);
int32_t calipso_hip_ldl_factorize_csc(calipso_hip_solver*, int64_t n, const int64_t* colptr, const int64_t* rowval, const double* nzval,
                                      int64_t inertia[3]);
int32_t calipso_hip_ldl_inertia(calipso_hip_solver*, int64_t inertia[3]);
/* Ordering / symbolic service (qdldl.jl:134-188,358-395,642-742).  Host-side integer work, 1-based Int64 like the reference.
 *   calipso_hip_ordering: elimination order of a sparse symmetric pattern (CSC, any triangle(s)): method 0 natural, 1 reverse Cuthill-McKee
 *     (minimum bandwidth: what the dense device factorisation exploits), 4 nested dissection (level-structure separators: minimum tree height, what
 *     the sparse device factorisation exploits), 2 minimum degree on the quotient graph (AMD's order class; AMD.jl's exact
 *     output, qdldl.jl:135, is third-party and unpinned).  perm[k] = vertex eliminated k-th.
 *   calipso_hip_symbolic: permute_symmetric + QDLDL_etree! for triu(A) under perm (NULL = natural): Pp[n+1], Pi[nnz triu], AtoPAPt[nnz A] (0 below
 *     the diagonal), etree[n] (-1 = root), Lnz[n]; returns nnz(L) (-1 in the reference's failure cases); info = [half bandwidth of PAP', nnz triu].
 *     Bit-exact against the oracle's restatement for the same perm.  Outputs may be NULL.
 *   calipso_hip_ldl_analyze_csc: installs the order on a calipso_hip_ldl_create handle (method 3 = the caller's perm); the following
 *     calipso_hip_ldl_factorize_csc calls factor P A P' and, if it is banded, only inside the band; calipso_hip_ldl_solve permutes the right-hand
 *     sides (qdldl.jl:330-351).  info = [half bandwidth, band blocks used (0 = dense treatment), nnz(L) symbolic, nnz triu]. */
int32_t calipso_hip_ordering(int64_t n, const int64_t* colptr, const int64_t* rowval, int32_t method, int64_t* perm);
int64_t calipso_hip_symbolic(int64_t n, const int64_t* colptr, const int64_t* rowval, const int64_t* perm, int64_t* Pp, int64_t* Pi, int64_t* AtoPAPt,
                             int64_t* etree, int64_t* Lnz, int64_t info[2]);
int32_t calipso_hip_ldl_analyze_csc(calipso_hip_solver*, int64_t n, const int64_t* colptr, const int64_t* rowval, int32_t method, int64_t* perm,
                                    int64_t info[4]);
int32_t calipso_hip_ldl_solve(calipso_hip_solver*, int64_t n, int64_t nrhs, const double* b, double* x);

/* ---- sparse LDL^T on the device (src/solver/qdldl.jl:134-188 analyse, :400-589 factor, :330-351,592-640 solve) --------
 * The LinearSolver seam WITHOUT dense n x n storage: memory O(nnz(L)).  The analyse phase (host) orders the matrix (method 0 natural, 1 RCM,
 * 2 minimum degree, 4 nested dissection, 3 = the caller's 1-based `perm`), builds P A P', the elimination tree, the pattern of L and the LEVEL
 * schedule; the numeric phase is a left-looking factorisation by tree levels (one launch per level, one workgroup per column, single-column levels
 * merged into chains) and level-scheduled triangular solves.  With a nested-dissection order the stages of a trajectory problem
 * (trajectory_optimization/sparsity.jl:28-129) are eliminated in parallel: the number of levels is the sequential depth.  No pivoting; inertia and
 * zero-pivot behaviour as calipso_hip_ldl_factorize_csc.  L and D agree with QDLDL's to rounding (the summation order differs).
 *   calipso_hip_sparse_create      = QDLDL analyse of qdldl(A; perm)          qdldl.jl:134-188, 358-395, 642-742  (pattern only: colptr, rowval 1-based)
 *   calipso_hip_sparse_factorize   = QDLDL_factor! + compute_inertia!         qdldl.jl:400-589, linear_solver.jl:19-44  (nzval in the pattern's order, host)
 *   calipso_hip_sparse_solve       = solve!(F, b) for nrhs columns            qdldl.jl:330-351
 *   calipso_hip_sparse_get_factor  = F.perm, F.L (strictly lower, 1-based CSC), F.D   qdldl.jl:160-166
 *   calipso_hip_sparse_info        info[8] = n, nnz(triu A), nnz(L), tree levels, launches per factorisation, widest level, multiply-adds,
 *                                  numeric phase (0 columns / global accumulator, 1 columns / LDS accumulator, 2 multifrontal)
 *   calipso_hip_sparse_timing      ms[2] = device time of the last factorisation / solve (HIP events on the handle's stream)
 *   calipso_hip_sparse_set_batch   `batch` matrices of the analysed pattern per call (BASELINE config C4: many independent problems of one structure):
 *                                  nzval = batch x nnz, inertia = batch x 3, b / x = batch x (n x nrhs); the multifrontal path factors them in the
 *                                  same launches.  calipso_hip_sparse_select picks the matrix calipso_hip_sparse_get_factor reads.
 * Numeric phase: with method 4 (nested dissection) the factorisation is MULTIFRONTAL over the dissection tree — the pieces of the dissection (cut into
 * chains of <= 64 columns) are the supernodes, each front is assembled and partially factored by one workgroup, in its LDS when the front has <= 196
 * rows, in global memory up to 1024 rows; one launch per tree level (~log2 T launches for a T-stage problem).  Larger fronts, every other order
 * and method 5 (nested-dissection order, column method) take the left-looking column method. */
typedef struct calipso_hip_sparse calipso_hip_sparse;
int32_t calipso_hip_sparse_create(int64_t n, const int64_t* colptr, const int64_t* rowval, int32_t method, const int64_t* perm, int32_t device,
                                  calipso_hip_sparse** out);
int32_t calipso_hip_sparse_destroy(calipso_hip_sparse*);
const char* calipso_hip_sparse_last_error(calipso_hip_sparse*);
int32_t calipso_hip_sparse_info(calipso_hip_sparse*, int64_t info[8]);
int32_t calipso_hip_sparse_set_batch(calipso_hip_sparse*, int64_t batch);
int32_t calipso_hip_sparse_select(calipso_hip_sparse*, int64_t instance);
int32_t calipso_hip_sparse_factorize(calipso_hip_sparse*, const double* nzval, int64_t* inertia);
int32_t calipso_hip_sparse_solve(calipso_hip_sparse*, int64_t nrhs, const double* b, double* x);
/* the same with values / right-hand sides / solutions already resident on the handle's device (device pointers) */
int32_t calipso_hip_sparse_factorize_device(calipso_hip_sparse*, const double* d_nzval, int64_t* inertia);
int32_t calipso_hip_sparse_solve_device(calipso_hip_sparse*, int64_t nrhs, const double* d_b, double* d_x);
int32_t calipso_hip_sparse_get_factor(calipso_hip_sparse*, int64_t* perm, int64_t* Lp, int64_t* Li, double* Lx, double* D);
int32_t calipso_hip_sparse_timing(calipso_hip_sparse*, double ms[2]);

/* ---- batched small systems (BASELINE config C5): LDS-resident LDL^T + multi-right-hand-side solve, one workgroup per
 * instance, one launch for the whole batch.  The sensitivity solves of the reference's MPC auto-tuning loop
 * (examples/autotuning/cartpole.jl:179-227 -> differentiate.jl:29-58: n = 89, 102 parameter columns per step) for thousands of
 * independent steps at once; semantics of the LinearSolver seam above (only triu(K) read, no pivoting, natural order).
 *   create(n <= 128, nrhs, batch, device)   set(K[batch][n*n], B[batch][n*nrhs]) column-major host arrays (NULL = keep what is resident)
 *   solve(&ms)   one launch, inputs resident; ms = its HIP-event duration   get(X[batch][n*nrhs], inertia[batch][3]) -> number of
 *   instances with an exact zero pivot (inertia as compute_inertia!, linear_solver.jl:33-44) or a negative status */
typedef struct calipso_hip_small calipso_hip_small;
int32_t calipso_hip_small_create(int64_t n, int64_t nrhs, int64_t batch, int32_t device, calipso_hip_small** out);
int32_t calipso_hip_small_destroy(calipso_hip_small*);
const char* calipso_hip_small_last_error(calipso_hip_small*);
int32_t calipso_hip_small_set(calipso_hip_small*, const double* K, const double* B);
int32_t calipso_hip_small_solve(calipso_hip_small*, double* ms);
int32_t calipso_hip_small_get(calipso_hip_small*, double* X, int64_t* inertia);

/* ---- solve! for a batch of SMALL conic QPs, the whole Newton iteration in one kernel (csrc/smallnewton.hip) ----------------------------------
 * Solver / initialize! / solve! (src/solver/solver.jl:46-150, initialize.jl:9-48, solve.jl:8-377) for `batch` independent problems of ONE shape that are too small
 * for the general path to be anything but launch latency (the MPC problems of examples/autotuning/cartpole.jl:179-227: n = 89): one workgroup per instance, problem
 * data ([A; -G], q, [-b; h]; the Hessian block stays in L2), iterates and the factor in the compute unit's LDS, every decision of solve.jl:98-368 (exit tests, inertia_correction!, iterative_refinement!, cone search,
 * filter line search, outer updates) on the device, ONE launch per call.  Evaluator: the QP of calipso_hip_qp_attach (min c x'Px + q'x s.t. Ax = b, h - Gx >= 0) with
 * nonnegative and second-order cones (dimension <= 16; wider: the general path); residual_norm = constraint_norm = 1.  Points have the layout of point.jl:13-22 (N = nx + 2 ne + 3 nc).  Limits: nx <= 128 and the
 * instance must fit 160 KB of LDS (n up to ~200), else CALIPSO_ERR_ARGUMENT at create: the general path (calipso_hip_create + groups) takes those.
 *   create(nx, ne, nc, batch, device)        set_option(name, value): options.jl:6-59 by name; plus "threads" = threads per instance (0: chosen by the LDS footprint so that
 *                                            a compute unit holds as many instances as fit; 64, 128 or 256 force a build of the kernel)
 *   set_qp(P, q, A, b, G, h, c, shared)      column-major host arrays, batch-major (instance k at k * size) or ONE problem for all (shared != 0)
 *   set_state(w, lambda, scalars)            points (batch x N; initialize!: x in the first nx entries), lambda (batch x ne), [central_path, fraction_to_boundary, penalty] (batch x 3)
 *   solve(result, ms)                        solve! of every instance: 1 converged, 0 iteration caps, CALIPSO_ERR_INERTIA / CALIPSO_ERR_CONE_SEARCH (the reference's error()s),
 *                                            -100 - CALIPSO_WARN_REFINEMENT where the reference would fall back to `H \ residual` (search_direction.jl:22: left to the general path)
 *   steps(count, advance, info, status, ms)  `count` passes of the inner loop body (solve.jl:98-353) from the resident state = calipso_hip_newton_steps for the batch
 *   get_state(w, lambda, scalars, counters)  scalars batch x 6 [central_path, fraction_to_boundary, penalty, primal_regularization, primal_regularization_last, dual_regularization],
 *                                            counters batch x 8 [total_iterations, outer, factorizations, refinement failures, max / last refinement rounds, Newton steps, accepted iterates]
 *   trace(rows, NULL) keeps the first `rows` accepted iterates of every instance (solution.all after solve.jl:309-326); trace(rows, out) reads them (batch x rows x N) */
typedef struct calipso_hip_smallnewton calipso_hip_smallnewton;
int32_t calipso_hip_smallnewton_create(int64_t nx, int64_t ne, int64_t nc, int64_t batch, int32_t device, calipso_hip_smallnewton** out);
int32_t calipso_hip_smallnewton_destroy(calipso_hip_smallnewton*);
const char* calipso_hip_smallnewton_last_error(calipso_hip_smallnewton*);
int32_t calipso_hip_smallnewton_set_option(calipso_hip_smallnewton*, const char* name, double value);
/* cone layout (indices.jl:45-63): the first n_nonnegative cone entries nonnegative (default: all nc), then n_soc second-order cones of dims[j] (2 .. 16) entries each, contiguous */
int32_t calipso_hip_smallnewton_set_cones(calipso_hip_smallnewton*, int64_t n_nonnegative, int64_t n_soc, const int64_t* dims);
int32_t calipso_hip_smallnewton_set_qp(calipso_hip_smallnewton*, const double* P, const double* q, const double* A, const double* b, const double* G, const double* h,
                                       double objective_scale, int32_t shared);
int32_t calipso_hip_smallnewton_set_state(calipso_hip_smallnewton*, const double* w, const double* lambda, const double* scalars);
int32_t calipso_hip_smallnewton_get_state(calipso_hip_smallnewton*, double* w, double* lambda, double* scalars, int64_t* counters);
int32_t calipso_hip_smallnewton_trace(calipso_hip_smallnewton*, int32_t rows, double* out);
int32_t calipso_hip_smallnewton_solve(calipso_hip_smallnewton*, int32_t* result, double* ms);
int32_t calipso_hip_smallnewton_steps(calipso_hip_smallnewton*, int32_t count, int32_t advance, double* info, int32_t* status, double* ms);
/* differentiate!(solver)  differentiate.jl:1-61 for every instance in ONE launch, at the resident points (after calipso_hip_smallnewton_solve): the condensed matrix
 * with the regularisation solve! left is factored once (:13-20), then search_direction_symmetric! per column of dR/dtheta (:29-52) and sensitivity = -1.0 * the
 * result (:55-57).  jacobian_parameters (residual_jacobian_parameters.jl:1-40: the caller's model; constant for the parametric QPs of an MPC loop,
 * examples/autotuning/cartpole.jl:179-227): batch x (N x p) doubles on the host, column-major per instance — or ONE N x p matrix for all instances (shared != 0) —,
 * N = nx + 2 ne + 3 nc in the order of point.jl; sensitivity: batch x (N x p).  Batches without second-order cones refine every column (iterative_refinement.jl:1-52:
 * the condensed solve is five digits short of the reference's QDLDL at a solution); with second-order cones the unrefined solve IS the reference's result (quirk B-3).
 * The cone Jacobians are those of the last search direction, as the reference's fields are (quirk B-12).  status[k] = 0, or 1 when the inertia of the factorisation is not (nx, ne + nc, 0) (the reference does not look). */
int32_t calipso_hip_smallnewton_differentiate(calipso_hip_smallnewton*, int64_t p, int32_t shared, const double* jacobian_parameters, double* sensitivity, int32_t* status, double* ms);

/* ---- multi-GPU exchange of the batched path: RCCL over xGMI, one process per GPU ---------------------------------
 * Problem instances are sharded block-contiguously over ranks and never interact (the reference's `Solver`s are independent); the
 * only exchange is after a batch round: all-gather of per-problem status rows, all-reduce of counters.  RCCL is dlopen'ed on first
 * use.  calipso_hip_comm_unique_id = ncclGetUniqueId (one rank), calipso_hip_comm_init = ncclCommInitRank (all ranks, same id). */
typedef struct calipso_hip_comm calipso_hip_comm;
int32_t calipso_hip_comm_unique_id(uint8_t id[128]);
int32_t calipso_hip_comm_init(int32_t rank, int32_t nranks, const uint8_t id[128], int32_t device, calipso_hip_comm** out);
int32_t calipso_hip_comm_destroy(calipso_hip_comm*);
/* out[0] = ncclCommCount, out[1] = ncclCommUserRank of the live communicator (what RCCL itself reports: the self-check of a multi-GPU run) */
int32_t calipso_hip_comm_size(calipso_hip_comm*, int32_t out[2]);
const char* calipso_hip_comm_last_error(calipso_hip_comm*);
/* rows: n_rows x 4 int32 of this rank (row counts may differ between ranks); all_rows: capacity cap_rows rows, filled in global
 * problem-id order; counts_out[nranks] (may be NULL) = rows per rank.  Returns the total row count or a negative status. */
int64_t calipso_hip_comm_gather_status(calipso_hip_comm*, const int32_t* rows, int64_t n_rows, int32_t* all_rows, int64_t cap_rows,
                                       int64_t* counts_out);
int32_t calipso_hip_comm_allreduce_sum(calipso_hip_comm*, double* values, int64_t count);

/* measured fp64 matrix-core ceiling (TFLOP/s) of `device`: back-to-back independent v_mfma_f64_16x16x4_f64, 8 wavefronts per SIMD
 * (a few milliseconds).  bench.py reports it as roofline.peak_measured next to the datasheet peak. */
int32_t calipso_hip_mfma_f64_peak(int32_t device, double* tflops);

/* timing of the last calipso_hip_newton_step / factorisation, in milliseconds, from HIP events on the handle's stream:
 * [0] evaluate + cone + residual + reductions   [1] cone pivots + Omega*hx   [2] search_direction! total (factor + solves + refinement)
 * [3] LDL^T of the Schur complement   [5] cone search + line search + accept   [6] whole step
 * [7] the Schur-complement MFMA kernel (k_schur), last launch   [8] number of factorisations timed so far */
int32_t calipso_hip_phase_times(calipso_hip_solver*, double out[9]);
/* per-kernel figures of the last factorisation and the handle's layout (bench.py: the live launch durations behind `roofline`):
 * [0] ms of the panel-step launches of the LDL^T of the Schur complement (k_ldl_diag + k_ldl_step: the pivot chain, one launch per 64
 *     pivots; HIP events around exactly these launches; the rest of [3] above is the parallel finish: factor columns + block inverses)
 * [1] number of those launches   [2] NP = padded order of the Schur complement   [3] bytes of the handle's device slab + the dense scratch a structured handle
 *     allocates for a calipso_device_eval_fn (none with a calipso_device_block_eval_fn)
 * [4] ms of ONE launch of the refinement residual's mat-vec kernel (k_gemv_t2_and_n: [gx; hx]' times two vectors and Lxx times one, the first residual of the
 *     last calipso_hip_newton_step; 0 when the handle takes another path) and [5] the bytes it reads, 8 (m nx + nx^2)
 * [6] 1 when the last factorisation took the left-looking schedule of one dense system (csrc/lfac.hip: the products of the Schur complement as slices of the panel
 *     launches — [0] / [1] then cover k_lfac's launches, Schur complement included, and phase [7] of calipso_hip_phase_times is ~0)   [7] bytes of that schedule's
 *     buffers outside the slab (Z = A M of every panel, the factor columns, the launch plan) */
int32_t calipso_hip_kernel_times(calipso_hip_solver*, double out[8]);
/* work of one Newton step on a handle that exploits its stage structure (src/trajectory_optimization/sparsity.jl:28-129 is where the reference's structure
 * comes from): [0] flops of the Schur complement by segment pairs (k_schur_blocks)   [1] doubles of the packed blocks of [gx; hx] and Lxx (both orientations)
 * [2] segment pairs   [3] flops of one multifrontal LDL^T of S   [4] nnz(L) of its fronts   [5] order of S   [6] 1 for a structured handle   [7] reserved */
int32_t calipso_hip_structure_work(calipso_hip_solver*, double out[8]);
int32_t calipso_hip_synchronize(calipso_hip_solver*);
/* Independent Solvers stepped from different host threads (solver.jl:46-150: the reference's Solvers are independent objects; batch.py: lanes) overlap on the GPU only
 * if the runtime put their streams behind different hardware dispatchers — which depends on the order in which the process created its streams and cannot be queried.
 * calipso_hip_streams_concurrent MEASURES it (a kernel of many more workgroups than the chip holds on one stream, a one-workgroup kernel on the other, both ways):
 * out[1], out[2] = the short kernel's time behind a's / b's long kernel (us), out[3] = the long kernel's duration (us) — and chains of 48 short kernels on both streams at
 * once against one chain alone (out[4], out[5], us: two streams of the highest priority class pass the first test and still run their chains one after the other);
 * out[0] = 1 if the streams run side by side by both tests, else 0.  calipso_hip_rebind_stream gives the handle a NEW stream (the old one is drained and destroyed; priority_class
 * 0, 1, 2 = the classes calipso_hip_create deals out by creation order, -1 = keep): the runtime binds it to the least used hardware queue of the class.  The Python
 * mirror's BatchSolver (and the Julia module's lanes) probe the leaders of their lanes at creation and rebind until every pair runs side by side. */
int32_t calipso_hip_streams_concurrent(calipso_hip_solver* a, calipso_hip_solver* b, double out[6]);
int32_t calipso_hip_rebind_stream(calipso_hip_solver*, int32_t priority_class);

/* SplitMix64 uniform stream of SURVEY.md 8(d): seed = 0xCA11B50000000000 + 4096*problem_id + stream_id,
 * u = (next() >> 11) * 2^-53, out[i] = lo + (hi-lo)*u.  Pure host function. */
int32_t calipso_hip_splitmix_uniform(uint64_t problem_id, uint64_t stream_id, double lo, double hi, int64_t count, double* out);

#ifdef __cplusplus
}
#endif
#endif
