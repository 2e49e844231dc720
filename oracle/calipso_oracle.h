/*
 * calipso_oracle.h — C interface of the CPU ORACLE.
 *
 * TEST INFRASTRUCTURE ONLY.  This library restates, on the CPU and single-threaded, the
 * algorithm of the reference's Newton/KKT hot path (thowell/CALIPSO.jl v0.1.1, src/solver/ *.jl files).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product
 * (libcalipso_hip.so) never links, loads or calls anything in oracle/.
 *
 * PARITY STATUS: the reference is pure Julia and cannot be run in the build container (no julia),
 * and it ships no golden-vector files (all of its test inputs are unseeded rand()).  The oracle is
 * pinned against the closed-form identities and known answers of the reference's own tests
 * (test/solver/problem.jl:112-211, wachter.jl:47, friction_cone.jl:56-63, portfolio.jl:59-62,
 * qp_equality.jl:106-122; see tests/test_oracle_*.py).  The elimination order of the LDL^T is
 * third-party (AMD.jl, Project.toml:7,19; call site src/solver/qdldl.jl:135) and is not pinned by
 * any reference test => "parity unpinned" at that boundary; the oracle takes the permutation as an
 * input and defaults to the constraint-first order [z | y | x].
 *
 * All indices crossing this interface are 1-based Int64, exactly as in the Julia reference, so the
 * index work (Indices, perm/iperm, AtoPAPt, etree, Lp/Li) can be compared bit-for-bit.
 */
#ifndef CALIPSO_ORACLE_H
#define CALIPSO_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_solver oracle_solver;

/* evaluate! flags (src/solver/evaluate.jl:1-23 keyword arguments, one bit each) */
enum {
    ORC_OBJECTIVE                     = 1u << 0,   /* f            -> objective[1]                                   */
    ORC_OBJECTIVE_GRADIENT            = 1u << 1,   /* fx           -> objective_gradient_variables[nx]               */
    ORC_OBJECTIVE_HESSIAN             = 1u << 2,   /* fxx          -> objective_jacobian_variables_variables[nx,nx]  */
    ORC_EQUALITY                      = 1u << 3,   /* g            -> equality_constraint[ne]                        */
    ORC_EQUALITY_JACOBIAN             = 1u << 4,   /* gx           -> equality_jacobian_variables[ne,nx]             */
    ORC_EQUALITY_DUAL_GRADIENT        = 1u << 5,   /* (g'y)x       -> equality_dual_jacobian_variables[nx]           */
    ORC_EQUALITY_DUAL_HESSIAN         = 1u << 6,   /* (g'y)xx      -> equality_dual_jacobian_variables_variables     */
    ORC_CONE                          = 1u << 7,   /* h            -> cone_constraint[nc]                            */
    ORC_CONE_JACOBIAN                 = 1u << 8,   /* hx           -> cone_jacobian_variables[nc,nx]                 */
    ORC_CONE_DUAL_GRADIENT            = 1u << 9,   /* (h'z)x       -> cone_dual_jacobian_variables[nx]               */
    ORC_CONE_DUAL_HESSIAN             = 1u << 10,  /* (h'z)xx      -> cone_dual_jacobian_variables_variables         */
    ORC_OBJECTIVE_JACOBIAN_PARAMETERS = 1u << 11,  /* fx\theta     -> objective_jacobian_variables_parameters[nx,np] */
    ORC_EQUALITY_JACOBIAN_PARAMETERS  = 1u << 12,  /* g\theta      -> equality_jacobian_parameters[ne,np]            */
    ORC_EQUALITY_DUAL_JACOBIAN_PARAMETERS = 1u << 13, /* (g'y)x\theta                                               */
    ORC_CONE_JACOBIAN_PARAMETERS      = 1u << 14,  /* h\theta      -> cone_jacobian_parameters[nc,np]                */
    ORC_CONE_DUAL_JACOBIAN_PARAMETERS = 1u << 15   /* (h'z)x\theta                                                   */
};

/* The user-evaluation callback (stands in for the Symbolics-generated functions that
 * src/solver/evaluate.jl:1-124 calls).  It must write the requested quantities at (x, y, z, theta)
 * into the solver's named buffers (oracle_buffer), column-major.  Return 0 on success. */
typedef int (*oracle_eval_fn)(void* user, uint32_t flags,
                              const double* x, const double* y, const double* z, const double* theta);

/* construction: Solver(methods, nx, np, ne, nc; nonnegative_indices, second_order_indices) (solver.jl:46-150).
 * nonneg_idx: 1-based cone-local indices; soc_ptr (n_soc+1, 0-based offsets into soc_idx); soc_idx 1-based. */
oracle_solver* oracle_create(int64_t nx, int64_t np, int64_t ne, int64_t nc,
                             int64_t n_nonneg, const int64_t* nonneg_idx,
                             int64_t n_soc, const int64_t* soc_ptr, const int64_t* soc_idx);
void oracle_destroy(oracle_solver*);

/* named double buffers (ProblemData / SolverData / Points / scalars); returns pointer and length.
 * See oracle/README.md for the list of names (they are the reference's field names). */
double* oracle_buffer(oracle_solver*, const char* name, int64_t* len);
/* named Int64 index vectors of `Indices` (indices.jl:1-63), 1-based values. */
const int64_t* oracle_index(oracle_solver*, const char* name, int64_t* len);
/* integer option / state access: "max_outer_iterations", ... (options.jl:6-59); doubles go through oracle_buffer("opt.<name>") */
int64_t* oracle_int(oracle_solver*, const char* name);

/* --- hot path, one function per reference function ------------------------------------------ */
/* cone!(problem, methods, idx, point; ...) cones/cone.jl:71-106; which: 0 = solution, 1 = candidate */
void oracle_cone(oracle_solver*, int which, int barrier, int barrier_gradient, int product, int jacobian, int target);
/* cone_violation(xhat, x, tau, idx_ineq, idx_soc) cones/cone.jl:62-68 on raw vectors of length nc */
int  oracle_cone_violation(oracle_solver*, const double* xhat, const double* x, double tau);
void oracle_residual(oracle_solver*);                                  /* residual.jl:1-51 */
void oracle_residual_jacobian_variables(oracle_solver*);               /* residual_jacobian_variables.jl:1-108 (block form of H) */
void oracle_residual_jacobian_variables_symmetric(oracle_solver*);     /* :110-167, dense K (both triangles) */
void oracle_H_dense(oracle_solver*, double* out /* N*N col-major */);  /* materialise H for block checks */
void oracle_H_mul(oracle_solver*, const double* v, double* out);       /* out = H*v (structured) */
void oracle_residual_symmetric(oracle_solver*, int which);             /* residual.jl:53-101; which 0: residual, 1: residual_error, 2: jacobian_parameters_vector */
/* factorize!(linear_solver, K; update) linear_solver.jl:19-31.  update=0 redoes symbolic analysis. returns posDCount (-1 on zero pivot) */
int64_t oracle_factorize(oracle_solver*, int update);
void oracle_compute_inertia(oracle_solver*, int64_t out[3]);           /* linear_solver.jl:33-44 -> positive, negative, zero */
void oracle_linear_solve(oracle_solver*, double* x, const double* b, int fact, int update); /* linear_solver.jl:52-60 */
/* search_direction_symmetric! search_direction.jl:25-104; which 0: (step,residual) 1: (step_correction,residual_error) */
void oracle_search_direction_symmetric(oracle_solver*, int which, int fact);
int  oracle_iterative_refinement(oracle_solver*);                      /* iterative_refinement.jl:1-52; 1 = true */
int  oracle_inertia_correction(oracle_solver*);                        /* inertia.jl:30-80; 0 ok, -1 "inertia correction failure" */
int  oracle_search_direction(oracle_solver*);                          /* search_direction.jl:1-23; >0: refinement failed, dense LU fallback used */
double oracle_merit(oracle_solver*, double f, const double* r, double Phi);   /* merit.jl:2-15 */
void   oracle_merit_gradient(oracle_solver*);                                 /* merit.jl:17-31 */
double oracle_constraint_violation(oracle_solver*, const double* g, const double* r, const double* h, const double* s); /* constraint_violation.jl:1-13 */
double oracle_optimality_error(oracle_solver*);                               /* optimality_error.jl:1-27 */
/* filter.jl */
void oracle_filter_reset(oracle_solver*);
int  oracle_check_filter(oracle_solver*, double theta, double merit);
void oracle_augment_filter(oracle_solver*, double theta, double merit);
int64_t oracle_filter_pairs(oracle_solver*, double* out /* 2*max_filter */);
/* line_search.jl */
int oracle_switching_condition(double step_size, const double* dir, const double* grad, int64_t n, double merit_exponent, double violation, double violation_exponent, double regularization);
int oracle_sufficient_progress(double violation, double violation_candidate, double merit, double merit_candidate, double violation_tolerance, double merit_tolerance, double machine_tolerance);
int oracle_armijo(double merit, double merit_candidate, const double* grad, const double* dir, int64_t n, double step_size, double armijo_tolerance, double machine_tolerance);

/* solve!(solver) solve.jl:8-377.  returns 1 (true) / 0 (false); negative = reference error():
 * -1 "inertia correction failure" (inertia.jl:72), -2 "cone search failure" (solve.jl:210,220), -3 callback error. */
int oracle_solve(oracle_solver*, oracle_eval_fn eval, void* user);
/* differentiate!(solver) differentiate.jl:1-61 */
int oracle_differentiate(oracle_solver*, oracle_eval_fn eval, void* user);
/* iterate trace of the last oracle_solve: row k = solution.all after the k-th accepted inner iteration (solve.jl:309-326).
 * Returns the number of rows available; copies min(rows, cap_rows) rows of length N. */
int64_t oracle_trace(oracle_solver*, double* out, int64_t cap_rows);
/* per-solve statistics: [total_iterations, outer, factorizations, refinement_failures, max_refinement_rounds] */
void oracle_stats(oracle_solver*, int64_t out[8]);
/* elimination order used by factorize! (1-based perm of 1:n).  Default: [z | y | x]. */
void oracle_set_perm(oracle_solver*, const int64_t* perm);

/* --- stand-alone restatement of the vendored QDLDL (qdldl.jl:358-742), 1-based CSC ------------- */
/* permute_symmetric (qdldl.jl:642-742): A upper-triangular CSC (n, Ap[n+1], Ai, Ax), iperm -> P (Pp, Pi, Px), AtoPAPt */
void oracle_qdldl_permute_symmetric(int64_t n, const int64_t* Ap, const int64_t* Ai, const double* Ax,
                                    const int64_t* iperm, int64_t* Pp, int64_t* Pi, double* Px, int64_t* AtoPAPt);
/* QDLDL_etree! (qdldl.jl:358-395): returns sum(Lnz) or -1 */
int64_t oracle_qdldl_etree(int64_t n, const int64_t* Ap, const int64_t* Ai, int64_t* work, int64_t* Lnz, int64_t* etree);
/* QDLDL_factor! (qdldl.jl:400-589) without dynamic regularisation (Dsigns == nothing, as CALIPSO calls it) */
int64_t oracle_qdldl_factor(int64_t n, const int64_t* Ap, const int64_t* Ai, const double* Ax,
                            int64_t* Lp, int64_t* Li, double* Lx, double* D, double* Dinv,
                            const int64_t* Lnz, const int64_t* etree);
/* QDLDL_solve! (qdldl.jl:616-622) in place */
void oracle_qdldl_solve(int64_t n, const int64_t* Lp, const int64_t* Li, const double* Lx, const double* Dinv, double* b);

/* --- synthetic problem generator shared bit-for-bit with the HIP side ---------- */
/* SplitMix64 stream: seed = 0xCA11B50000000000 + 4096*problem_id + stream_id; u = (next>>11)*2^-53 */
void oracle_splitmix_uniform(uint64_t problem_id, uint64_t stream_id, double lo, double hi, int64_t count, double* out);

#ifdef __cplusplus
}
#endif
#endif
