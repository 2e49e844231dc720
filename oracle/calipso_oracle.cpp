// calipso_oracle.cpp — CPU ORACLE (test infrastructure, NOT product code; see calipso_oracle.h).
//
// A single-threaded restatement of the reference algorithm, function by function.  Every function
// cites the reference file:line it follows (paths relative to /root/reference/src/solver/).
// Storage is dense column-major (the reference's ProblemData blocks are dense Matrix{T},
// problem_data.jl:2-31); the unreduced Jacobian H is kept by its non-zero blocks (the reference
// holds the same entries in a SparseMatrixCSC).  Indices are 0-based internally, 1-based at the API.
//
// Nothing here was copied from the reference: it is Julia, this is C++; the arithmetic order of
// the factorisation (qdldl.jl:400-589) is followed deliberately because rounding depends on it.

#include "calipso_oracle.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

typedef int64_t i64;
typedef std::vector<double> vec;
typedef std::vector<i64> ivec;

namespace {

// ------------------------------------------------------------------------------------------
// options.jl:6-59
struct Options {
    double residual_norm = 1.0, constraint_norm = 1.0;
    i64 max_outer_iterations = 10, max_residual_iterations = 100;
    double scaling_line_search = 0.5;
    i64 max_residual_line_search = 25, max_cone_line_search = 25;
    i64 iterative_refinement = 1, max_iterative_refinement = 10, min_iterative_refinement = 1;
    double iterative_refinement_tolerance = 1.0e-10;
    double central_path_initial = 1.0, central_path_update_tolerance = 10.0, central_path_scaling = 0.2,
           central_path_exponent = 1.5;
    double penalty_initial = 1.0, penalty_scaling = 10.0, dual_initial = 0.0;
    double residual_tolerance = 1.0e-4, optimality_tolerance = 1.0e-4, slack_tolerance = 1.0e-4,
           equality_tolerance = 1.0e-4, complementarity_tolerance = 1.0e-4;
    double min_regularization = 1.0e-20, primal_regularization_initial = 1.0e-7,
           dual_regularization_initial = 1.0e-7, max_regularization = 1.0e40, dual_regularization = 1.0e-8,
           dual_regularization_exponent = 0.25, scaling_regularization_initial = 100.0,
           scaling_regularization = 8.0, scaling_regularization_last = 1.0 / 3.0;
    double min_central_path = 1.0e-8, max_penalty = 1.0e8;
    i64 constraint_tensor = 1;
    i64 update_factorization = 1;
    double violation_tolerance = 1.0e-5, violation_exponent = 1.1, merit_tolerance = 1.0e-5,
           merit_exponent = 2.3, armijo_tolerance = 1.0e-4, machine_tolerance = 1.0e-16;
    i64 max_filter = 1000;
    i64 differentiate = 1;
    i64 warmstart = 0;
    i64 linear_solve_refactor = 1;  // linear_solver.jl:53 `fact=true` default (quirk B-4); 0 = factor once per matrix
};

// ------------------------------------------------------------------------------------------
// stand-alone QDLDL restatement (0-based internally)
// qdldl.jl:358-395
i64 qdldl_etree(i64 n, const i64* Ap, const i64* Ai, i64* work, i64* Lnz, i64* etree) {
    for (i64 i = 0; i < n; ++i) {
        work[i] = 0; Lnz[i] = 0; etree[i] = -1;
        if (Ap[i] == Ap[i + 1]) return -1;
    }
    for (i64 j = 0; j < n; ++j) {
        work[j] = j;
        for (i64 p = Ap[j]; p < Ap[j + 1]; ++p) {
            i64 i = Ai[p];
            if (i > j) return -1;
            while (work[i] != j) {
                if (etree[i] == -1) etree[i] = j;
                Lnz[i] += 1;
                work[i] = j;
                i = etree[i];
            }
        }
    }
    i64 s = 0;
    for (i64 i = 0; i < n; ++i) s += Lnz[i];
    return s;
}

// qdldl.jl:400-589 (Dsigns == nothing: CALIPSO never enables dynamic regularisation, linear_solver.jl:27,47)
i64 qdldl_factor(i64 n, const i64* Ap, const i64* Ai, const double* Ax, i64* Lp, i64* Li, double* Lx,
                 double* D, double* Dinv, const i64* Lnz, const i64* etree,
                 std::vector<char>& yMarkers, ivec& iwork, vec& yVals) {
    i64 positiveValuesInD = 0;
    yMarkers.assign(n, 0); iwork.assign(3 * n, 0); yVals.assign(n, 0.0);
    i64* yIdx = iwork.data();
    i64* elimBuffer = iwork.data() + n;
    i64* LNextSpaceInCol = iwork.data() + 2 * n;
    Lp[0] = 0;
    for (i64 i = 0; i < n; ++i) {
        Lp[i + 1] = Lp[i] + Lnz[i];
        yMarkers[i] = 0; yVals[i] = 0.0; D[i] = 0.0;
        LNextSpaceInCol[i] = Lp[i];
    }
    D[0] = Ax[0];
    if (D[0] == 0.0) return -1;
    if (D[0] > 0.0) positiveValuesInD += 1;
    Dinv[0] = 1 / D[0];
    for (i64 k = 1; k < n; ++k) {
        i64 nnzY = 0;
        for (i64 i = Ap[k]; i < Ap[k + 1]; ++i) {
            i64 bidx = Ai[i];
            if (bidx == k) { D[k] = Ax[i]; continue; }
            yVals[bidx] = Ax[i];
            i64 nextIdx = bidx;
            if (!yMarkers[nextIdx]) {
                yMarkers[nextIdx] = 1;
                elimBuffer[0] = nextIdx;
                i64 nnzE = 1;
                nextIdx = etree[bidx];
                while (nextIdx != -1 && nextIdx < k) {
                    if (yMarkers[nextIdx]) break;
                    yMarkers[nextIdx] = 1;
                    elimBuffer[nnzE] = nextIdx;
                    nnzE += 1;
                    nextIdx = etree[nextIdx];
                }
                while (nnzE != 0) { yIdx[nnzY++] = elimBuffer[--nnzE]; }
            }
        }
        for (i64 i = nnzY - 1; i >= 0; --i) {
            i64 cidx = yIdx[i];
            i64 tmpIdx = LNextSpaceInCol[cidx];
            double yVals_cidx = yVals[cidx];
            for (i64 j = Lp[cidx]; j < tmpIdx; ++j) yVals[Li[j]] -= Lx[j] * yVals_cidx;
            Lx[tmpIdx] = yVals_cidx * Dinv[cidx];
            D[k] -= yVals_cidx * Lx[tmpIdx];
            Li[tmpIdx] = k;
            LNextSpaceInCol[cidx] += 1;
            yVals[cidx] = 0.0;
            yMarkers[cidx] = 0;
        }
        if (D[k] == 0.0) return -1;
        if (D[k] > 0.0) positiveValuesInD += 1;
        Dinv[k] = 1 / D[k];
    }
    return positiveValuesInD;
}

// qdldl.jl:592-622
void qdldl_solve(i64 n, const i64* Lp, const i64* Li, const double* Lx, const double* Dinv, double* x) {
    for (i64 i = 0; i < n; ++i) {
        double xi = x[i];
        for (i64 j = Lp[i]; j < Lp[i + 1]; ++j) x[Li[j]] -= Lx[j] * xi;
    }
    for (i64 i = 0; i < n; ++i) x[i] *= Dinv[i];
    for (i64 i = n - 1; i >= 0; --i) {
        double xi = x[i];
        for (i64 j = Lp[i]; j < Lp[i + 1]; ++j) xi -= Lx[j] * x[Li[j]];
        x[i] = xi;
    }
}

// qdldl.jl:642-742 (0-based; A upper triangular CSC)
void qdldl_permute_symmetric(i64 n, const i64* Ac, const i64* Ar, const double* Av, const i64* iperm,
                             i64* Pc, i64* Pr, double* Pv, i64* AtoPAPt) {
    ivec num_entries(n, 0);
    for (i64 colA = 0; colA < n; ++colA) {
        i64 colP = iperm[colA];
        for (i64 p = Ac[colA]; p < Ac[colA + 1]; ++p) {
            i64 rowA = Ar[p];
            i64 rowP = iperm[rowA];
            if (rowA <= colA) num_entries[std::max(rowP, colP)] += 1;
        }
    }
    Pc[0] = 0;
    for (i64 k = 0; k < n; ++k) { Pc[k + 1] = Pc[k] + num_entries[k]; num_entries[k] = Pc[k]; }
    for (i64 colA = 0; colA < n; ++colA) {
        i64 colP = iperm[colA];
        for (i64 p = Ac[colA]; p < Ac[colA + 1]; ++p) {
            i64 rowA = Ar[p];
            if (rowA <= colA) {
                i64 rowP = iperm[rowA];
                i64 col_idx = std::max(colP, rowP);
                i64 q = num_entries[col_idx];
                Pr[q] = std::min(colP, rowP);
                Pv[q] = Av[p];
                AtoPAPt[p] = q;
                num_entries[col_idx] += 1;
            }
        }
    }
}

// dense LU with partial pivoting: stands in for `matrix \ residual` (SparseArrays UMFPACK,
// search_direction.jl:113) — third-party, result pinned only to solve accuracy.
bool dense_lu_solve(i64 n, vec A /* col-major copy */, double* b) {
    ivec piv(n);
    for (i64 k = 0; k < n; ++k) {
        i64 p = k; double m = std::fabs(A[k + k * n]);
        for (i64 i = k + 1; i < n; ++i) if (std::fabs(A[i + k * n]) > m) { m = std::fabs(A[i + k * n]); p = i; }
        if (m == 0.0) return false;
        piv[k] = p;
        if (p != k) { for (i64 j = 0; j < n; ++j) std::swap(A[k + j * n], A[p + j * n]); std::swap(b[k], b[p]); }
        double inv = 1.0 / A[k + k * n];
        for (i64 i = k + 1; i < n; ++i) A[i + k * n] *= inv;
        for (i64 j = k + 1; j < n; ++j) {
            double akj = A[k + j * n];
            if (akj != 0.0) for (i64 i = k + 1; i < n; ++i) A[i + j * n] -= A[i + k * n] * akj;
        }
        double bk = b[k];
        for (i64 i = k + 1; i < n; ++i) b[i] -= A[i + k * n] * bk;
    }
    for (i64 k = n - 1; k >= 0; --k) {
        b[k] /= A[k + k * n];
        double bk = b[k];
        for (i64 i = 0; i < k; ++i) b[i] -= A[i + k * n] * bk;
    }
    return true;
}

// cones/second_order.jl:50-65  second_order_vector_inverse(u, x): exact inverse of arrow(u) applied to x
void second_order_vector_inverse(i64 n, const double* u, const double* x, double* out) {
    double uu = 0.0;
    for (i64 i = 1; i < n; ++i) uu += u[i] * u[i];
    double alpha = -1.0 / (u[0] * u[0]) * uu;
    double beta = 1.0 / (1.0 + alpha);
    // us = u[2:end] / u[1]
    // x0 = x - [us' * x[2:end]; 0]
    double d0 = 0.0;
    for (i64 i = 1; i < n; ++i) d0 += (u[i] / u[0]) * x[i];
    double x0_1 = x[0] - d0;
    // x1 = x - beta * [0; us * x0[1]]
    // x2 = x1 - [us' * x1[2:end]; 0]
    double d1 = 0.0;
    for (i64 i = 1; i < n; ++i) {
        out[i] = x[i] - beta * ((u[i] / u[0]) * x0_1);
        d1 += (u[i] / u[0]) * out[i];
    }
    double x2_1 = x[0] - d1;   // x1[1] == x[1]
    out[0] = 1.0 / u[0] * x2_1;
    for (i64 i = 1; i < n; ++i) out[i] = 1.0 / u[0] * out[i];
}

}  // namespace

// ------------------------------------------------------------------------------------------
struct oracle_solver {
    i64 nx, np, ne, nc, n, N;
    // Indices (indices.jl:20-63), 1-based values
    std::map<std::string, ivec> index;
    ivec nonneg;                 // 0-based cone-local
    std::vector<ivec> soc;       // 0-based cone-local
    ivec vcat_order;             // cones/cone.jl:27-59: order in which the cone functions emit entries
    Options opt;
    std::map<std::string, vec> buf;
    std::map<std::string, double*> optd;
    std::map<std::string, i64*> opti;
    // filter.jl:1-13
    std::vector<std::pair<double, double>> filter_pairs, filter_cache;
    i64 filter_index = 0;
    // linear solver (linear_solver.jl:3-8 + qdldl.jl workspace)
    ivec perm, iperm;           // 0-based
    ivec Kp, Ki;                // structural pattern of triu(K), 0-based CSC
    ivec Pp, Pi, AtoPAPt; vec Px;   // triuA = P K P' (upper)
    ivec etree, Lnz, Lp, Li; vec Lx, D, Dinv, fwork;
    std::vector<char> bwork; ivec iwork; vec ywork;
    i64 positive_inertia = -1;
    bool have_symbolic = false;
    i64 inertia[3] = {0, 0, 0};
    // stats
    std::vector<vec> trace;   // solution.all after every accepted inner iteration (test fixture support)
    i64 stat_total_iterations = 0, stat_outer = 0, stat_factorizations = 0, stat_refine_fail = 0,
        stat_refine_max = 0, stat_lu_fallback = 0, stat_last_refine_rounds = 0;

    vec& B(const char* k) { return buf[k]; }
    double* P(const char* k) { return buf[k].data(); }
    // offsets
    i64 ox() const { return 0; }
    i64 orr() const { return nx; }
    i64 os() const { return nx + ne; }
    i64 oy() const { return nx + ne + nc; }
    i64 oz() const { return nx + ne + nc + ne; }
    i64 ot() const { return nx + ne + nc + ne + nc; }
};

typedef oracle_solver S;

namespace {

ivec range1(i64 off, i64 len) { ivec v(len); for (i64 i = 0; i < len; ++i) v[i] = off + i + 1; return v; }

void build_pattern(S* s) {
    // structural pattern of triu(K) for dense blocks: the pattern the reference obtains from its
    // warm-up assembly (solver.jl:88-122) when every block entry is structurally non-zero.
    i64 nx = s->nx, ne = s->ne, nc = s->nc, n = s->n;
    // cone-block membership: for each cone-local index, list of block partners
    std::vector<ivec> partners(nc);
    for (i64 i : s->nonneg) partners[i].push_back(i);
    for (auto& c : s->soc) for (i64 i : c) for (i64 j : c) partners[j].push_back(i);   // rows i in column j
    s->Kp.assign(n + 1, 0); s->Ki.clear();
    for (i64 j = 0; j < n; ++j) {
        if (j < nx) { for (i64 i = 0; i <= j; ++i) s->Ki.push_back(i); }
        else if (j < nx + ne) { for (i64 i = 0; i < nx; ++i) s->Ki.push_back(i); s->Ki.push_back(j); }
        else {
            for (i64 i = 0; i < nx; ++i) s->Ki.push_back(i);
            i64 c = j - nx - ne;
            ivec rows;
            for (i64 i : partners[c]) if (i <= c) rows.push_back(nx + ne + i);
            if (std::find(rows.begin(), rows.end(), j) == rows.end()) rows.push_back(j);  // cone entry in no set: keep a diagonal
            std::sort(rows.begin(), rows.end());
            rows.erase(std::unique(rows.begin(), rows.end()), rows.end());
            for (i64 r : rows) s->Ki.push_back(r);
        }
        s->Kp[j + 1] = (i64)s->Ki.size();
    }
}

// ---- cones ---------------------------------------------------------------------------------
// cones/cone.jl:7-25 cone_barrier ; nonnegative.jl:11 ; second_order.jl:13
double cone_barrier(S* s, const double* x) {
    double Phi = 0.0;
    if (!s->nonneg.empty()) { double a = 0.0; for (i64 i : s->nonneg) a += std::log(x[i]); Phi += a; }
    for (auto& c : s->soc) if (!c.empty()) {
        double d = 0.0; for (size_t k = 1; k < c.size(); ++k) d += x[c[k]] * x[c[k]];
        Phi += 0.5 * std::log(x[c[0]] * x[c[0]] - d);
    }
    return Phi;
}
// cones/cone.jl:27-31 ; nonnegative.jl:12 ; second_order.jl:14   (vcat order!)
void cone_barrier_gradient(S* s, const double* x, double* out) {
    i64 k = 0;
    for (i64 i : s->nonneg) out[k++] = 1.0 / x[i];
    for (auto& c : s->soc) if (!c.empty()) {
        double d = 0.0; for (size_t q = 1; q < c.size(); ++q) d += x[c[q]] * x[c[q]];
        double sc = 1.0 / (x[c[0]] * x[c[0]] - d);
        out[k++] = sc * x[c[0]];
        for (size_t q = 1; q < c.size(); ++q) out[k++] = sc * (-x[c[q]]);
    }
}
// cones/cone.jl:34-38 ; nonnegative.jl:15 ; second_order.jl:17
void cone_product(S* s, const double* a, const double* b, double* out) {
    i64 k = 0;
    for (i64 i : s->nonneg) out[k++] = a[i] * b[i];
    for (auto& c : s->soc) if (!c.empty()) {
        double d = 0.0; for (i64 q : c) d += a[q] * b[q];
        out[k++] = d;
        for (size_t q = 1; q < c.size(); ++q) out[k++] = a[c[0]] * b[c[q]] + b[c[0]] * a[c[q]];
    }
}
// cones/cone.jl:40-45 ; nonnegative.jl:17-19 (Diagonal(b)) ; second_order.jl:19-22 (arrow(b)); block-diagonal cat, nc x nc col-major
void cone_product_jacobian(S* s, const double* /*a*/, const double* b, double* out) {
    i64 nc = s->nc;
    std::fill(out, out + nc * nc, 0.0);
    i64 k = 0;
    for (i64 i : s->nonneg) { out[k + k * nc] = b[i]; ++k; }
    for (auto& c : s->soc) if (!c.empty()) {
        i64 d = (i64)c.size();
        for (i64 p = 0; p < d; ++p) out[(k + p) + (k + p) * nc] = b[c[0]];
        for (i64 p = 1; p < d; ++p) { out[k + (k + p) * nc] = b[c[p]]; out[(k + p) + k * nc] = b[c[p]]; }
        k += d;
    }
}
// cones/cone.jl:55-59 ; nonnegative.jl:26 ; second_order.jl:42
void cone_target(S* s, double* out) {
    i64 k = 0;
    for (size_t i = 0; i < s->nonneg.size(); ++i) out[k++] = 1.0;
    for (auto& c : s->soc) if (!c.empty()) { out[k++] = 1.0; for (size_t q = 1; q < c.size(); ++q) out[k++] = 0.0; }
}
// cones/cone.jl:62-68 ; nonnegative.jl:29-34 ; second_order.jl:45-47
bool cone_violation(S* s, const double* xh, const double* x, double tau) {
    for (i64 i : s->nonneg) if (xh[i] <= (1.0 - tau) * x[i]) return true;
    for (auto& c : s->soc) if (!c.empty()) {
        double nrm = 0.0;
        for (size_t q = 1; q < c.size(); ++q) { double d = xh[c[q]] - (1.0 - tau) * x[c[q]]; nrm += d * d; }
        if (xh[c[0]] - (1.0 - tau) * x[c[0]] <= std::sqrt(nrm)) return true;
    }
    return false;
}
// cones/cone.jl:1-4 initialize_cone! ; nonnegative.jl:2-8 ; second_order.jl:2-10
void initialize_cone(S* s, double* x) {
    for (i64 i : s->nonneg) x[i] = 1.0;
    for (auto& c : s->soc) for (size_t q = 0; q < c.size(); ++q) x[c[q]] = (q == 0 ? 1.0 : 0.1);
}

double norm_inf(const double* v, i64 n) { double m = 0.0; for (i64 i = 0; i < n; ++i) m = std::max(m, std::fabs(v[i])); return m; }
double norm_1(const double* v, i64 n) { double m = 0.0; for (i64 i = 0; i < n; ++i) m += std::fabs(v[i]); return m; }
double norm_p(const double* v, i64 n, double p) {
    if (p == 1.0) return norm_1(v, n);
    if (std::isinf(p)) return norm_inf(v, n);
    if (p == 2.0) { double m = 0.0; for (i64 i = 0; i < n; ++i) m += v[i] * v[i]; return std::sqrt(m); }
    double m = 0.0; for (i64 i = 0; i < n; ++i) m += std::pow(std::fabs(v[i]), p); return std::pow(m, 1.0 / p);
}
double dot(const double* a, const double* b, i64 n) { double m = 0.0; for (i64 i = 0; i < n; ++i) m += a[i] * b[i]; return m; }

}  // namespace

extern "C" {

oracle_solver* oracle_create(int64_t nx, int64_t np, int64_t ne, int64_t nc, int64_t n_nonneg,
                             const int64_t* nonneg_idx, int64_t n_soc, const int64_t* soc_ptr,
                             const int64_t* soc_idx) {
    S* s = new S();
    s->nx = nx; s->np = np; s->ne = ne; s->nc = nc;
    s->n = nx + ne + nc;             // dimensions.jl:35
    s->N = nx + ne + nc + ne + 2 * nc;  // dimensions.jl:22-23
    // indices.jl:25-43
    s->index["variables"] = range1(0, nx);
    s->index["equality_slack"] = range1(nx, ne);
    s->index["cone_slack"] = range1(nx + ne, nc);
    s->index["equality_dual"] = range1(nx + ne + nc, ne);
    s->index["cone_dual"] = range1(nx + ne + nc + ne, nc);
    s->index["cone_slack_dual"] = range1(nx + ne + nc + ne + nc, nc);
    s->index["symmetric"] = range1(0, nx + ne + nc);
    s->index["symmetric_equality"] = range1(nx, ne);
    s->index["symmetric_cone"] = range1(nx + ne, nc);
    s->index["primals"] = range1(0, nx + ne + nc);
    s->index["duals"] = range1(nx + ne + nc, ne + nc + nc);
    s->index["violation_equality"] = range1(0, ne);
    s->index["violation_cone"] = range1(ne, nc);
    s->index["parameters"] = range1(0, np);
    s->index["cone_nonnegative"] = ivec(nonneg_idx, nonneg_idx + n_nonneg);
    ivec socflat;
    ivec socptr(soc_ptr, soc_ptr + n_soc + 1);
    for (i64 i = 0; i < n_nonneg; ++i) s->nonneg.push_back(nonneg_idx[i] - 1);
    for (i64 j = 0; j < n_soc; ++j) {
        ivec c;
        for (i64 p = soc_ptr[j]; p < soc_ptr[j + 1]; ++p) { c.push_back(soc_idx[p] - 1); socflat.push_back(soc_idx[p]); }
        s->soc.push_back(c);
    }
    s->index["cone_second_order_ptr"] = socptr;
    s->index["cone_second_order"] = socflat;
    for (i64 i : s->nonneg) s->vcat_order.push_back(i);
    for (auto& c : s->soc) for (i64 i : c) s->vcat_order.push_back(i);

    i64 n = s->n, N = s->N;
    auto mk = [&](const char* name, i64 len) { s->buf[name] = vec((size_t)len, 0.0); };
    // problem_data.jl:33-100
    mk("objective", 1); mk("objective_gradient_variables", nx); mk("objective_gradient_parameters", np);
    mk("objective_jacobian_variables_variables", nx * nx); mk("objective_jacobian_variables_parameters", nx * np);
    mk("equality_constraint", ne); mk("equality_jacobian_variables", ne * nx); mk("equality_jacobian_parameters", ne * np);
    mk("equality_dual_jacobian_variables", nx); mk("equality_dual_jacobian_variables_variables", nx * nx);
    mk("equality_dual_jacobian_variables_parameters", nx * np);
    mk("cone_constraint", nc); mk("cone_jacobian_variables", nc * nx); mk("cone_jacobian_parameters", nc * np);
    mk("cone_dual_jacobian_variables", nx); mk("cone_dual_jacobian_variables_variables", nx * nx);
    mk("cone_dual_jacobian_variables_parameters", nx * np);
    mk("cone_product", nc); mk("cone_product_jacobian_primal", nc * nc); mk("cone_product_jacobian_dual", nc * nc);
    mk("cone_target", nc); mk("barrier", 1); mk("barrier_gradient", n);
    // solver_data.jl:26-93 and solver.jl:76-127
    mk("solution", N); mk("candidate", N); mk("parameters", np);
    mk("residual", N); mk("residual_error", N); mk("step", N); mk("step_correction", N);
    mk("residual_symmetric", n); mk("step_symmetric", n); mk("jacobian_variables_symmetric", n * n);
    mk("merit_gradient", n); mk("constraint_violation", ne + nc);
    mk("jacobian_parameters", N * np); mk("solution_sensitivity", N * np);
    mk("jacobian_parameters_vector", N); mk("solution_sensitivity_vector", N);
    mk("central_path", 1); mk("fraction_to_boundary", 1); mk("penalty", 1); mk("dual", ne);
    mk("primal_regularization", 1); mk("primal_regularization_last", 1); mk("dual_regularization", 1);
    s->buf["central_path"][0] = 0.1; s->buf["fraction_to_boundary"][0] = 0.99; s->buf["penalty"][0] = 10.0;  // solver.jl:81-85
    // non-zero blocks of the unreduced Jacobian H (residual_jacobian_variables.jl:1-108)
    mk("H.xx", nx * nx); mk("H.rr", ne); mk("H.ss", nc); mk("H.yy", ne); mk("H.zz", nc);
    mk("H.ts", nc * nc); mk("H.tt", nc * nc);
    mk("scratch.n", n); mk("scratch.N", N);

    Options& o = s->opt;
#define OD(f) s->optd["opt." #f] = &o.f
#define OI(f) s->opti[#f] = &o.f
    OD(residual_norm); OD(constraint_norm); OD(scaling_line_search); OD(iterative_refinement_tolerance);
    OD(central_path_initial); OD(central_path_update_tolerance); OD(central_path_scaling); OD(central_path_exponent);
    OD(penalty_initial); OD(penalty_scaling); OD(dual_initial); OD(residual_tolerance); OD(optimality_tolerance);
    OD(slack_tolerance); OD(equality_tolerance); OD(complementarity_tolerance); OD(min_regularization);
    OD(primal_regularization_initial); OD(dual_regularization_initial); OD(max_regularization); OD(dual_regularization);
    OD(dual_regularization_exponent); OD(scaling_regularization_initial); OD(scaling_regularization);
    OD(scaling_regularization_last); OD(min_central_path); OD(max_penalty); OD(violation_tolerance);
    OD(violation_exponent); OD(merit_tolerance); OD(merit_exponent); OD(armijo_tolerance); OD(machine_tolerance);
    OI(max_outer_iterations); OI(max_residual_iterations); OI(max_residual_line_search); OI(max_cone_line_search);
    OI(iterative_refinement); OI(max_iterative_refinement); OI(min_iterative_refinement); OI(constraint_tensor);
    OI(update_factorization); OI(max_filter); OI(differentiate); OI(warmstart); OI(linear_solve_refactor);
#undef OD
#undef OI
    s->filter_pairs.assign(o.max_filter, {1.0e8, 1.0e8});
    s->filter_cache.assign(o.max_filter, {1.0e8, 1.0e8});
    // default elimination order: constraint-first [z | y | x]  (AMD.jl is third-party and absent: parity unpinned)
    s->perm.resize(n);
    { i64 k = 0;
      for (i64 i = 0; i < nc; ++i) s->perm[k++] = nx + ne + i;
      for (i64 i = 0; i < ne; ++i) s->perm[k++] = nx + i;
      for (i64 i = 0; i < nx; ++i) s->perm[k++] = i; }
    s->iperm.resize(n);
    for (i64 i = 0; i < n; ++i) s->iperm[s->perm[i]] = i;
    build_pattern(s);
    return s;
}

void oracle_destroy(oracle_solver* s) { delete s; }

double* oracle_buffer(oracle_solver* s, const char* name, int64_t* len) {
    auto it = s->buf.find(name);
    if (it != s->buf.end()) { if (len) *len = (i64)it->second.size(); return it->second.data(); }
    auto jt = s->optd.find(name);
    if (jt != s->optd.end()) { if (len) *len = 1; return jt->second; }
    if (len) *len = -1;
    return nullptr;
}
const int64_t* oracle_index(oracle_solver* s, const char* name, int64_t* len) {
    auto it = s->index.find(name);
    if (it == s->index.end()) { if (len) *len = -1; return nullptr; }
    if (len) *len = (i64)it->second.size();
    return it->second.data();
}
int64_t* oracle_int(oracle_solver* s, const char* name) {
    auto it = s->opti.find(name);
    return it == s->opti.end() ? nullptr : it->second;
}
void oracle_set_perm(oracle_solver* s, const int64_t* perm) {
    for (i64 i = 0; i < s->n; ++i) s->perm[i] = perm[i] - 1;
    for (i64 i = 0; i < s->n; ++i) s->iperm[s->perm[i]] = i;
    s->have_symbolic = false;
}

// cones/cone.jl:71-106
void oracle_cone(oracle_solver* s, int which, int barrier, int barrier_gradient, int product, int jacobian, int target) {
    double* w = s->P(which == 0 ? "solution" : "candidate");
    const double* sl = w + s->os();
    const double* t = w + s->ot();
    i64 nc = s->nc;
    if (barrier) s->P("barrier")[0] = cone_barrier(s, sl);
    if (barrier_gradient) cone_barrier_gradient(s, sl, s->P("barrier_gradient"));
    if (product && nc > 0) cone_product(s, sl, t, s->P("cone_product"));
    if (jacobian && nc > 0) {
        cone_product_jacobian(s, sl, t, s->P("cone_product_jacobian_primal"));   // d(s o t)/ds = J(t)
        cone_product_jacobian(s, t, sl, s->P("cone_product_jacobian_dual"));     // d(s o t)/dt = J(s)
    }
    if (target && nc > 0) cone_target(s, s->P("cone_target"));
}

int oracle_cone_violation(oracle_solver* s, const double* xhat, const double* x, double tau) {
    return cone_violation(s, xhat, x, tau) ? 1 : 0;
}

// residual.jl:1-51
void oracle_residual(oracle_solver* s) {
    i64 nx = s->nx, ne = s->ne, nc = s->nc;
    double* w = s->P("solution");
    const double *r = w + s->orr(), *sl = w + s->os(), *y = w + s->oy(), *z = w + s->oz(), *t = w + s->ot();
    double* res = s->P("residual");
    double kappa = s->P("central_path")[0], rho = s->P("penalty")[0];
    const double* lam = s->P("dual");
    std::fill(res, res + s->N, 0.0);
    for (i64 i = 0; i < nx; ++i) res[i] = s->P("objective_gradient_variables")[i];
    for (i64 i = 0; i < nx; ++i) {
        res[i] += s->P("equality_dual_jacobian_variables")[i];
        res[i] += s->P("cone_dual_jacobian_variables")[i];
    }
    for (i64 i = 0; i < ne; ++i) res[s->orr() + i] = lam[i] + rho * r[i] - y[i];
    for (i64 i = 0; i < nc; ++i) res[s->os() + i] = -z[i] - t[i];
    for (i64 i = 0; i < ne; ++i) res[s->oy() + i] = s->P("equality_constraint")[i];
    for (i64 i = 0; i < ne; ++i) res[s->oy() + i] -= r[i];
    for (i64 i = 0; i < nc; ++i) res[s->oz() + i] = s->P("cone_constraint")[i];
    for (i64 i = 0; i < nc; ++i) res[s->oz() + i] -= sl[i];
    for (i64 i = 0; i < nc; ++i) res[s->ot() + i] = s->P("cone_product")[i] - kappa * s->P("cone_target")[i];
}

// residual_jacobian_variables.jl:1-108 — the non-zero blocks of H
void oracle_residual_jacobian_variables(oracle_solver* s) {
    i64 nx = s->nx, ne = s->ne, nc = s->nc;
    double rho = s->P("penalty")[0];
    double ep = s->P("primal_regularization")[0], ed = s->P("dual_regularization")[0];
    double* Hxx = s->P("H.xx");
    const double* fxx = s->P("objective_jacobian_variables_variables");
    const double* gyxx = s->P("equality_dual_jacobian_variables_variables");
    const double* hzxx = s->P("cone_dual_jacobian_variables_variables");
    bool ct = s->opt.constraint_tensor != 0;
    for (i64 i = 0; i < nx; ++i) for (i64 j = 0; j < nx; ++j) {
        double v = fxx[i + j * nx];
        if (ct) v += gyxx[i + j * nx];
        if (ct) v += hzxx[i + j * nx];
        Hxx[i + j * nx] = v;
    }
    for (i64 i = 0; i < ne; ++i) s->P("H.rr")[i] = rho;                 // :59-62
    double* Hts = s->P("H.ts"); double* Htt = s->P("H.tt");
    std::fill(Hts, Hts + nc * nc, 0.0); std::fill(Htt, Htt + nc * nc, 0.0);
    const double* Jp = s->P("cone_product_jacobian_primal");
    const double* Jd = s->P("cone_product_jacobian_dual");
    for (i64 i : s->nonneg) { Hts[i + i * nc] = Jp[i + i * nc]; Htt[i + i * nc] = Jd[i + i * nc]; }   // :64-68
    for (auto& c : s->soc) for (i64 i : c) for (i64 j : c) {                                          // :70-80
        Hts[i + j * nc] = Jp[i + j * nc]; Htt[i + j * nc] = Jd[i + j * nc];
    }
    // regularisation :82-105
    for (i64 i = 0; i < nx; ++i) Hxx[i + i * nx] += ep;
    for (i64 i = 0; i < ne; ++i) s->P("H.rr")[i] += ep;
    for (i64 i = 0; i < nc; ++i) s->P("H.ss")[i] = 0.0 + ep;
    for (i64 i = 0; i < ne; ++i) s->P("H.yy")[i] = 0.0 - ed;
    for (i64 i = 0; i < nc; ++i) s->P("H.zz")[i] = 0.0 - ed;
    for (i64 i = 0; i < nc; ++i) Htt[i + i * nc] -= ed;
}

void oracle_H_dense(oracle_solver* s, double* H) {
    i64 nx = s->nx, ne = s->ne, nc = s->nc, N = s->N;
    std::fill(H, H + N * N, 0.0);
    auto at = [&](i64 i, i64 j) -> double& { return H[i + j * N]; };
    const double* gx = s->P("equality_jacobian_variables");
    const double* hx = s->P("cone_jacobian_variables");
    for (i64 i = 0; i < nx; ++i) for (i64 j = 0; j < nx; ++j) at(i, j) = s->P("H.xx")[i + j * nx];
    for (i64 i = 0; i < ne; ++i) { at(s->orr() + i, s->oy() + i) = -1.0; at(s->oy() + i, s->orr() + i) = -1.0; }
    for (i64 i = 0; i < nc; ++i) { at(s->os() + i, s->oz() + i) = -1.0; at(s->oz() + i, s->os() + i) = -1.0; at(s->os() + i, s->ot() + i) = -1.0; }
    for (i64 i = 0; i < ne; ++i) for (i64 j = 0; j < nx; ++j) { at(s->oy() + i, j) = gx[i + j * ne]; at(j, s->oy() + i) = gx[i + j * ne]; }
    for (i64 i = 0; i < nc; ++i) for (i64 j = 0; j < nx; ++j) { at(s->oz() + i, j) = hx[i + j * nc]; at(j, s->oz() + i) = hx[i + j * nc]; }
    for (i64 i = 0; i < ne; ++i) { at(s->orr() + i, s->orr() + i) = s->P("H.rr")[i]; at(s->oy() + i, s->oy() + i) = s->P("H.yy")[i]; }
    for (i64 i = 0; i < nc; ++i) { at(s->os() + i, s->os() + i) = s->P("H.ss")[i]; at(s->oz() + i, s->oz() + i) = s->P("H.zz")[i]; }
    for (i64 i = 0; i < nc; ++i) for (i64 j = 0; j < nc; ++j) {
        at(s->ot() + i, s->os() + j) = s->P("H.ts")[i + j * nc];
        at(s->ot() + i, s->ot() + j) = s->P("H.tt")[i + j * nc];
    }
}

// mul!(e, H, v) as used by iterative_refinement.jl:9,39 (block form)
void oracle_H_mul(oracle_solver* s, const double* v, double* out) {
    i64 nx = s->nx, ne = s->ne, nc = s->nc;
    const double *vx = v, *vr = v + s->orr(), *vs = v + s->os(), *vy = v + s->oy(), *vz = v + s->oz(), *vt = v + s->ot();
    const double* Hxx = s->P("H.xx");
    const double* gx = s->P("equality_jacobian_variables");
    const double* hx = s->P("cone_jacobian_variables");
    for (i64 i = 0; i < s->N; ++i) out[i] = 0.0;
    for (i64 j = 0; j < nx; ++j) { double vj = vx[j]; for (i64 i = 0; i < nx; ++i) out[i] += Hxx[i + j * nx] * vj; }
    for (i64 j = 0; j < nx; ++j) {
        double a = 0.0; for (i64 i = 0; i < ne; ++i) a += gx[i + j * ne] * vy[i];
        double b = 0.0; for (i64 i = 0; i < nc; ++i) b += hx[i + j * nc] * vz[i];
        out[j] += a; out[j] += b;
    }
    for (i64 i = 0; i < ne; ++i) out[s->orr() + i] = s->P("H.rr")[i] * vr[i] - vy[i];
    for (i64 i = 0; i < nc; ++i) out[s->os() + i] = s->P("H.ss")[i] * vs[i] - vz[i] - vt[i];
    for (i64 j = 0; j < nx; ++j) {
        double vj = vx[j];
        for (i64 i = 0; i < ne; ++i) out[s->oy() + i] += gx[i + j * ne] * vj;
        for (i64 i = 0; i < nc; ++i) out[s->oz() + i] += hx[i + j * nc] * vj;
    }
    for (i64 i = 0; i < ne; ++i) out[s->oy() + i] += -vr[i] + s->P("H.yy")[i] * vy[i];
    for (i64 i = 0; i < nc; ++i) out[s->oz() + i] += -vs[i] + s->P("H.zz")[i] * vz[i];
    const double* Hts = s->P("H.ts"); const double* Htt = s->P("H.tt");
    auto blockmul = [&](const ivec& c) {
        for (i64 i : c) { double a = 0.0; for (i64 j : c) a += Hts[i + j * nc] * vs[j] + Htt[i + j * nc] * vt[j]; out[s->ot() + i] = a; }
    };
    for (i64 i : s->nonneg) out[s->ot() + i] = Hts[i + i * nc] * vs[i] + Htt[i + i * nc] * vt[i];
    for (auto& c : s->soc) blockmul(c);
}

// residual_jacobian_variables.jl:110-167 — condensed K, dense n x n, both triangles written
void oracle_residual_jacobian_variables_symmetric(oracle_solver* s) {
    i64 nx = s->nx, ne = s->ne, nc = s->nc, n = s->n;
    double* K = s->P("jacobian_variables_symmetric");
    std::fill(K, K + n * n, 0.0);
    const double* Hxx = s->P("H.xx");
    const double* gx = s->P("equality_jacobian_variables");
    const double* hx = s->P("cone_jacobian_variables");
    const double* Hts = s->P("H.ts"); const double* Htt = s->P("H.tt");
    for (i64 i = 0; i < nx; ++i) for (i64 j = 0; j < nx; ++j) K[i + j * n] = Hxx[i + j * nx];
    for (i64 i = 0; i < ne; ++i) for (i64 j = 0; j < nx; ++j) { K[(nx + i) + j * n] = gx[i + j * ne]; K[j + (nx + i) * n] = gx[i + j * ne]; }
    for (i64 i = 0; i < ne; ++i) K[(nx + i) + (nx + i) * n] = -1.0 / s->P("H.rr")[i] + s->P("H.yy")[i];
    i64 sz = nx + ne;
    for (i64 i = 0; i < nc; ++i) for (i64 j = 0; j < nx; ++j) { K[(sz + i) + j * n] = hx[i + j * nc]; K[j + (sz + i) * n] = hx[i + j * nc]; }
    for (i64 i : s->nonneg) {
        double Sb = Htt[i + i * nc], Ti = Hts[i + i * nc], Pi = s->P("H.ss")[i], Di = s->P("H.zz")[i];
        K[(sz + i) + (sz + i) * n] += -1.0 * Sb / (Ti + Sb * Pi) + Di;
    }
    for (auto& c : s->soc) if (!c.empty()) {
        i64 d = (i64)c.size();
        vec U(d * d), u(d), col(d), outc(d);
        // Cs + Cbar_t * P  (P = H[s,s] block: diagonal)
        for (i64 a = 0; a < d; ++a) for (i64 b = 0; b < d; ++b) {
            double acc = 0.0;
            for (i64 l = 0; l < d; ++l) acc += Htt[c[a] + c[l] * nc] * (l == b ? s->P("H.ss")[c[b]] : 0.0);
            U[a + b * d] = Hts[c[a] + c[b] * nc] + acc;
        }
        for (i64 b = 0; b < d; ++b) u[b] = U[0 + b * d];   // second_order_matrix_inverse uses U[1, :]  (second_order.jl:63-65)
        for (i64 i = 0; i < d; ++i) {
            for (i64 a = 0; a < d; ++a) col[a] = Htt[c[a] + c[i] * nc];
            second_order_vector_inverse(d, u.data(), col.data(), outc.data());
            for (i64 a = 0; a < d; ++a) K[(sz + c[a]) + (sz + c[i]) * n] -= outc[a];
        }
        for (i64 a = 0; a < d; ++a) for (i64 b = 0; b < d; ++b)
            K[(sz + c[a]) + (sz + c[b]) * n] += (a == b ? s->P("H.zz")[c[a]] : 0.0);
    }
}

// residual.jl:53-101
void oracle_residual_symmetric(oracle_solver* s, int which) {
    i64 nx = s->nx, ne = s->ne, nc = s->nc;
    const double* res = s->P(which == 0 ? "residual" : (which == 1 ? "residual_error" : "jacobian_parameters_vector"));
    double* rsym = s->P("residual_symmetric");
    const double *rx = res, *rr = res + s->orr(), *rs = res + s->os(), *ry = res + s->oy(), *rz = res + s->oz(), *rt = res + s->ot();
    std::fill(rsym, rsym + s->n, 0.0);
    for (i64 i = 0; i < nx; ++i) rsym[i] = rx[i];
    for (i64 i = 0; i < ne; ++i) rsym[nx + i] = ry[i];
    for (i64 i = 0; i < nc; ++i) rsym[nx + ne + i] = rz[i];
    for (i64 i = 0; i < ne; ++i) rsym[nx + i] += rr[i] / s->P("H.rr")[i];
    const double* Hts = s->P("H.ts"); const double* Htt = s->P("H.tt");
    for (i64 i : s->nonneg) {
        double Sb = Htt[i + i * nc], Ti = Hts[i + i * nc], Pi = s->P("H.ss")[i];
        rsym[nx + ne + i] += (rt[i] + Sb * rs[i]) / (Ti + Sb * Pi);
    }
    for (auto& c : s->soc) if (!c.empty()) {
        i64 d = (i64)c.size();
        vec u(d), v(d), o(d);
        for (i64 b = 0; b < d; ++b) u[b] = Hts[c[0] + c[b] * nc] + Htt[c[0] + c[b] * nc] * s->P("H.ss")[c[b]];
        for (i64 a = 0; a < d; ++a) { double acc = 0.0; for (i64 b = 0; b < d; ++b) acc += Htt[c[a] + c[b] * nc] * rs[c[b]]; v[a] = acc + rt[c[a]]; }
        second_order_vector_inverse(d, u.data(), v.data(), o.data());
        for (i64 a = 0; a < d; ++a) rsym[nx + ne + c[a]] += o[a];
    }
}

// linear_solver.jl:19-31 + qdldl.jl:134-188 (fresh) / :199-213,269-278 (update)
int64_t oracle_factorize(oracle_solver* s, int update) {
    i64 n = s->n;
    const double* K = s->P("jacobian_variables_symmetric");
    i64 nnz = s->Kp[n];
    vec Ax((size_t)nnz);
    for (i64 j = 0; j < n; ++j) for (i64 p = s->Kp[j]; p < s->Kp[j + 1]; ++p) Ax[p] = K[s->Ki[p] + j * n];   // triu(K).nzval
    if (!update || !s->have_symbolic) {
        s->Pp.assign(n + 1, 0); s->Pi.assign(nnz, 0); s->Px.assign(nnz, 0.0); s->AtoPAPt.assign(nnz, 0);
        qdldl_permute_symmetric(n, s->Kp.data(), s->Ki.data(), Ax.data(), s->iperm.data(), s->Pp.data(), s->Pi.data(), s->Px.data(), s->AtoPAPt.data());
        s->etree.assign(n, 0); s->Lnz.assign(n, 0); ivec work(n);
        i64 sumLnz = qdldl_etree(n, s->Pp.data(), s->Pi.data(), work.data(), s->Lnz.data(), s->etree.data());
        if (sumLnz < 0) return -2;
        s->Lp.assign(n + 1, 0); s->Li.assign(sumLnz, 0); s->Lx.assign(sumLnz, 0.0);
        s->D.assign(n, 0.0); s->Dinv.assign(n, 0.0); s->fwork.assign(n, 0.0);
        s->have_symbolic = true;
    } else {
        for (i64 p = 0; p < nnz; ++p) s->Px[s->AtoPAPt[p]] = Ax[p];   // update_values!(F, 1:nnz, A.nzval)
    }
    s->positive_inertia = qdldl_factor(n, s->Pp.data(), s->Pi.data(), s->Px.data(), s->Lp.data(), s->Li.data(), s->Lx.data(),
                                       s->D.data(), s->Dinv.data(), s->Lnz.data(), s->etree.data(), s->bwork, s->iwork, s->ywork);
    s->stat_factorizations += 1;
    return s->positive_inertia;
}

// linear_solver.jl:33-44
void oracle_compute_inertia(oracle_solver* s, int64_t out[3]) {
    s->inertia[0] = s->positive_inertia;
    i64 ng = 0, z = 0;
    for (double d : s->D) { if (d <= 0.0) ng += 1; if (d == 0.0) z += 1; }
    s->inertia[1] = ng; s->inertia[2] = z;
    if (out) { out[0] = s->inertia[0]; out[1] = s->inertia[1]; out[2] = s->inertia[2]; }
}

// linear_solver.jl:52-60 + qdldl.jl:330-351
void oracle_linear_solve(oracle_solver* s, double* x, const double* b, int fact, int update) {
    i64 n = s->n;
    if (fact) oracle_factorize(s, update);
    for (i64 i = 0; i < n; ++i) x[i] = b[i];
    for (i64 j = 0; j < n; ++j) s->fwork[j] = x[s->perm[j]];                       // permute!
    qdldl_solve(n, s->Lp.data(), s->Li.data(), s->Lx.data(), s->Dinv.data(), s->fwork.data());
    for (i64 j = 0; j < n; ++j) x[s->perm[j]] = s->fwork[j];                       // ipermute!
}

namespace {
// search_direction.jl:25-104 on explicit vectors
void search_direction_symmetric(S* s, double* step, const double* res, int which_res, int fact, int update) {
    i64 nx = s->nx, ne = s->ne, nc = s->nc;
    oracle_residual_symmetric(s, which_res);
    oracle_linear_solve(s, s->P("step_symmetric"), s->P("residual_symmetric"), fact, update);
    const double* dsym = s->P("step_symmetric");
    const double *dx = dsym, *dy = dsym + nx, *dz = dsym + nx + ne;
    for (i64 i = 0; i < nx; ++i) step[i] = dx[i];
    for (i64 i = 0; i < ne; ++i) step[s->oy() + i] = dy[i];
    for (i64 i = 0; i < nc; ++i) step[s->oz() + i] = dz[i];
    double *Dr = step + s->orr(), *Ds = step + s->os(), *Dt = step + s->ot();
    const double *rr = res + s->orr(), *rs = res + s->os(), *rt = res + s->ot();
    for (i64 i = 0; i < ne; ++i) Dr[i] = (rr[i] + dy[i]) / s->P("H.rr")[i];
    const double* Hts = s->P("H.ts"); const double* Htt = s->P("H.tt");
    for (i64 i : s->nonneg) {
        double Sb = Htt[i + i * nc], Ti = Hts[i + i * nc], Pi = s->P("H.ss")[i];
        Ds[i] = (rt[i] + Sb * (rs[i] + dz[i])) / (Ti + Sb * Pi);
        Dt[i] = (rt[i] - Ti * Ds[i]) / Sb;
    }
    for (auto& c : s->soc) if (!c.empty()) {
        i64 d = (i64)c.size();
        vec u(d), v(d), o(d), ct(d);
        for (i64 b = 0; b < d; ++b) u[b] = Hts[c[0] + c[b] * nc] + Htt[c[0] + c[b] * nc] * s->P("H.ss")[c[b]];
        for (i64 a = 0; a < d; ++a) { double acc = 0.0; for (i64 b = 0; b < d; ++b) acc += Htt[c[a] + c[b] * nc] * (rs[c[b]] + dz[c[b]]); v[a] = rt[c[a]] + acc; }
        second_order_vector_inverse(d, u.data(), v.data(), o.data());
        for (i64 a = 0; a < d; ++a) Ds[c[a]] = o[a];
        for (i64 b = 0; b < d; ++b) ct[b] = Htt[c[0] + c[b] * nc];
        for (i64 a = 0; a < d; ++a) { double acc = 0.0; for (i64 b = 0; b < d; ++b) acc += Hts[c[a] + c[b] * nc] * Ds[c[b]]; v[a] = rt[c[a]] - acc; }
        second_order_vector_inverse(d, ct.data(), v.data(), o.data());
        for (i64 a = 0; a < d; ++a) Dt[c[a]] = o[a];
    }
}

// search_direction.jl:106-119 `matrix \ residual` (dense LU stand-in; only for small N)
bool search_direction_nonsymmetric(S* s, double* step, const double* res) {
    i64 N = s->N;
    vec H((size_t)(N * N));
    oracle_H_dense(s, H.data());
    for (i64 i = 0; i < N; ++i) step[i] = res[i];
    s->stat_lu_fallback += 1;
    return dense_lu_solve(N, H, step);
}
}  // namespace

void oracle_search_direction_symmetric(oracle_solver* s, int which, int fact) {
    if (which == 0) search_direction_symmetric(s, s->P("step"), s->P("residual"), 0, fact, (int)s->opt.update_factorization);
    else if (which == 1) search_direction_symmetric(s, s->P("step_correction"), s->P("residual_error"), 1, fact, 1);
    else search_direction_symmetric(s, s->P("solution_sensitivity_vector"), s->P("jacobian_parameters_vector"), 2, fact, 1);
}

// iterative_refinement.jl:1-52 (refines buf "step")
int oracle_iterative_refinement(oracle_solver* s) {
    i64 N = s->N;
    double* step = s->P("step"); double* corr = s->P("step_correction"); double* err = s->P("residual_error");
    const double* res = s->P("residual"); double* tmp = s->P("scratch.N");
    std::fill(corr, corr + N, 0.0); std::fill(err, err + N, 0.0);
    i64 iteration = 0;
    oracle_H_mul(s, step, tmp);
    for (i64 i = 0; i < N; ++i) err[i] = res[i] - tmp[i];
    double residual_norm = norm_inf(err, N);
    double residual_norm_initial = residual_norm;
    int fact = s->opt.linear_solve_refactor ? 1 : 0;
    while (iteration <= s->opt.max_iterative_refinement) {
        if (residual_norm <= s->opt.iterative_refinement_tolerance && iteration >= s->opt.min_iterative_refinement) {
            s->stat_last_refine_rounds = iteration; s->stat_refine_max = std::max(s->stat_refine_max, iteration);
            return 1;
        }
        search_direction_symmetric(s, corr, err, 1, fact, 1);   // update=true default (search_direction.jl:36)
        for (i64 i = 0; i < N; ++i) step[i] += corr[i];
        oracle_H_mul(s, step, tmp);
        for (i64 i = 0; i < N; ++i) err[i] = res[i] - tmp[i];
        residual_norm = norm_inf(err, N);
        iteration += 1;
    }
    s->stat_last_refine_rounds = iteration; s->stat_refine_max = std::max(s->stat_refine_max, iteration);
    if (residual_norm <= residual_norm_initial) return 1;
    s->stat_refine_fail += 1;
    return 0;
}

namespace {
// inertia.jl:7-11
bool inertia_ok(S* s) { return s->inertia[0] == s->nx && s->inertia[1] == s->ne + s->nc && s->inertia[2] == 0; }
// inertia.jl:13-28
void factorize_regularized(S* s) {
    oracle_residual_jacobian_variables(s);
    oracle_residual_jacobian_variables_symmetric(s);
    oracle_factorize(s, (int)s->opt.update_factorization);
    oracle_compute_inertia(s, nullptr);
}
}  // namespace

// inertia.jl:30-80 (quirk B-1: the `primal_regularization_last == 0.0` test at :48 compares a Vector with a
// Float64 and is always false, so IC-3 always takes the else branch)
int oracle_inertia_correction(oracle_solver* s) {
    double& ep = s->P("primal_regularization")[0];
    double& ed = s->P("dual_regularization")[0];
    double& ep_last = s->P("primal_regularization_last")[0];
    ep = s->opt.primal_regularization_initial;
    ed = s->opt.dual_regularization_initial;
    factorize_regularized(s);                                   // IC-1
    if (inertia_ok(s)) return 0;
    if (s->inertia[2] != 0)                                     // IC-2
        ed = s->opt.dual_regularization * std::pow(s->P("central_path")[0], s->opt.dual_regularization_exponent);
    ep = std::max(s->opt.min_regularization, s->opt.scaling_regularization_last * ep_last);   // IC-3 (else branch)
    while (!inertia_ok(s)) {
        factorize_regularized(s);                               // IC-4
        if (inertia_ok(s)) break;
        if (ep_last == 0.0) ep = s->opt.scaling_regularization_initial * ep;   // IC-5
        else ep = s->opt.scaling_regularization * ep;
        if (ep > s->opt.max_regularization) return -1;          // IC-6 error("inertia correction failure")
    }
    ep_last = ep;
    return 0;
}

// search_direction.jl:1-23 (linear_solver == :QDLDL)
int oracle_search_direction(oracle_solver* s) {
    int ic = oracle_inertia_correction(s);
    if (ic < 0) return ic;
    int fact = s->opt.linear_solve_refactor ? 1 : 0;
    search_direction_symmetric(s, s->P("step"), s->P("residual"), 0, fact, (int)s->opt.update_factorization);
    if (s->opt.iterative_refinement) {
        if (!oracle_iterative_refinement(s)) { search_direction_nonsymmetric(s, s->P("step"), s->P("residual")); return 2; }
    }
    return 0;
}

// merit.jl:2-15
double oracle_merit(oracle_solver* s, double f, const double* r, double Phi) {
    double kappa = s->P("central_path")[0], rho = s->P("penalty")[0];
    const double* lam = s->P("dual");
    double M = 0.0;
    M += f;
    M += dot(lam, r, s->ne) + 0.5 * rho * dot(r, r, s->ne);
    M -= kappa * Phi;
    return M;
}
// merit.jl:17-31
void oracle_merit_gradient(oracle_solver* s) {
    double* grad = s->P("merit_gradient");
    const double* fx = s->P("objective_gradient_variables");
    const double* r = s->P("solution") + s->orr();
    const double* Phis = s->P("barrier_gradient");
    double kappa = s->P("central_path")[0], rho = s->P("penalty")[0];
    const double* lam = s->P("dual");
    for (i64 i = 0; i < s->nx; ++i) grad[i] = fx[i];
    for (i64 i = 0; i < s->ne; ++i) grad[s->nx + i] = lam[i] + rho * r[i];
    for (i64 i = 0; i < s->nc; ++i) grad[s->nx + s->ne + i] = -1.0 * kappa * Phis[i];
}
// constraint_violation.jl:1-13
double oracle_constraint_violation(oracle_solver* s, const double* g, const double* r, const double* h, const double* sl) {
    double* c = s->P("constraint_violation");
    for (i64 i = 0; i < s->ne; ++i) c[i] = g[i] - r[i];
    for (i64 j = 0; j < s->nc; ++j) c[s->ne + j] = h[j] - sl[j];
    i64 len = s->ne + s->nc;
    return norm_p(c, len, s->opt.constraint_norm) / (double)len;
}
// optimality_error.jl:1-27
double oracle_optimality_error(oracle_solver* s) {
    const double* w = s->P("solution"); const double* res = s->P("residual");
    i64 ne = s->ne, nc = s->nc;
    const double *y = w + s->oy(), *z = w + s->oz(), *t = w + s->ot();
    double sd = (ne + nc > 0) ? std::max(100.0, (norm_1(y, ne) + norm_1(z, nc)) / (double)(ne + nc)) / 100.0 : 1.0;
    double sc = (nc > 0) ? std::max(100.0, norm_1(t, nc) / (double)nc) / 100.0 : 1.0;
    double a = norm_inf(res, s->n) / sd;
    double b = norm_inf(res + s->oy(), ne);
    double c = norm_inf(res + s->oz(), nc);
    double d = norm_inf(res + s->ot(), nc) / sc;
    return std::max(std::max(a, b), std::max(c, d));
}

// filter.jl:22-41
void oracle_filter_reset(oracle_solver* s) {
    for (i64 i = 0; i < s->filter_index; ++i) s->filter_cache[i] = {1.0e8, 1.0e8};
    for (i64 i = 0; i < s->filter_index; ++i) s->filter_pairs[i] = {1.0e8, 1.0e8};
    s->filter_index = 0;
}
// filter.jl:43-50
int oracle_check_filter(oracle_solver* s, double theta, double merit) {
    for (auto& f : s->filter_pairs) if (!(theta < f.first || merit < f.second)) return 0;
    return 1;
}
// filter.jl:52-79
void oracle_augment_filter(oracle_solver* s, double theta, double merit) {
    if (s->filter_index == 0) { s->filter_pairs[0] = {theta, merit}; s->filter_index += 1; return; }
    if (oracle_check_filter(s, theta, merit)) {
        i64 nold = s->filter_index;
        for (i64 i = 0; i < nold; ++i) s->filter_cache[i] = s->filter_pairs[i];
        for (i64 i = 0; i < nold; ++i) s->filter_pairs[i] = {1.0e8, 1.0e8};
        s->filter_index = 0;
        s->filter_pairs[s->filter_index++] = {theta, merit};
        for (i64 i = 0; i < nold; ++i)
            if (!(s->filter_cache[i].first >= theta && s->filter_cache[i].second >= merit))
                s->filter_pairs[s->filter_index++] = s->filter_cache[i];
    }
}
int64_t oracle_filter_pairs(oracle_solver* s, double* out) {
    for (i64 i = 0; i < (i64)s->filter_pairs.size(); ++i) { out[2 * i] = s->filter_pairs[i].first; out[2 * i + 1] = s->filter_pairs[i].second; }
    return s->filter_index;
}
// line_search.jl:2-6
int oracle_switching_condition(double step_size, const double* dir, const double* grad, int64_t n, double merit_exponent,
                               double violation, double violation_exponent, double regularization) {
    double d = dot(grad, dir, n);
    return (d < 0.0 && step_size * std::pow(-d, merit_exponent) > regularization * std::pow(violation, violation_exponent)) ? 1 : 0;
}
// line_search.jl:9-12
int oracle_sufficient_progress(double violation, double violation_candidate, double merit, double merit_candidate,
                               double violation_tolerance, double merit_tolerance, double machine_tolerance) {
    return (violation_candidate - 10.0 * machine_tolerance * std::fabs(violation) <= (1.0 - violation_tolerance) * violation ||
            merit_candidate - 10.0 * machine_tolerance * std::fabs(merit) <= merit - merit_tolerance * violation) ? 1 : 0;
}
// line_search.jl:15-18
int oracle_armijo(double merit, double merit_candidate, const double* grad, const double* dir, int64_t n, double step_size,
                  double armijo_tolerance, double machine_tolerance) {
    double d = dot(grad, dir, n);
    return (merit_candidate - merit - 10.0 * machine_tolerance * std::fabs(merit) <= armijo_tolerance * step_size * d) ? 1 : 0;
}

int64_t oracle_trace(oracle_solver* s, double* out, int64_t cap_rows) {
    i64 n = std::min<i64>((i64)s->trace.size(), cap_rows);
    if (out) for (i64 k = 0; k < n; ++k) std::copy(s->trace[k].begin(), s->trace[k].end(), out + k * s->N);
    return (i64)s->trace.size();
}

void oracle_stats(oracle_solver* s, int64_t out[8]) {
    out[0] = s->stat_total_iterations; out[1] = s->stat_outer; out[2] = s->stat_factorizations; out[3] = s->stat_refine_fail;
    out[4] = s->stat_refine_max; out[5] = s->stat_lu_fallback; out[6] = s->stat_last_refine_rounds; out[7] = 0;
}

namespace {
int call_eval(S* s, oracle_eval_fn eval, void* user, const double* w, uint32_t flags) {
    return eval(user, flags, w, w + s->oy(), w + s->oz(), s->P("parameters"));
}
}  // namespace

// solve.jl:8-377
int oracle_solve(oracle_solver* s, oracle_eval_fn eval, void* user) {
    i64 nx = s->nx, ne = s->ne, nc = s->nc, N = s->N;
    Options& o = s->opt;
    double* w = s->P("solution"); double* wc = s->P("candidate");
    double *x = w, *r = w + s->orr(), *sl = w + s->os(), *y = w + s->oy(), *z = w + s->oz(), *t = w + s->ot();
    double *xh = wc, *rh = wc + s->orr(), *sh = wc + s->os(), *th = wc + s->ot();
    double* step = s->P("step");
    double *Dx = step, *Dr = step + s->orr(), *Ds = step + s->os(), *Dy = step + s->oy(), *Dz = step + s->oz(), *Dt = step + s->ot();
    const double* Dp = step;   // step.primals = first nx+ne+nc entries (point.jl:20)
    double& kappa = s->P("central_path")[0]; double& tau = s->P("fraction_to_boundary")[0];
    double& rho = s->P("penalty")[0]; double* lam = s->P("dual");
    s->trace.clear();
    s->stat_total_iterations = 0; s->stat_outer = 0; s->stat_factorizations = 0; s->stat_refine_fail = 0; s->stat_refine_max = 0; s->stat_lu_fallback = 0;

    if (!o.warmstart) {
        // initialize_slacks! initialize.jl:15-29
        if (call_eval(s, eval, user, w, ORC_EQUALITY | ORC_CONE)) return -3;
        for (i64 i = 0; i < ne; ++i) r[i] = s->P("equality_constraint")[i];
        initialize_cone(s, sl);
        // initialize_duals! initialize.jl:31-36
        for (i64 i = 0; i < ne; ++i) y[i] = 0.0;
        for (i64 i = 0; i < nc; ++i) z[i] = 0.0;
        initialize_cone(s, t);
    }
    kappa = o.central_path_initial; tau = std::max(0.99, 1.0 - kappa);      // initialize.jl:38-42
    rho = o.penalty_initial; for (i64 i = 0; i < ne; ++i) lam[i] = o.dual_initial;   // initialize.jl:44-48

    i64 total_iterations = 1;
    if (call_eval(s, eval, user, w, ORC_OBJECTIVE | ORC_EQUALITY | ORC_EQUALITY_JACOBIAN | ORC_CONE)) return -3;   // :78-83
    double equality_violation = norm_inf(s->P("equality_constraint"), ne);    // :85
    double cone_product_violation = norm_inf(s->P("cone_product"), nc);       // :86 (stale on first use: quirk B-6)
    oracle_cone(s, 0, 0, 0, 1, 0, 1);                                         // :88-91
    oracle_filter_reset(s);                                                   // :95

    for (i64 j = 1; j <= o.max_outer_iterations; ++j) {
        s->stat_outer = j;
        for (i64 i = 1; i <= o.max_residual_iterations; ++i) {
            if (call_eval(s, eval, user, w, ORC_OBJECTIVE_GRADIENT | ORC_EQUALITY_DUAL_GRADIENT | ORC_CONE_DUAL_GRADIENT)) return -3;   // :100-104
            oracle_cone(s, 0, 1, 1, 0, 0, 0);                                 // :106-109
            double M = oracle_merit(s, s->P("objective")[0], r, s->P("barrier")[0]);   // :112-116
            oracle_merit_gradient(s);                                         // :118-124
            oracle_residual(s);                                               // :127
            double residual_violation = norm_p(s->P("residual"), N, o.residual_norm) / (double)N;   // :130
            double optimality_violation = oracle_optimality_error(s);         // :131
            double slack_violation = std::max(norm_inf(s->P("residual") + s->oy(), ne), norm_inf(s->P("residual") + s->oz(), nc));   // :132-135
            if (residual_violation < o.residual_tolerance && slack_violation < o.slack_tolerance &&
                equality_violation <= o.equality_tolerance && cone_product_violation <= o.complementarity_tolerance) {   // :138-143
                if (o.differentiate) { int dr = oracle_differentiate(s, eval, user); if (dr < 0) return dr; }
                s->stat_total_iterations = total_iterations;
                return 1;
            } else if (optimality_violation <= std::max(o.central_path_update_tolerance * kappa, o.optimality_tolerance)) {   // :165
                break;
            }
            double theta = oracle_constraint_violation(s, s->P("equality_constraint"), r, s->P("cone_constraint"), sl);   // :170-172
            uint32_t fl = ORC_OBJECTIVE_HESSIAN | ORC_EQUALITY_JACOBIAN | ORC_CONE_JACOBIAN;
            if (o.constraint_tensor) fl |= ORC_EQUALITY_DUAL_HESSIAN | ORC_CONE_DUAL_HESSIAN;
            if (call_eval(s, eval, user, w, fl)) return -3;                   // :175-181
            oracle_cone(s, 0, 0, 0, 0, 1, 0);                                 // :183-185
            int sd = oracle_search_direction(s);                              // :187
            if (sd < 0) return sd;

            double step_size = 1.0, step_size_t = 1.0;                        // :190-191
            for (i64 k = 0; k < nc; ++k) sh[k] = sl[k] - step_size * Ds[k];
            for (i64 k = 0; k < nc; ++k) th[k] = t[k] - step_size_t * Dt[k];
            i64 cone_iteration = 0;
            while (cone_violation(s, sh, sl, tau)) {                          // :204-211
                step_size = o.scaling_line_search * step_size;
                for (i64 k = 0; k < nc; ++k) sh[k] = sl[k] - step_size * Ds[k];
                cone_iteration += 1;
                if (cone_iteration > o.max_cone_line_search) return -2;
            }
            cone_iteration = 0;
            while (cone_violation(s, th, t, tau)) {                           // :214-221
                step_size_t = o.scaling_line_search * step_size_t;
                for (i64 k = 0; k < nc; ++k) th[k] = t[k] - step_size_t * Dt[k];
                cone_iteration += 1;
                if (cone_iteration > o.max_cone_line_search) return -2;
            }
            for (i64 k = 0; k < nx; ++k) xh[k] = x[k] - step_size * Dx[k];    // :224-229
            for (i64 k = 0; k < ne; ++k) rh[k] = r[k] - step_size * Dr[k];
            if (call_eval(s, eval, user, wc, ORC_OBJECTIVE | ORC_EQUALITY | ORC_CONE)) return -3;   // :231-235
            oracle_cone(s, 1, 1, 1, 0, 0, 0);                                 // :237-240
            double Mh = oracle_merit(s, s->P("objective")[0], rh, s->P("barrier")[0]);
            double thetah = oracle_constraint_violation(s, s->P("equality_constraint"), rh, s->P("cone_constraint"), sh);
            i64 residual_iteration = 0;
            while (residual_iteration < o.max_residual_line_search) {         // :254-302
                if (oracle_check_filter(s, thetah, Mh)) {
                    if (theta <= o.slack_tolerance &&
                        oracle_switching_condition(step_size, Dp, s->P("merit_gradient"), s->n, o.merit_exponent, theta, o.violation_exponent, 1.0) &&
                        oracle_armijo(M, Mh, s->P("merit_gradient"), Dp, s->n, step_size, o.armijo_tolerance, o.machine_tolerance)) {
                        break;
                    } else if (oracle_sufficient_progress(theta, thetah, M, Mh, o.violation_tolerance, o.merit_tolerance, o.machine_tolerance)) {
                        break;
                    }
                }
                step_size = o.scaling_line_search * step_size;
                for (i64 k = 0; k < nx; ++k) xh[k] = x[k] - step_size * Dx[k];
                for (i64 k = 0; k < ne; ++k) rh[k] = r[k] - step_size * Dr[k];
                for (i64 k = 0; k < nc; ++k) sh[k] = sl[k] - step_size * Ds[k];
                if (call_eval(s, eval, user, wc, ORC_OBJECTIVE | ORC_EQUALITY | ORC_CONE)) return -3;
                oracle_cone(s, 1, 1, 1, 0, 0, 0);
                Mh = oracle_merit(s, s->P("objective")[0], rh, s->P("barrier")[0]);
                thetah = oracle_constraint_violation(s, s->P("equality_constraint"), rh, s->P("cone_constraint"), sh);
                residual_iteration += 1;
            }
            // augment_filter!(solver, ...) filter.jl:81-89
            if (!oracle_switching_condition(step_size, Dp, s->P("merit_gradient"), s->n, o.merit_exponent, theta, o.violation_exponent, 1.0) ||
                !oracle_armijo(M, Mh, s->P("merit_gradient"), Dp, s->n, step_size, o.armijo_tolerance, o.machine_tolerance))
                oracle_augment_filter(s, (1.0 - o.violation_tolerance) * theta, M - o.merit_tolerance * theta);
            for (i64 k = 0; k < nx; ++k) x[k] = xh[k];                        // :309-326
            for (i64 k = 0; k < ne; ++k) r[k] = rh[k];
            for (i64 k = 0; k < nc; ++k) sl[k] = sh[k];
            for (i64 k = 0; k < ne; ++k) y[k] = y[k] - step_size * Dy[k];
            for (i64 k = 0; k < nc; ++k) z[k] = z[k] - step_size * Dz[k];
            for (i64 k = 0; k < nc; ++k) t[k] = th[k];
            oracle_cone(s, 0, 0, 0, 1, 0, 0);                                 // :328-330
            equality_violation = norm_inf(s->P("equality_constraint"), ne);   // :332 (values at the accepted candidate)
            cone_product_violation = norm_inf(s->P("cone_product"), nc);      // :333
            total_iterations += 1;
            s->stat_total_iterations = total_iterations;
            if (s->trace.size() < 512) s->trace.push_back(vec(w, w + N));
        }
        kappa = std::max(o.residual_tolerance / 10.0, std::min(o.central_path_scaling * kappa, std::pow(kappa, o.central_path_exponent)));   // :356
        tau = std::max(0.99, 1.0 - kappa);                                    // :359
        for (i64 k = 0; k < ne; ++k) lam[k] = lam[k] + rho * r[k];            // :362-364
        rho = std::min(std::max(o.penalty_scaling * rho, 1.0 / kappa), o.max_penalty);   // :365
        oracle_filter_reset(s);                                               // :368
    }
    s->stat_total_iterations = total_iterations;
    return 0;
}

// differentiate.jl:1-61 + residual_jacobian_parameters.jl:1-40
int oracle_differentiate(oracle_solver* s, oracle_eval_fn eval, void* user) {
    i64 nx = s->nx, ne = s->ne, nc = s->nc, N = s->N, np = s->np;
    if (call_eval(s, eval, user, s->P("solution"),
                  ORC_OBJECTIVE_JACOBIAN_PARAMETERS | ORC_EQUALITY_JACOBIAN_PARAMETERS | ORC_EQUALITY_DUAL_JACOBIAN_PARAMETERS |
                  ORC_CONE_JACOBIAN_PARAMETERS | ORC_CONE_DUAL_JACOBIAN_PARAMETERS)) return -3;
    oracle_residual_jacobian_variables(s);
    oracle_residual_jacobian_variables_symmetric(s);
    oracle_factorize(s, (int)s->opt.update_factorization);
    double* Jp = s->P("jacobian_parameters");
    std::fill(Jp, Jp + N * np, 0.0);
    const double* fxp = s->P("objective_jacobian_variables_parameters");
    const double* gyxp = s->P("equality_dual_jacobian_variables_parameters");
    const double* hzxp = s->P("cone_dual_jacobian_variables_parameters");
    const double* gp = s->P("equality_jacobian_parameters");
    const double* hp = s->P("cone_jacobian_parameters");
    for (i64 i = 0; i < nx; ++i) for (i64 j = 0; j < np; ++j) {
        double v = fxp[i + j * nx]; v += gyxp[i + j * nx]; v += hzxp[i + j * nx];
        Jp[i + j * N] = v;
    }
    for (i64 i = 0; i < ne; ++i) for (i64 j = 0; j < np; ++j) Jp[(s->oy() + i) + j * N] = gp[i + j * ne];
    for (i64 i = 0; i < nc; ++i) for (i64 j = 0; j < np; ++j) Jp[(s->oz() + i) + j * N] = hp[i + j * nc];
    double* Sens = s->P("solution_sensitivity");
    std::fill(Sens, Sens + N * np, 0.0);
    int fact = s->opt.linear_solve_refactor ? 1 : 0;
    for (i64 i = 0; i < np; ++i) {
        double* v = s->P("jacobian_parameters_vector");
        for (i64 k = 0; k < N; ++k) v[k] = Jp[k + i * N];
        search_direction_symmetric(s, s->P("solution_sensitivity_vector"), v, 2, fact, 1);
        for (i64 k = 0; k < N; ++k) Sens[k + i * N] = -1.0 * s->P("solution_sensitivity_vector")[k];
    }
    return 0;
}

// ---- stand-alone QDLDL API (1-based in/out) -------------------------------------------------
void oracle_qdldl_permute_symmetric(int64_t n, const int64_t* Ap, const int64_t* Ai, const double* Ax, const int64_t* iperm,
                                    int64_t* Pp, int64_t* Pi, double* Px, int64_t* AtoPAPt) {
    i64 nnz = Ap[n] - 1;
    ivec ap(n + 1), ai(nnz), ip(n), pp(n + 1), pi(nnz), map(nnz);
    for (i64 i = 0; i <= n; ++i) ap[i] = Ap[i] - 1;
    for (i64 i = 0; i < nnz; ++i) ai[i] = Ai[i] - 1;
    for (i64 i = 0; i < n; ++i) ip[i] = iperm[i] - 1;
    qdldl_permute_symmetric(n, ap.data(), ai.data(), Ax, ip.data(), pp.data(), pi.data(), Px, map.data());
    for (i64 i = 0; i <= n; ++i) Pp[i] = pp[i] + 1;
    for (i64 i = 0; i < nnz; ++i) { Pi[i] = pi[i] + 1; AtoPAPt[i] = map[i] + 1; }
}
int64_t oracle_qdldl_etree(int64_t n, const int64_t* Ap, const int64_t* Ai, int64_t* work, int64_t* Lnz, int64_t* etree) {
    i64 nnz = Ap[n] - 1;
    ivec ap(n + 1), ai(nnz);
    for (i64 i = 0; i <= n; ++i) ap[i] = Ap[i] - 1;
    for (i64 i = 0; i < nnz; ++i) ai[i] = Ai[i] - 1;
    i64 r = qdldl_etree(n, ap.data(), ai.data(), work, Lnz, etree);
    // Julia values: work holds 1-based column ids, etree 1-based parents with -1 = unknown (QDLDL_UNKNOWN)
    for (i64 i = 0; i < n; ++i) { work[i] += 1; if (etree[i] >= 0) etree[i] += 1; }
    return r;
}
int64_t oracle_qdldl_factor(int64_t n, const int64_t* Ap, const int64_t* Ai, const double* Ax, int64_t* Lp, int64_t* Li, double* Lx,
                            double* D, double* Dinv, const int64_t* Lnz, const int64_t* etree) {
    i64 nnz = Ap[n] - 1;
    ivec ap(n + 1), ai(nnz), et(n);
    for (i64 i = 0; i <= n; ++i) ap[i] = Ap[i] - 1;
    for (i64 i = 0; i < nnz; ++i) ai[i] = Ai[i] - 1;
    for (i64 i = 0; i < n; ++i) et[i] = etree[i] >= 1 ? etree[i] - 1 : -1;
    std::vector<char> bw; ivec iw; vec fw;
    i64 r = qdldl_factor(n, ap.data(), ai.data(), Ax, Lp, Li, Lx, D, Dinv, Lnz, et.data(), bw, iw, fw);
    i64 lnz = 0; for (i64 i = 0; i < n; ++i) lnz += Lnz[i];
    for (i64 i = 0; i <= n; ++i) Lp[i] += 1;
    for (i64 i = 0; i < lnz; ++i) Li[i] += 1;
    return r;
}
void oracle_qdldl_solve(int64_t n, const int64_t* Lp, const int64_t* Li, const double* Lx, const double* Dinv, double* b) {
    i64 lnz = Lp[n] - 1;
    ivec lp(n + 1), li(lnz);
    for (i64 i = 0; i <= n; ++i) lp[i] = Lp[i] - 1;
    for (i64 i = 0; i < lnz; ++i) li[i] = Li[i] - 1;
    qdldl_solve(n, lp.data(), li.data(), Lx, Dinv, b);
}

// SplitMix64, u = (next() >> 11) * 2^-53
void oracle_splitmix_uniform(uint64_t problem_id, uint64_t stream_id, double lo, double hi, int64_t count, double* out) {
    uint64_t state = 0xCA11B50000000000ULL + 4096ULL * problem_id + stream_id;
    for (i64 i = 0; i < count; ++i) {
        state += 0x9E3779B97F4A7C15ULL;
        uint64_t zz = state;
        zz = (zz ^ (zz >> 30)) * 0xBF58476D1CE4E5B9ULL;
        zz = (zz ^ (zz >> 27)) * 0x94D049BB133111EBULL;
        zz = zz ^ (zz >> 31);
        double u = (double)(zz >> 11) * (1.0 / 9007199254740992.0);
        out[i] = lo + (hi - lo) * u;
    }
}

}  // extern "C"
