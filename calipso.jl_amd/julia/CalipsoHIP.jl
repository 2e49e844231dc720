# CalipsoHIP.jl — Julia host layer over libcalipso_hip.so (include/calipso_hip.h).
#
# Keeps the reference's surface for the Newton/KKT hot path (CALIPSO.jl v0.1.1):
#     Solver(...)  ->  HIPSolver(solver::CALIPSO.Solver)      wraps an ordinary CALIPSO.Solver (its codegen'd methods,
#                                                            ProblemData, Indices, Options are reused unchanged)
#     initialize!(hs, x0)            src/solver/initialize.jl:9-13
#     solve!(hs)::Bool               src/solver/solve.jl:8-377   — the whole loop runs in the library; Julia only evaluates the
#                                                                  user functions (evaluate!) when the library calls back
#     HIPLDLSolver <: LinearSolver   src/solver/linear_solver.jl:1-60 seam: factorize!(s, A), compute_inertia!(s), linear_solve!(s, x, A, b) on
#                                    the device for the sparse K the reference assembles itself (calipso_hip_ldl_*)
#
# NOTE: Julia is not installed in the build container of this repository, so this file has not been executed there; it is the
# binding a maintainer adds (see INTEGRATION.md).  The same C ABI is exercised end-to-end by the Python ctypes mirror
# (calipso.jl_amd/__init__.py) in tests/.
module CalipsoHIP

using CALIPSO
using LinearAlgebra
using SparseArrays

const lib = get(ENV, "CALIPSO_HIP_LIB", joinpath(@__DIR__, "..", "libcalipso_hip.so"))

# evaluate! flags (include/calipso_hip.h)
const EVAL_OBJECTIVE = UInt32(1) << 0
const EVAL_OBJECTIVE_GRADIENT = UInt32(1) << 1
const EVAL_OBJECTIVE_HESSIAN = UInt32(1) << 2
const EVAL_EQUALITY = UInt32(1) << 3
const EVAL_EQUALITY_JACOBIAN = UInt32(1) << 4
const EVAL_EQUALITY_DUAL_GRADIENT = UInt32(1) << 5
const EVAL_EQUALITY_DUAL_HESSIAN = UInt32(1) << 6
const EVAL_CONE = UInt32(1) << 7
const EVAL_CONE_JACOBIAN = UInt32(1) << 8
const EVAL_CONE_DUAL_GRADIENT = UInt32(1) << 9
const EVAL_CONE_DUAL_HESSIAN = UInt32(1) << 10
const EVAL_OBJECTIVE_JACOBIAN_PARAMETERS = UInt32(1) << 11
const EVAL_EQUALITY_JACOBIAN_PARAMETERS = UInt32(1) << 12
const EVAL_EQUALITY_DUAL_JACOBIAN_PARAMETERS = UInt32(1) << 13
const EVAL_CONE_JACOBIAN_PARAMETERS = UInt32(1) << 14
const EVAL_CONE_DUAL_JACOBIAN_PARAMETERS = UInt32(1) << 15

struct HIPError <: Exception
    code::Int32
    msg::String
end

mutable struct HIPSolver
    handle::Ptr{Cvoid}
    solver::CALIPSO.Solver          # the reference Solver: methods, problem (ProblemData), indices, options, parameters
    eval_cfunction::Base.CFunction  # keeps the @cfunction alive
end

last_error(h) = unsafe_string(ccall((:calipso_hip_last_error, lib), Cstring, (Ptr{Cvoid},), h))

function check(h, rc::Integer, what)
    rc == -1 && error("inertia correction failure")    # same text as src/solver/inertia.jl:72
    rc == -2 && error("cone search failure")           # src/solver/solve.jl:210,220
    rc < 0 && throw(HIPError(Int32(rc), "$what: $(last_error(h))"))
    rc == 1 && @warn "Zero entry in D (matrix is not quasidefinite)"     # src/solver/qdldl.jl:309-311
    rc == 2 && @warn "iterative refinement failure"                       # src/solver/iterative_refinement.jl:50
    return rc
end

set_field!(h, name::String, v::AbstractArray{Float64}) =
    check(h, ccall((:calipso_hip_set_field, lib), Int32, (Ptr{Cvoid}, Cstring, Ptr{Float64}, Int64), h, name, v, length(v)), "set_field($name)")
set_field!(h, name::String, v::Real) = set_field!(h, name, [Float64(v)])
function get_field(h, name::String, len::Integer)
    out = zeros(len)
    check(h, ccall((:calipso_hip_get_field, lib), Int32, (Ptr{Cvoid}, Cstring, Ptr{Float64}, Int64), h, name, out, len), "get_field($name)")
    return out
end

"""
    HIPSolver(solver::CALIPSO.Solver; device=0)

Create the device handle for an existing reference `Solver` (src/solver/solver.jl:46-150).  Index sets are passed 1-based,
exactly as `solver.indices` holds them.
"""
# structure = (row_first, row_last, hessian_block_start) (1-based Int vectors) makes a STRUCTURED handle (calipso_hip_create_structured): only the stage
# blocks of the problem live on the device.  `declared_structure(solver)` reads it off the methods' sparsity lists.
function HIPSolver(solver::CALIPSO.Solver; device::Integer=0, structure=nothing)
    d = solver.dimensions
    idx = solver.indices
    nn = Vector{Int64}(idx.cone_nonnegative)
    soc = idx.cone_second_order
    ptr = Int64[0]
    flat = Int64[]
    for c in soc
        append!(flat, c)
        push!(ptr, length(flat))
    end
    href = Ref{Ptr{Cvoid}}(C_NULL)
    rc = if structure === nothing
        ccall((:calipso_hip_create, lib), Int32,
            (Int64, Int64, Int64, Int64, Int64, Ptr{Int64}, Int64, Ptr{Int64}, Ptr{Int64}, Int32, Ptr{Ptr{Cvoid}}),
            d.variables, d.parameters, d.equality_dual, d.cone_dual, length(nn), nn, length(soc), ptr, flat, device, href)
    else
        rf, rl, hb = Vector{Int64}(structure[1]), Vector{Int64}(structure[2]), Vector{Int64}(structure[3])
        ccall((:calipso_hip_create_structured, lib), Int32,
            (Int64, Int64, Int64, Int64, Int64, Ptr{Int64}, Int64, Ptr{Int64}, Ptr{Int64}, Int32, Ptr{Int64}, Ptr{Int64}, Int64, Ptr{Int64}, Ptr{Ptr{Cvoid}}),
            d.variables, d.parameters, d.equality_dual, d.cone_dual, length(nn), nn, length(soc), ptr, flat, device, rf, rl, length(hb), hb, href)
    end
    if rc != 0 && structure !== nothing
        # the declared structure has nothing for the block path to exploit (one Hessian block: e.g. dynamics whose y'f Hessian couples x_t with x_{t+1}, src/
        # trajectory_optimization/dynamics.jl:245-259): a dense handle instead, whose banded / stage-parallel treatment calipso_hip_analyze_structure can still turn on
        @warn "calipso_hip_create_structured refused the declared structure ($(last_error(C_NULL))); falling back to a dense handle"
        rc = ccall((:calipso_hip_create, lib), Int32,
            (Int64, Int64, Int64, Int64, Int64, Ptr{Int64}, Int64, Ptr{Int64}, Ptr{Int64}, Int32, Ptr{Ptr{Cvoid}}),
            d.variables, d.parameters, d.equality_dual, d.cone_dual, length(nn), nn, length(soc), ptr, flat, device, href)
    end
    rc != 0 && throw(HIPError(rc, "calipso_hip_create: $(last_error(href[]))"))
    hs = HIPSolver(href[], solver, @cfunction($(evaluate_callback), Int32, (Ptr{Cvoid}, UInt32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64})))
    finalizer(x -> ccall((:calipso_hip_destroy, lib), Int32, (Ptr{Cvoid},), x.handle), hs)
    # options (src/solver/options.jl:6-59) -> handle
    for f in fieldnames(typeof(solver.options))
        v = getfield(solver.options, f)
        v isa Real && f != :linear_solver && ccall((:calipso_hip_set_field, lib), Int32, (Ptr{Cvoid}, Cstring, Ptr{Float64}, Int64), hs.handle, "opt.$f", [Float64(v)], 1)
    end
    length(solver.parameters) > 0 && set_field!(hs.handle, "parameters", solver.parameters)
    return hs
end

# The structure of a solver whose methods carry sparsity lists (src/solver/methods.jl; every trajectory problem): per row of [equality; cone] the extrema of
# the columns in the Jacobian lists, and the diagonal blocks of the three Hessian lists (a new block starts where no listed entry couples to an earlier column).
function declared_structure(solver::CALIPSO.Solver)
    d = solver.dimensions; m = solver.methods
    ne, nc, nx = d.equality_dual, d.cone_dual, d.variables
    rf = fill(1, ne + nc); rl = fill(0, ne + nc); seen = falses(ne + nc)
    for (off, list) in ((0, m.equality_jacobian_variables_sparsity), (ne, m.cone_jacobian_variables_sparsity)), (i, j) in list
        k = off + i
        rf[k] = seen[k] ? min(rf[k], j) : j; rl[k] = seen[k] ? max(rl[k], j) : j; seen[k] = true
    end
    reach = collect(1:nx)
    for list in (m.objective_jacobian_variables_variables_sparsity, m.equality_dual_jacobian_variables_variables_sparsity, m.cone_dual_jacobian_variables_variables_sparsity), (i, j) in list
        lo, hi = minmax(i, j); reach[lo] = max(reach[lo], hi)
    end
    starts = Int64[1]; r = 0
    for j in 1:nx
        j > starts[end] && r < j && push!(starts, j)
        r = max(r, reach[j])
    end
    return (rf, rl, starts)
end

const ACTIVE = Ref{Union{Nothing,HIPSolver}}(nothing)   # fallback for callers that pass user = C_NULL to the C drivers

# evaluate!(problem, methods, idx, point, parameters; <flags>)  (src/solver/evaluate.jl:1-124) at the point the library hands
# over, then upload of the flagged ProblemData fields (the three Hessian terms as one summed "lagrangian_hessian",
# src/solver/residual_jacobian_variables.jl:10-16).
function evaluate_callback(user::Ptr{Cvoid}, flags::UInt32, px::Ptr{Float64}, py::Ptr{Float64}, pz::Ptr{Float64}, pth::Ptr{Float64})::Int32
    # No exception may unwind through the @cfunction frame into C: everything, including the lookup of the solver, is inside `try`.
    try
        # `user` = pointer_from_objref(hs) as passed to calipso_hip_solve / calipso_hip_group_set_evaluators (the caller keeps `hs`
        # rooted with GC.@preserve for the duration of the ccall); ACTIVE[] serves callers that passed C_NULL
        hs = user != C_NULL ? (unsafe_pointer_to_objref(user)::HIPSolver) : (ACTIVE[]::HIPSolver)
        s = hs.solver
        d = s.dimensions
        pt = s.candidate                      # scratch Point the generated functions read from
        pt.variables .= unsafe_wrap(Array, px, d.variables)
        d.equality_dual > 0 && (pt.equality_dual .= unsafe_wrap(Array, py, d.equality_dual))
        d.cone_dual > 0 && (pt.cone_dual .= unsafe_wrap(Array, pz, d.cone_dual))
        has(f) = (flags & f) != 0
        CALIPSO.evaluate!(s.problem, s.methods, s.indices, pt, s.parameters;
            objective=has(EVAL_OBJECTIVE), objective_gradient_variables=has(EVAL_OBJECTIVE_GRADIENT),
            objective_jacobian_variables_variables=has(EVAL_OBJECTIVE_HESSIAN),
            equality_constraint=has(EVAL_EQUALITY), equality_jacobian_variables=has(EVAL_EQUALITY_JACOBIAN),
            equality_dual_jacobian_variables=has(EVAL_EQUALITY_DUAL_GRADIENT),
            equality_dual_jacobian_variables_variables=has(EVAL_EQUALITY_DUAL_HESSIAN),
            cone_constraint=has(EVAL_CONE), cone_jacobian_variables=has(EVAL_CONE_JACOBIAN),
            cone_dual_jacobian_variables=has(EVAL_CONE_DUAL_GRADIENT),
            cone_dual_jacobian_variables_variables=has(EVAL_CONE_DUAL_HESSIAN),
            objective_jacobian_variables_parameters=has(EVAL_OBJECTIVE_JACOBIAN_PARAMETERS),
            equality_jacobian_parameters=has(EVAL_EQUALITY_JACOBIAN_PARAMETERS),
            equality_dual_jacobian_variables_parameters=has(EVAL_EQUALITY_DUAL_JACOBIAN_PARAMETERS),
            cone_jacobian_parameters=has(EVAL_CONE_JACOBIAN_PARAMETERS),
            cone_dual_jacobian_variables_parameters=has(EVAL_CONE_DUAL_JACOBIAN_PARAMETERS))
        p = s.problem
        h = hs.handle
        has(EVAL_OBJECTIVE) && set_field!(h, "objective", p.objective)
        has(EVAL_OBJECTIVE_GRADIENT) && set_field!(h, "objective_gradient_variables", p.objective_gradient_variables)
        has(EVAL_EQUALITY) && d.equality_dual > 0 && set_field!(h, "equality_constraint", p.equality_constraint)
        has(EVAL_EQUALITY_JACOBIAN) && d.equality_dual > 0 && set_field!(h, "equality_jacobian_variables", p.equality_jacobian_variables)
        has(EVAL_EQUALITY_DUAL_GRADIENT) && set_field!(h, "equality_dual_jacobian_variables", p.equality_dual_jacobian_variables)
        has(EVAL_CONE) && d.cone_dual > 0 && set_field!(h, "cone_constraint", p.cone_constraint)
        has(EVAL_CONE_JACOBIAN) && d.cone_dual > 0 && set_field!(h, "cone_jacobian_variables", p.cone_jacobian_variables)
        has(EVAL_CONE_DUAL_GRADIENT) && set_field!(h, "cone_dual_jacobian_variables", p.cone_dual_jacobian_variables)
        if has(EVAL_OBJECTIVE_HESSIAN)
            L = copy(p.objective_jacobian_variables_variables)
            if s.options.constraint_tensor
                L .+= p.equality_dual_jacobian_variables_variables
                L .+= p.cone_dual_jacobian_variables_variables
            end
            set_field!(h, "lagrangian_hessian", L)
        end
        if has(EVAL_OBJECTIVE_JACOBIAN_PARAMETERS) && d.parameters > 0
            G = p.objective_jacobian_variables_parameters + p.equality_dual_jacobian_variables_parameters + p.cone_dual_jacobian_variables_parameters
            set_field!(h, "lagrangian_gradient_parameters", G)
            d.equality_dual > 0 && set_field!(h, "equality_jacobian_parameters", p.equality_jacobian_parameters)
            d.cone_dual > 0 && set_field!(h, "cone_jacobian_parameters", p.cone_jacobian_parameters)
        end
        return Int32(0)
    catch err
        @error "evaluate! callback failed" err
        return Int32(1)
    end
end

"initialize!(solver, guess)  src/solver/initialize.jl:9-13"
function CALIPSO.initialize!(hs::HIPSolver, guess)
    g = Vector{Float64}(guess)
    hs.solver.solution.variables .= g
    check(hs.handle, ccall((:calipso_hip_initialize, lib), Int32, (Ptr{Cvoid}, Ptr{Float64}), hs.handle, g), "initialize!")
    return
end

"""device-side evaluation (src/solver/evaluate.jl:37-121 as kernels on the solver's stream): `fn` = address of a `calipso_device_eval_fn` (dense ProblemData layout) or,
with `blocks = true` on a handle made with `structure = ...`, of a `calipso_device_block_eval_fn` that writes the handle's packed blocks (no dense scratch);
`user` is handed to it unchanged.  solve! then never calls back into Julia for an evaluation."""
function set_device_evaluator!(hs::HIPSolver, fn::Ptr{Cvoid}, user::Ptr{Cvoid} = C_NULL; blocks::Bool = false)
    rc = blocks ? ccall((:calipso_hip_set_device_block_evaluator, lib), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), hs.handle, fn, user) :
                  ccall((:calipso_hip_set_device_evaluator, lib), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), hs.handle, fn, user)
    check(hs.handle, rc, "set_device_evaluator!")
    return
end

"solve!(solver)::Bool  src/solver/solve.jl:8-377 — results are copied back into the wrapped Solver's fields"
function CALIPSO.solve!(hs::HIPSolver)
    rc = GC.@preserve hs ccall((:calipso_hip_solve, lib), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), hs.handle, hs.eval_cfunction, pointer_from_objref(hs))
    check(hs.handle, rc, "solve!")
    copy_back!(hs)
    return rc == 1
end

"copy the results of the device handle into the wrapped reference Solver's own fields (what tests and examples read)"
function copy_back!(hs::HIPSolver)
    s = hs.solver
    d = s.dimensions
    s.solution.all .= get_field(hs.handle, "solution", d.total)
    s.data.residual.all .= get_field(hs.handle, "residual", d.total)
    s.data.step.all .= get_field(hs.handle, "step", d.total)
    d.cone_dual > 0 && (s.problem.cone_product .= get_field(hs.handle, "cone_product", d.cone_dual))
    for (name, ref) in (("central_path", s.central_path), ("penalty", s.penalty), ("fraction_to_boundary", s.fraction_to_boundary),
                        ("primal_regularization", s.primal_regularization), ("dual_regularization", s.dual_regularization),
                        ("primal_regularization_last", s.primal_regularization_last))
        ref[1] = get_field(hs.handle, name, 1)[1]
    end
    d.equality_dual > 0 && (s.dual .= get_field(hs.handle, "dual", d.equality_dual))
    if s.options.differentiate && d.parameters > 0
        s.data.solution_sensitivity .= reshape(get_field(hs.handle, "solution_sensitivity", d.total * d.parameters), d.total, d.parameters)
    end
    return
end

# ---- linear-solver seam (src/solver/linear_solver.jl:1-60) -------------------------------------------------------------------
"""
    HIPLDLSolver(A::SparseMatrixCSC) / hip_ldl_solver(A)

Drop-in for `LDLSolver` (linear_solver.jl:3-17,46-50): a device LDL^T for the sparse symmetric quasi-definite matrix the
reference assembles itself.  `factorize!(s, A)` ships A's CSC arrays (colptr / rowval / nzval, 1-based as Julia stores them; only
triu(A) is read, linear_solver.jl:23) to the device and factors there; `compute_inertia!` / `linear_solve!` have the reference's
signatures and semantics, so the reference's own `search_direction!` (search_direction.jl:1-23), `iterative_refinement!`
(iterative_refinement.jl:20-26), `inertia_correction!` (inertia.jl:17-28) and `differentiate!` (differentiate.jl:19-46) run
unmodified with `solver.linear_solver = hip_ldl_solver(solver.data.jacobian_variables_symmetric)`.
(One-line change in the reference for that assignment: the field is declared `linear_solver::LDLSolver{T,Int}` in
solver.jl:14 — widen it to `linear_solver::LinearSolver`.)
"""
mutable struct HIPLDLSolver <: CALIPSO.LinearSolver
    handle::Ptr{Cvoid}
    n::Int
    inertia::CALIPSO.Inertia
end

function HIPLDLSolver(n::Integer; device::Integer=0)
    href = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:calipso_hip_ldl_create, lib), Int32, (Int64, Int32, Ptr{Ptr{Cvoid}}), n, device, href)
    rc != 0 && throw(HIPError(rc, "calipso_hip_ldl_create: $(last_error(href[]))"))
    s = HIPLDLSolver(href[], n, CALIPSO.Inertia(0, 0, 0))
    finalizer(x -> ccall((:calipso_hip_destroy, lib), Int32, (Ptr{Cvoid},), x.handle), s)
    return s
end

"ldl_solver(A)  linear_solver.jl:46-50 (factors once at construction, as `qdldl(A)` does there)"
function hip_ldl_solver(A::SparseMatrixCSC{Float64,Int}; device::Integer=0)
    s = HIPLDLSolver(size(A, 1); device=device)
    CALIPSO.factorize!(s, A)
    return s
end
hip_ldl_solver(A::Matrix{Float64}; kw...) = hip_ldl_solver(sparse(A); kw...)

"factorize!(s, A; update)  linear_solver.jl:19-31.  `update` only selects between value update and symbolic re-analysis in QDLDL; the dense device factorisation has no symbolic phase, so both take the same path.  Like the reference with update=true, only triu(A) matters (A itself is not mutated here)."
function CALIPSO.factorize!(s::HIPLDLSolver, A::SparseMatrixCSC{Float64,Int}; update=false)
    out = zeros(Int64, 3)
    rc = ccall((:calipso_hip_ldl_factorize_csc, lib), Int32, (Ptr{Cvoid}, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Float64}, Ptr{Int64}),
               s.handle, s.n, A.colptr, A.rowval, A.nzval, out)
    rc < 0 && throw(HIPError(rc, "factorize!: $(last_error(s.handle))"))
    rc == 1 && @warn "Zero entry in D (matrix is not quasidefinite)"     # qdldl.jl:309-311
    s.inertia.positive, s.inertia.negative, s.inertia.zero = out
    return nothing
end

"compute_inertia!(s)  linear_solver.jl:33-44"
function CALIPSO.compute_inertia!(s::HIPLDLSolver)
    out = zeros(Int64, 3)
    check(s.handle, ccall((:calipso_hip_ldl_inertia, lib), Int32, (Ptr{Cvoid}, Ptr{Int64}), s.handle, out), "compute_inertia!")
    s.inertia.positive, s.inertia.negative, s.inertia.zero = out
    return nothing
end

"linear_solve!(s, x, A, b; fact, update)  linear_solver.jl:52-60"
function CALIPSO.linear_solve!(s::HIPLDLSolver, x::Vector{Float64}, A::SparseMatrixCSC{Float64,Int}, b::Vector{Float64}; fact=true, update=true)
    fact && CALIPSO.factorize!(s, A; update=update)
    check(s.handle, ccall((:calipso_hip_ldl_solve, lib), Int32, (Ptr{Cvoid}, Int64, Int64, Ptr{Float64}, Ptr{Float64}), s.handle, s.n, 1, b, x), "linear_solve!")
    return
end

"linear_solve!(s, X, A, B; fact, update) for matrices  linear_solver.jl:82-99: all columns through one factorisation"
function CALIPSO.linear_solve!(s::HIPLDLSolver, x::Matrix{Float64}, A::SparseMatrixCSC{Float64,Int}, b::Matrix{Float64}; fact=true, update=true)
    fact && CALIPSO.factorize!(s, A; update=update)
    check(s.handle, ccall((:calipso_hip_ldl_solve, lib), Int32, (Ptr{Cvoid}, Int64, Int64, Ptr{Float64}, Ptr{Float64}), s.handle, s.n, size(b, 2), b, x), "linear_solve!")
    return
end

# ---- sparse variant: qdldl(A; perm) on the device without dense storage (csrc/sparse.hip) ---------------------------------------------------
"""
    HIPSparseLDLSolver(A::SparseMatrixCSC; method=:nested_dissection, perm=nothing) / hip_sparse_ldl_solver(A)

`LDLSolver` (linear_solver.jl:1-60) with the analyse phase of `qdldl(A; perm)` (qdldl.jl:134-188) at construction and the numeric
factorisation / triangular solves level-scheduled on the device; memory O(nnz(L)).  The pattern of `A` is fixed at construction
(as `update=true` assumes in linear_solver.jl:24-27); `factorize!` sends only `A.nzval`.
"""
mutable struct HIPSparseLDLSolver <: CALIPSO.LinearSolver
    handle::Ptr{Cvoid}
    n::Int
    nnz::Int
    inertia::CALIPSO.Inertia
end
const SPARSE_METHODS = Dict(:natural => 0, :rcm => 1, :minimum_degree => 2, :nested_dissection => 4, :nested_dissection_columns => 5)
sparse_last_error(h) = unsafe_string(ccall((:calipso_hip_sparse_last_error, lib), Cstring, (Ptr{Cvoid},), h))

function HIPSparseLDLSolver(A::SparseMatrixCSC{Float64,Int}; method::Symbol=:nested_dissection, perm::Union{Nothing,Vector{Int}}=nothing, device::Integer=0)
    href = Ref{Ptr{Cvoid}}(C_NULL)
    m = perm === nothing ? SPARSE_METHODS[method] : 3
    rc = ccall((:calipso_hip_sparse_create, lib), Int32, (Int64, Ptr{Int64}, Ptr{Int64}, Int32, Ptr{Int64}, Int32, Ptr{Ptr{Cvoid}}),
               size(A, 1), A.colptr, A.rowval, m, perm === nothing ? C_NULL : perm, device, href)
    if rc != 0
        msg = sparse_last_error(href[])
        href[] != C_NULL && ccall((:calipso_hip_sparse_destroy, lib), Int32, (Ptr{Cvoid},), href[])
        throw(HIPError(rc, "calipso_hip_sparse_create: $msg"))
    end
    s = HIPSparseLDLSolver(href[], size(A, 1), nnz(A), CALIPSO.Inertia(0, 0, 0))
    finalizer(x -> ccall((:calipso_hip_sparse_destroy, lib), Int32, (Ptr{Cvoid},), x.handle), s)
    return s
end
function hip_sparse_ldl_solver(A::SparseMatrixCSC{Float64,Int}; kw...)
    s = HIPSparseLDLSolver(A; kw...)
    CALIPSO.factorize!(s, A)
    return s
end

function CALIPSO.factorize!(s::HIPSparseLDLSolver, A::SparseMatrixCSC{Float64,Int}; update=false)
    nnz(A) == s.nnz || throw(HIPError(-1, "factorize!: the pattern of A differs from the analysed one (build a new HIPSparseLDLSolver)"))
    out = zeros(Int64, 3)
    rc = ccall((:calipso_hip_sparse_factorize, lib), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Int64}), s.handle, A.nzval, out)
    rc < 0 && throw(HIPError(rc, "factorize!: $(sparse_last_error(s.handle))"))
    rc == 1 && @warn "Zero entry in D (matrix is not quasidefinite)"     # qdldl.jl:309-311
    s.inertia.positive, s.inertia.negative, s.inertia.zero = out
    return nothing
end
CALIPSO.compute_inertia!(s::HIPSparseLDLSolver) = nothing      # (set by factorize!, as the values of linear_solver.jl:33-44 only change there)
function CALIPSO.linear_solve!(s::HIPSparseLDLSolver, x::VecOrMat{Float64}, A::SparseMatrixCSC{Float64,Int}, b::VecOrMat{Float64}; fact=true, update=true)
    fact && CALIPSO.factorize!(s, A; update=update)
    rc = ccall((:calipso_hip_sparse_solve, lib), Int32, (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}), s.handle, size(b, 2), b, x)
    rc < 0 && throw(HIPError(rc, "linear_solve!: $(sparse_last_error(s.handle))"))
    return
end

# The handle-resident variant: the KKT blocks are already on the device (uploaded by the evaluation callback or by set_field!), the
# condensed matrix is never formed on the host.  factorize! = calipso_hip_factorize on those blocks with the handle's kappa / rho /
# regularisation scalars (which the caller syncs with sync_scalars!); `A` is accepted for signature parity only.
mutable struct HIPKKTSolver <: CALIPSO.LinearSolver
    hs::HIPSolver
    inertia::CALIPSO.Inertia
end
HIPKKTSolver(hs::HIPSolver) = HIPKKTSolver(hs, CALIPSO.Inertia(0, 0, 0))

"push the reference Solver's iterate and scalars (solver.jl:81-127) to the handle before factorize!/linear_solve! in HIPKKTSolver mode"
function sync_scalars!(hs::HIPSolver)
    s = hs.solver
    set_field!(hs.handle, "solution", s.solution.all)
    for (name, ref) in (("central_path", s.central_path), ("penalty", s.penalty), ("fraction_to_boundary", s.fraction_to_boundary),
                        ("primal_regularization", s.primal_regularization), ("dual_regularization", s.dual_regularization))
        set_field!(hs.handle, name, ref[1])
    end
    return
end

function CALIPSO.factorize!(s::HIPKKTSolver, A=nothing; update=true)
    sync_scalars!(s.hs)
    out = zeros(Int64, 3)
    check(s.hs.handle, ccall((:calipso_hip_factorize, lib), Int32, (Ptr{Cvoid}, Ptr{Int64}), s.hs.handle, out), "factorize!")
    s.inertia.positive, s.inertia.negative, s.inertia.zero = out
    return nothing
end
CALIPSO.compute_inertia!(s::HIPKKTSolver) = nothing      # filled by factorize! (one device pass counts the pivot signs)

function CALIPSO.linear_solve!(s::HIPKKTSolver, x::Vector{Float64}, A, b::Vector{Float64}; fact=true, update=true)
    fact && CALIPSO.factorize!(s, A; update=update)
    set_field!(s.hs.handle, "residual_symmetric", b)
    check(s.hs.handle, ccall((:calipso_hip_linear_solve, lib), Int32, (Ptr{Cvoid},), s.hs.handle), "linear_solve!")
    x .= get_field(s.hs.handle, "step_symmetric", length(b))
    return
end

# ---- search_direction_nonsymmetric! (src/solver/search_direction.jl:106-119): step = H \ residual on the device ---------------
function search_direction_nonsymmetric!(hs::HIPSolver)
    check(hs.handle, ccall((:calipso_hip_search_direction_nonsymmetric, lib), Int32, (Ptr{Cvoid},), hs.handle), "search_direction_nonsymmetric!")
    return
end

# ---- stage-banded structure: what the reference gets from its sparsity pattern (src/trajectory_optimization/sparsity.jl) ---------
"Analyse the non-zero pattern of the blocks currently on the device; afterwards the factorisation and the solves skip everything outside the band of the Schur complement. Returns (half_bandwidth, band_blocks, equality_rows_per_group, cone_rows_per_group)."
function analyze_structure!(hs::HIPSolver)
    out = zeros(Int64, 4)
    check(hs.handle, ccall((:calipso_hip_analyze_structure, lib), Int32, (Ptr{Cvoid}, Ptr{Int64}), hs.handle, out), "analyze_structure!")
    return Tuple(out)
end
clear_structure!(hs::HIPSolver) = check(hs.handle, ccall((:calipso_hip_clear_structure, lib), Int32, (Ptr{Cvoid},), hs.handle), "clear_structure!")
# calipso_hip_set_stage_blocks (after analyze_structure!): packed blocks of [gx; hx] / the Lagrangian Hessian, block mat-vecs, Schur complement by
# segment pairs (csrc/blocks.hip).  Returns (z_blocks, hessian_blocks, segments, packed_doubles).
function set_stage_blocks!(hs::HIPSolver, on::Bool=true)
    out = zeros(Int64, 4)
    check(hs.handle, ccall((:calipso_hip_set_stage_blocks, lib), Int32, (Ptr{Cvoid}, Int32, Ptr{Int64}), hs.handle, on ? 1 : 0, out), "set_stage_blocks!")
    return (z_blocks = out[1], hessian_blocks = out[2], segments = out[3], packed_doubles = out[4])
end

# calipso_hip_kernel_times: [1] ms of the panel-step launches of the last LDL^T of S, [2] their number, [3] NP, [4] bytes of the device slab
function kernel_times(hs::HIPSolver)
    out = zeros(Float64, 8)
    ccall((:calipso_hip_kernel_times, lib), Int32, (Ptr{Cvoid}, Ptr{Float64}), hs.handle, out)
    return out
end

"After `analyze_structure!`: factor the Schur complement by the multifrontal sparse LDL' over a nested dissection of its pattern (log2(stages) launches instead of the chain of nx pivots); `batch` >= the largest group this solver leads. Returns (tree levels, largest front, nnz of the pattern, 2)."
function set_stage_parallel!(hs::HIPSolver, on::Bool=true; batch::Integer=1)
    out = zeros(Int64, 4)
    check(hs.handle, ccall((:calipso_hip_set_stage_parallel, lib), Int32, (Ptr{Cvoid}, Int32, Int32, Ptr{Int64}), hs.handle, on ? 1 : 0, batch, out), "set_stage_parallel!")
    return Tuple(out)
end

# ---- groups: many same-shape solvers stepped through the same kernel launches (BASELINE config C4) ------------------------------
"Up to 16 `HIPSolver`s of one shape on one device; `newton_step!` advances every member by one inner iteration of solve!."
mutable struct HIPGroup
    handle::Ptr{Cvoid}
    members::Vector{HIPSolver}
end
function HIPGroup(members::Vector{HIPSolver})
    hs = Ptr{Cvoid}[m.handle for m in members]
    g = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:calipso_hip_group_create, lib), Int32, (Ptr{Ptr{Cvoid}}, Int32, Ptr{Ptr{Cvoid}}), hs, length(hs), g)
    rc == 0 || error("calipso_hip_group_create failed ($rc)")
    grp = HIPGroup(g[], members)
    finalizer(x -> ccall((:calipso_hip_group_destroy, lib), Int32, (Ptr{Cvoid},), x.handle), grp)
    return grp
end
"Do the HIP streams of two handles run side by side?  (calipso_hip_streams_concurrent: measured — a long kernel on one, a short one on the other, both ways)"
function streams_concurrent(a::HIPSolver, b::HIPSolver)
    out = zeros(Float64, 6)
    rc = ccall((:calipso_hip_streams_concurrent, lib), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Float64}), a.handle, b.handle, out)
    rc == 0 || error("calipso_hip_streams_concurrent failed ($rc)")
    return out[1] == 1.0, out[2], out[3], out[4], out[5], out[6]
end
"A new HIP stream (another hardware queue) for the handle: calipso_hip_rebind_stream; priority_class 0, 1, 2 or -1 (keep)."
function rebind_stream!(s::HIPSolver, priority_class::Integer=-1)
    rc = ccall((:calipso_hip_rebind_stream, lib), Int32, (Ptr{Cvoid}, Int32), s.handle, priority_class)
    rc == 0 || error("calipso_hip_rebind_stream failed ($rc)")
    return s
end
"""
Handles (or the first members of groups) that are stepped at the same time from different tasks: probe every pair and give the later one of a colliding pair a new
stream until all run side by side (what `BatchSolver.spread_streams` of the Python mirror does at creation).  Returns the number of streams replaced.
"""
function spread_streams!(leaders::Vector{HIPSolver}; max_rebinds::Integer=12)
    rebinds = 0
    for attempt in 0:max_rebinds
        bad = [(i, j) for j in 1:length(leaders) for i in 1:j-1 if !streams_concurrent(leaders[i], leaders[j])[1]]
        (isempty(bad) || attempt == max_rebinds) && break
        rebind_stream!(leaders[bad[1][2]], (rebinds + bad[1][2]) % 3)
        rebinds += 1
    end
    return rebinds
end
"One inner Newton iteration (solve.jl:98-353) for every member (device evaluator attached); returns (info 6 x B, status B)."
function newton_step!(g::HIPGroup; advance::Bool=true)
    B = length(g.members)
    info = zeros(Float64, 6, B); status = zeros(Int32, B)
    rc = ccall((:calipso_hip_group_newton_step, lib), Int32, (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Int32}), g.handle, advance ? 1 : 0, info, status)
    rc < 0 && error("calipso_hip_group_newton_step failed ($rc)")
    return info, status
end

"solve! (solve.jl:8-377) of every member in lockstep; members are evaluated through `evaluate_callback` (their own CALIPSO.evaluate!), which finds its solver through the per-member `user` pointer.  Returns the per-member results (1 converged, 0 caps reached, < 0 error code)."
function CALIPSO.solve!(g::HIPGroup)
    members = g.members
    res = zeros(Int32, length(members))
    cb = @cfunction(evaluate_callback, Int32, (Ptr{Cvoid}, UInt32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}))
    evals = fill(cb, length(members))
    GC.@preserve members begin
        users = Ptr{Cvoid}[pointer_from_objref(m) for m in members]
        rc = ccall((:calipso_hip_group_set_evaluators, lib), Int32, (Ptr{Cvoid}, Ptr{Ptr{Cvoid}}, Ptr{Ptr{Cvoid}}), g.handle, evals, users)
        rc < 0 && error("calipso_hip_group_set_evaluators failed ($rc)")
        rc = ccall((:calipso_hip_group_solve, lib), Int32, (Ptr{Cvoid}, Ptr{Int32}), g.handle, res)
        rc < 0 && error("calipso_hip_group_solve failed ($rc): $(last_error(members[1].handle))")
    end
    for (m, r) in zip(members, res)
        r >= 0 && copy_back!(m)
    end
    return res
end

# ---- solve! for a batch of small QPs in one kernel launch (include/calipso_hip.h: calipso_hip_smallnewton_*; csrc/smallnewton.hip) ----------------
"`batch` independent QPs (min c x'Px + q'x s.t. Ax = b, h - Gx >= 0) of one shape: `HIPSmallNewton(nx, ne, nc, batch)`, `set_qp!`, `initialize!`, `solve!` — every instance's whole solve! in ONE launch."
mutable struct HIPSmallNewton
    handle::Ptr{Cvoid}
    nx::Int; ne::Int; nc::Int; batch::Int
end
function HIPSmallNewton(nx::Integer, ne::Integer, nc::Integer, batch::Integer; device::Integer=0)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:calipso_hip_smallnewton_create, lib), Int32, (Int64, Int64, Int64, Int64, Int32, Ptr{Ptr{Cvoid}}), nx, ne, nc, batch, device, h)
    rc == 0 || error("calipso_hip_smallnewton_create failed ($rc): " * unsafe_string(ccall((:calipso_hip_smallnewton_last_error, lib), Cstring, (Ptr{Cvoid},), h[])))
    s = HIPSmallNewton(h[], nx, ne, nc, batch)
    finalizer(x -> ccall((:calipso_hip_smallnewton_destroy, lib), Int32, (Ptr{Cvoid},), x.handle), s)
    return s
end
sn_check(s::HIPSmallNewton, rc, what) = rc < 0 ? error("$what failed ($rc): " * unsafe_string(ccall((:calipso_hip_smallnewton_last_error, lib), Cstring, (Ptr{Cvoid},), s.handle))) : rc
"cone layout: the first `n_nonnegative` cone entries nonnegative, then second-order cones of the given dimensions (contiguous; 2 .. 16 entries each)"
function set_cones!(s::HIPSmallNewton, n_nonnegative::Integer, dims::Vector{Int64}=Int64[])
    sn_check(s, ccall((:calipso_hip_smallnewton_set_cones, lib), Int32, (Ptr{Cvoid}, Int64, Int64, Ptr{Int64}), s.handle, n_nonnegative, length(dims), isempty(dims) ? C_NULL : dims), "calipso_hip_smallnewton_set_cones")
end
"column-major arrays of ONE problem (shared = true) or stacked along a trailing batch dimension: P (nx, nx[, batch]), A (ne, nx[, batch]), G (nc, nx[, batch])"
function set_qp!(s::HIPSmallNewton, P, q, A, b, G, h; objective_scale::Float64=0.5, shared::Bool=ndims(P) == 2)
    f(a) = isempty(a) ? zeros(1) : collect(Float64, vec(a))
    sn_check(s, ccall((:calipso_hip_smallnewton_set_qp, lib), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Float64, Int32),
                      s.handle, f(P), f(q), f(A), f(b), f(G), f(h), objective_scale, shared ? 1 : 0), "calipso_hip_smallnewton_set_qp")
end
"initialize!(solver, guess) for every instance: x0 is nx x batch"
function initialize!(s::HIPSmallNewton, x0::AbstractMatrix)
    N = s.nx + 2 * s.ne + 3 * s.nc
    w = zeros(N, s.batch); w[1:s.nx, :] .= x0
    sn_check(s, ccall((:calipso_hip_smallnewton_set_state, lib), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), s.handle, w, C_NULL, C_NULL), "calipso_hip_smallnewton_set_state")
end
"solve!(solver) for every instance in one launch: (result per instance, launch milliseconds)"
function solve!(s::HIPSmallNewton)
    res = zeros(Int32, s.batch); ms = Ref{Float64}(0.0)
    sn_check(s, ccall((:calipso_hip_smallnewton_solve, lib), Int32, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Float64}), s.handle, res, ms), "calipso_hip_smallnewton_solve")
    return res, ms[]
end
"differentiate!(solver) for every instance in one launch (differentiate.jl:1-61): jacobian_parameters is N x p x batch (dR/dtheta per instance) or N x p (one matrix for all); returns (sensitivity N x p x batch, status, ms)"
function differentiate!(s::HIPSmallNewton, jacobian_parameters::AbstractArray{Float64})
    N = s.nx + 2 * s.ne + 3 * s.nc
    shared = ndims(jacobian_parameters) == 2
    size(jacobian_parameters, 1) == N && (shared || size(jacobian_parameters, 3) == s.batch) || error("jacobian_parameters must be N x p x batch or N x p")
    p = size(jacobian_parameters, 2)
    J = Array{Float64}(jacobian_parameters); sens = zeros(N, p, s.batch); st = zeros(Int32, s.batch); ms = Ref{Float64}(0.0)
    sn_check(s, ccall((:calipso_hip_smallnewton_differentiate, lib), Int32, (Ptr{Cvoid}, Int64, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ptr{Float64}), s.handle, p, shared ? 1 : 0, J, sens, st, ms), "calipso_hip_smallnewton_differentiate")
    return sens, st, ms[]
end
"solution.all of every instance (N x batch)"
function solution(s::HIPSmallNewton)
    N = s.nx + 2 * s.ne + 3 * s.nc
    w = zeros(N, s.batch)
    sn_check(s, ccall((:calipso_hip_smallnewton_get_state, lib), Int32, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int64}), s.handle, w, C_NULL, C_NULL, C_NULL), "calipso_hip_smallnewton_get_state")
    return w
end

# ---- multi-GPU exchange (include/calipso_hip.h: calipso_hip_comm_*): RCCL over xGMI, one process per GPU -------------------------
"RCCL communicator: `id = comm_unique_id()` on one rank, distributed by the launcher (file / MPI / Distributed.jl), then `HIPComm(rank, nranks, id; device)` on every rank."
mutable struct HIPComm
    handle::Ptr{Cvoid}
    rank::Int
    nranks::Int
end
function comm_unique_id()
    id = zeros(UInt8, 128)
    rc = ccall((:calipso_hip_comm_unique_id, lib), Int32, (Ptr{UInt8},), id)
    rc == 0 || error("calipso_hip_comm_unique_id failed ($rc)")
    return id
end
function HIPComm(rank::Integer, nranks::Integer, id::Vector{UInt8}; device::Integer=0)
    c = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall((:calipso_hip_comm_init, lib), Int32, (Int32, Int32, Ptr{UInt8}, Int32, Ptr{Ptr{Cvoid}}), rank, nranks, id, device, c)
    rc == 0 || error("calipso_hip_comm_init failed ($rc)")
    comm = HIPComm(c[], rank, nranks)
    finalizer(x -> ccall((:calipso_hip_comm_destroy, lib), Int32, (Ptr{Cvoid},), x.handle), comm)
    return comm
end
"(ranks, own rank) as the live communicator reports them (ncclCommCount / ncclCommUserRank)"
function comm_size(c::HIPComm)
    out = zeros(Int32, 2)
    rc = ccall((:calipso_hip_comm_size, lib), Int32, (Ptr{Cvoid}, Ptr{Int32}), c.handle, out)
    rc == 0 || error("calipso_hip_comm_size failed ($rc)")
    return Int(out[1]), Int(out[2])
end
"all-gather of the per-problem status rows (4 x k Int32 per rank, k may differ) in global problem-id order; `capacity` = total rows"
function gather_status(c::HIPComm, rows::Matrix{Int32}, capacity::Integer)
    out = zeros(Int32, 4, capacity); counts = zeros(Int64, c.nranks)
    n = ccall((:calipso_hip_comm_gather_status, lib), Int64, (Ptr{Cvoid}, Ptr{Int32}, Int64, Ptr{Int32}, Int64, Ptr{Int64}), c.handle, rows, size(rows, 2), out, capacity, counts)
    n < 0 && error("calipso_hip_comm_gather_status failed ($n)")
    return out[:, 1:n], counts
end
function allreduce_sum!(c::HIPComm, v::Vector{Float64})
    rc = ccall((:calipso_hip_comm_allreduce_sum, lib), Int32, (Ptr{Cvoid}, Ptr{Float64}, Int64), c.handle, v, length(v))
    rc == 0 || error("calipso_hip_comm_allreduce_sum failed ($rc)")
    return v
end

export HIPSolver, streams_concurrent, rebind_stream!, spread_streams!, HIPLDLSolver, HIPSparseLDLSolver, hip_sparse_ldl_solver, HIPKKTSolver, hip_ldl_solver, HIPGroup, HIPSmallNewton, set_cones!, set_qp!, solution, HIPComm, comm_unique_id, comm_size, gather_status, allreduce_sum!, newton_step!,
       search_direction_nonsymmetric!, analyze_structure!, clear_structure!, set_stage_parallel!, set_stage_blocks!, declared_structure, kernel_times, sync_scalars!, copy_back!

end # module
