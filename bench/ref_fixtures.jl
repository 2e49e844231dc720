# ref_fixtures.jl — turns the two known-answer Newton steps of tests/golden/ into REFERENCE-generated fixtures.
#
# The build container has no Julia, so tests/golden/kat_*.npz are produced by this repo's CPU oracle ("parity unpinned",
# DESIGN.md section 2).  Wherever Julia >= 1.7 with CALIPSO.jl v0.1.1 (and its dependencies) is available, run
#
#     julia --project=/path/to/CALIPSO.jl bench/ref_fixtures.jl
#
# It reads the inputs of each KAT problem from tests/golden/<name>_inputs.txt (written by tests/golden/make_golden.py: the QP data
# P, q, A, b, G, h, the cone index sets, the iterate w, the dual, and the constants kappa = .17, rho = 52, eps_p = .12, eps_d = .21
# of test/solver/problem.jl:56-65), drives the reference's own functions in the order of test/solver/problem.jl:76-211, and writes
# tests/golden/ref_<name>.txt with: H (jacobian_variables, dense), K (jacobian_variables_symmetric, dense, both triangles as the
# reference writes them), R (residual), b (residual_symmetric), inertia, the first (unrefined) step, the refined step, the cone
# step sizes, merit, theta, merit gradient and the optimality error.  tests/test_reference_fixtures.py then checks the oracle (CPU)
# and the HIP path (GPU) against those files — that is the step which turns "parity unpinned" into pinned.
#
# Text format (both directions): one array per record, "name rows cols" on a line followed by rows*cols numbers (column-major),
# printed with 17 significant digits (round-trips fp64).
using CALIPSO
using LinearAlgebra
using SparseArrays

const HERE = @__DIR__
const GOLDEN = joinpath(HERE, "..", "tests", "golden")

function read_records(path)
    out = Dict{String,Matrix{Float64}}()
    toks = split(read(path, String))
    i = 1
    while i <= length(toks)
        name = String(toks[i]); r = parse(Int, toks[i+1]); c = parse(Int, toks[i+2]); i += 3
        vals = [parse(Float64, toks[i+k]) for k in 0:(r*c-1)]; i += r * c
        out[name] = reshape(vals, r, c)
    end
    return out
end

function write_record(io, name, A)
    M = A isa Number ? fill(Float64(A), 1, 1) : (ndims(A) == 1 ? reshape(Vector{Float64}(A), :, 1) : Matrix{Float64}(A))
    println(io, name, " ", size(M, 1), " ", size(M, 2))
    for v in vec(M)
        println(io, repr(v))
    end
end

function run_kat(name)
    d = read_records(joinpath(GOLDEN, name * "_inputs.txt"))
    P, q, A, b, G, h = d["P"], vec(d["q"]), d["A"], vec(d["b"]), d["G"], vec(d["h"])
    nx = length(q)
    scale = d["objective_scale"][1]
    nonneg = Int.(vec(d["nonnegative_indices"]))
    ptr = Int.(vec(d["second_order_ptr"]))
    flat = Int.(vec(d["second_order_indices"]))
    soc = [flat[ptr[k]+1:ptr[k+1]] for k in 1:length(ptr)-1]
    isempty(soc) && (soc = [Int[]])
    objective(z) = scale * (transpose(z) * P * z) + transpose(q) * z
    equality(z) = A * z - b
    cone(z) = h - G * z
    solver = Solver(objective, equality, cone, nx; nonnegative_indices=nonneg, second_order_indices=soc)
    solver.solution.all .= vec(d["w"])
    κ = [d["central_path"][1]]; ρ = [d["penalty"][1]]; λ = vec(d["dual"])
    ϵp = d["primal_regularization"][1]; ϵd = d["dual_regularization"][1]; τ = d["fraction_to_boundary"][1]
    solver.central_path .= κ; solver.penalty .= ρ; solver.dual .= λ
    solver.primal_regularization[1] = ϵp; solver.dual_regularization[1] = ϵd; solver.fraction_to_boundary[1] = τ
    idx = solver.indices
    CALIPSO.evaluate!(solver.problem, solver.methods, idx, solver.solution, solver.parameters,
        objective=true, objective_gradient_variables=true, objective_jacobian_variables_variables=true,
        equality_constraint=true, equality_jacobian_variables=true, equality_dual_jacobian_variables=true,
        equality_dual_jacobian_variables_variables=true, cone_constraint=true, cone_jacobian_variables=true,
        cone_dual_jacobian_variables=true, cone_dual_jacobian_variables_variables=true)
    CALIPSO.cone!(solver.problem, solver.cone_methods, idx, solver.solution,
        barrier=true, barrier_gradient=true, product=true, jacobian=true, target=true)
    CALIPSO.residual_jacobian_variables!(solver.data, solver.problem, idx, κ, ρ, λ, ϵp, ϵd)
    CALIPSO.residual_jacobian_variables_symmetric!(solver.data.jacobian_variables_symmetric, solver.data.jacobian_variables, idx,
        solver.problem.second_order_jacobians, solver.problem.second_order_jacobians_inverse)
    CALIPSO.residual!(solver.data, solver.problem, idx, solver.solution, κ, ρ, λ)
    CALIPSO.residual_symmetric!(solver.data.residual_symmetric, solver.data.residual, solver.data.residual_second_order,
        solver.data.jacobian_variables, idx)
    open(joinpath(GOLDEN, "ref_" * name * ".txt"), "w") do io
        write_record(io, "H", Matrix(solver.data.jacobian_variables))
        write_record(io, "K", Matrix(solver.data.jacobian_variables_symmetric))      # before factorize!(update=true) applies triu!
        write_record(io, "residual", solver.data.residual.all)
        write_record(io, "residual_symmetric", solver.data.residual_symmetric.all)
        write_record(io, "cone_product", solver.problem.cone_product)
        write_record(io, "cone_target", solver.problem.cone_target)
        write_record(io, "barrier", solver.problem.barrier[1])
        write_record(io, "barrier_gradient", solver.problem.barrier_gradient)
        # factorize! + compute_inertia!  (inertia.jl:17-28)
        CALIPSO.factorize!(solver.linear_solver, solver.data.jacobian_variables_symmetric; update=solver.options.update_factorization)
        CALIPSO.compute_inertia!(solver.linear_solver)
        inr = solver.linear_solver.inertia
        write_record(io, "inertia", Float64[inr.positive, inr.negative, inr.zero])
        write_record(io, "permutation", Float64.(solver.linear_solver.F.perm))       # AMD order: the third-party quantity SURVEY 8(c) calls unpinned
        CALIPSO.search_direction_symmetric!(solver.data.step, solver.data.residual, solver.data.jacobian_variables,
            solver.data.step_symmetric, solver.data.residual_symmetric, solver.data.jacobian_variables_symmetric, idx,
            solver.data.step_second_order, solver.data.residual_second_order, solver.linear_solver;
            update=solver.options.update_factorization)
        write_record(io, "step_first", solver.data.step.all)
        ok = CALIPSO.iterative_refinement!(solver.data.step, solver)
        write_record(io, "refinement_ok", ok ? 1.0 : 0.0)
        write_record(io, "step", solver.data.step.all)
        # cone fraction-to-boundary search  solve.jl:190-221
        s = solver.solution.cone_slack; t = solver.solution.cone_slack_dual
        Δs = solver.data.step.cone_slack; Δt = solver.data.step.cone_slack_dual
        αs = 1.0; αt = 1.0
        while length(s) > 0 && CALIPSO.cone_violation(s - αs * Δs, s, τ, idx.cone_nonnegative, idx.cone_second_order)
            αs *= solver.options.scaling_line_search
        end
        while length(t) > 0 && CALIPSO.cone_violation(t - αt * Δt, t, τ, idx.cone_nonnegative, idx.cone_second_order)
            αt *= solver.options.scaling_line_search
        end
        write_record(io, "alpha", [αs, αt])
        M = CALIPSO.merit(solver.problem.objective[1], solver.solution.equality_slack, solver.problem.barrier[1], κ[1], λ, ρ[1])
        write_record(io, "merit", M)
        θ = CALIPSO.constraint_violation!(solver.data.constraint_violation, solver.problem.equality_constraint, solver.solution.equality_slack,
            solver.problem.cone_constraint, solver.solution.cone_slack, idx, norm_type=solver.options.constraint_norm)
        write_record(io, "theta", θ)
        CALIPSO.merit_gradient!(solver.data.merit_gradient, solver.problem.objective_gradient_variables, solver.solution.equality_slack,
            solver.problem.barrier_gradient, κ[1], λ, ρ[1], idx)
        write_record(io, "merit_gradient", solver.data.merit_gradient)
        write_record(io, "optimality_error", CALIPSO.optimality_error(solver.solution, solver.data.residual, idx))   # optimality_error.jl:1
    end
    println("wrote ", joinpath(GOLDEN, "ref_" * name * ".txt"))
end

# search_direction! as a whole (search_direction.jl:1-23) from the default regularisation start: the cases whose answers no closed form of test/solver/problem.jl holds —
# the IC-1 .. IC-6 sequence of inertia_correction! (inertia.jl:30-80) on a non-convex Hessian, and a second-order cone of dimension 12 (test/solver/portfolio.jl:33-62)
# going through the condensed blocks, the triu-only factorisation and the refinement.  Inputs: tests/golden/<name>_inputs.txt (make_golden.py: kat_search_direction).
function run_search_direction(name)
    d = read_records(joinpath(GOLDEN, name * "_inputs.txt"))
    P, q, A, b, G, h = d["P"], vec(d["q"]), d["A"], vec(d["b"]), d["G"], vec(d["h"])
    nx = length(q)
    scale = d["objective_scale"][1]
    nonneg = Int.(vec(d["nonnegative_indices"]))
    ptr = Int.(vec(d["second_order_ptr"]))
    flat = Int.(vec(d["second_order_indices"]))
    soc = [flat[ptr[k]+1:ptr[k+1]] for k in 1:length(ptr)-1]
    isempty(soc) && (soc = [Int[]])
    objective(z) = scale * (transpose(z) * P * z) + transpose(q) * z
    equality(z) = A * z - b
    cone(z) = h - G * z
    solver = Solver(objective, equality, cone, nx; nonnegative_indices=nonneg, second_order_indices=soc)
    solver.solution.all .= vec(d["w"])
    solver.central_path .= d["central_path"][1]; solver.penalty .= d["penalty"][1]; solver.dual .= vec(d["dual"])
    solver.fraction_to_boundary[1] = d["fraction_to_boundary"][1]
    solver.primal_regularization[1] = 0.0; solver.primal_regularization_last[1] = 0.0; solver.dual_regularization[1] = 0.0
    idx = solver.indices
    CALIPSO.evaluate!(solver.problem, solver.methods, idx, solver.solution, solver.parameters,
        objective=true, objective_gradient_variables=true, objective_jacobian_variables_variables=true,
        equality_constraint=true, equality_jacobian_variables=true, equality_dual_jacobian_variables=true,
        equality_dual_jacobian_variables_variables=true, cone_constraint=true, cone_jacobian_variables=true,
        cone_dual_jacobian_variables=true, cone_dual_jacobian_variables_variables=true)
    CALIPSO.cone!(solver.problem, solver.cone_methods, idx, solver.solution, product=true, jacobian=true, target=true)
    CALIPSO.residual!(solver.data, solver.problem, idx, solver.solution, solver.central_path, solver.penalty, solver.dual)
    CALIPSO.search_direction!(solver)
    CALIPSO.compute_inertia!(solver.linear_solver)
    inr = solver.linear_solver.inertia
    open(joinpath(GOLDEN, "ref_" * name * ".txt"), "w") do io
        write_record(io, "residual", solver.data.residual.all)
        write_record(io, "inertia", Float64[inr.positive, inr.negative, inr.zero])
        write_record(io, "primal_regularization", solver.primal_regularization[1])
        write_record(io, "primal_regularization_last", solver.primal_regularization_last[1])
        write_record(io, "dual_regularization", solver.dual_regularization[1])
        write_record(io, "step", solver.data.step.all)
    end
    println("wrote ", joinpath(GOLDEN, "ref_" * name * ".txt"))
end

for name in ("kat_qp_10_5_5", "kat_soc_6_3_9")
    run_kat(name)
end
for name in ("kat_sd_nonconvex_12_3_4", "kat_sd_portfolio_soc12")
    run_search_direction(name)
end
