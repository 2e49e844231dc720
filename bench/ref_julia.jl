# ref_julia.jl — CPU row B2 of BASELINE.md: the reference itself (CALIPSO.jl v0.1.1) timed on ONE Newton step of the C3 synthetic conic QP
# (SURVEY.md 8(d): SplitMix64 streams, bit-identical inputs to bench.py / the oracle).  bench.py runs it only where a `julia` with CALIPSO's
# dependencies exists (the build container and the GPU boxes of this project have none: the row then says "julia unavailable").
#     julia --project=/path/to/CALIPSO.jl bench/ref_julia.jl [nx ne n_nonneg n_soc soc_dim]
# Prints one JSON line: {"value": steps/s, "seconds": ..., "factorizations": ..., "threads": 1}
using CALIPSO, LinearAlgebra, SparseArrays

function splitmix_uniform(problem_id::UInt64, stream_id::UInt64, lo, hi, count)
    state = 0xCA11B50000000000 + UInt64(4096) * problem_id + stream_id
    out = zeros(count)
    for i in 1:count
        state += 0x9E3779B97F4A7C15
        z = state
        z = (z ⊻ (z >> 30)) * 0xBF58476D1CE4E5B9
        z = (z ⊻ (z >> 27)) * 0x94D049BB133111EB
        z = z ⊻ (z >> 31)
        out[i] = lo + (hi - lo) * (Float64(z >> 11) * 2.0^-53)
    end
    return out
end

const STREAMS = Dict("B"=>1, "q"=>2, "A"=>3, "G"=>4, "xbar"=>5, "cone_point_tail"=>6, "x"=>10, "r"=>11, "y"=>12, "z"=>13, "lam"=>14,
                     "s_nn"=>15, "t_nn"=>16, "s_tail"=>17, "t_tail"=>18, "hpos"=>19)

function main()
    nx, ne, n_nn, n_soc, dim = length(ARGS) >= 5 ? parse.(Int, ARGS[1:5]) : (2500, 1500, 400, 200, 3)
    nc = n_nn + n_soc * dim
    U(name, lo, hi, cnt) = splitmix_uniform(UInt64(0), UInt64(STREAMS[name]), lo, hi, cnt)
    B = reshape(U("B", -1, 1, nx * nx), nx, nx)                      # column-major fill, as tests/problems.py: synthetic_conic_qp
    P = (B + B') / (2 * sqrt(nx)) + 2I
    q = U("q", -1, 1, nx)
    A = reshape(U("A", -1, 1, ne * nx) / sqrt(nx), ne, nx)
    G = reshape(U("G", -1, 1, nc * nx) / sqrt(nx), nc, nx)
    xbar = U("xbar", -1, 1, nx)
    b = A * xbar
    cp = zeros(nc); cp[1:n_nn] = U("hpos", 0.5, 1.5, n_nn)
    tails = U("cone_point_tail", -0.3, 0.3, nc)
    soc = [collect(n_nn + (k - 1) * dim + 1:n_nn + k * dim) for k in 1:n_soc]
    for c in soc
        cp[c[2:end]] = tails[c[2:end]]; cp[c[1]] = 1 + norm(cp[c[2:end]])
    end
    h = G * xbar + cp
    objective(z) = 0.5 * (transpose(z) * P * z) + transpose(q) * z
    equality(z) = A * z - b
    cone(z) = h - G * z
    solver = Solver(objective, equality, cone, nx; nonnegative_indices=collect(1:n_nn), second_order_indices=soc)   # Symbolics codegen: minutes at this size
    s = zeros(nc); t = zeros(nc)
    s[1:n_nn] = U("s_nn", 0.5, 1.5, n_nn); t[1:n_nn] = U("t_nn", 0.5, 1.5, n_nn)
    st = U("s_tail", -0.3, 0.3, nc); tt = U("t_tail", -0.3, 0.3, nc)
    for c in soc
        s[c[2:end]] = st[c[2:end]]; t[c[2:end]] = tt[c[2:end]]
        s[c[1]] = 1 + norm(s[c[2:end]]); t[c[1]] = 1 + norm(t[c[2:end]])
    end
    solver.solution.variables .= U("x", -1, 1, nx); solver.solution.equality_slack .= 0.1 * U("r", -1, 1, ne)
    solver.solution.cone_slack .= s; solver.solution.equality_dual .= U("y", -1, 1, ne)
    solver.solution.cone_dual .= U("z", -1, 1, nc); solver.solution.cone_slack_dual .= t
    solver.dual .= U("lam", -1, 1, ne); solver.central_path[1] = 0.17; solver.penalty[1] = 52.0
    step() = begin
        CALIPSO.evaluate!(solver.problem, solver.methods, solver.indices, solver.solution, solver.parameters,
            objective=true, objective_gradient_variables=true, objective_jacobian_variables_variables=true, equality_constraint=true,
            equality_jacobian_variables=true, equality_dual_jacobian_variables=true, equality_dual_jacobian_variables_variables=true,
            cone_constraint=true, cone_jacobian_variables=true, cone_dual_jacobian_variables=true, cone_dual_jacobian_variables_variables=true)
        CALIPSO.cone!(solver.problem, solver.cone_methods, solver.indices, solver.solution, barrier=true, barrier_gradient=true, product=true, jacobian=true, target=true)
        CALIPSO.residual!(solver.data, solver.problem, solver.indices, solver.solution, solver.central_path, solver.penalty, solver.dual)
        CALIPSO.search_direction!(solver)
    end
    step()                                                           # compile
    sec = @elapsed step()
    println("{\"value\": $(1 / sec), \"seconds\": $sec, \"threads\": 1, \"what\": \"CALIPSO.jl evaluate! + cone! + residual! + search_direction! on C3 problem 0\"}")
end
main()
