#!/bin/bash
# One command for a maintainer who HAS Julia (>= 1.7) and CALIPSO.jl's dependencies: run the reference itself on the committed known-answer
# inputs and flip the oracle's parity status from "unpinned" to pinned.
#   bash bench/pin_with_julia.sh /path/to/CALIPSO.jl        (the checkout; its Project.toml is instantiated on first use)
# 1. bench/ref_fixtures.jl drives CALIPSO's own functions (cone!, residual!, residual_jacobian_variables(_symmetric)!, factorize!,
#    search_direction!, iterative_refinement!, inertia_correction!, cone_violation ...) on tests/golden/kat_*_inputs.txt and writes
#    tests/golden/ref_kat_*.txt (H, K, R, b, step, inertia, alpha_s / alpha_t, merit, theta — produced by the REFERENCE, not by oracle/) and, for the two
#    search_direction! cases (a non-convex Hessian: the IC-1 .. IC-6 sequence; a portfolio step through a second-order cone of dimension 12),
#    tests/golden/ref_kat_sd_*.txt (residual, inertia, the final regularisation, the step).
# 2. tests/test_reference_fixtures.py then compares the oracle with those files (it skips, loudly, while they are absent).
# 3. bench/ref_julia.jl (optional, B2 row of bench.py) times CALIPSO.search_direction! on the C3 SplitMix64 inputs.
# Nothing here runs on the GPU box or in the build container (no Julia in either); commit the ref_kat_*.txt files it produces.
set -euo pipefail
REF=${1:?usage: bash bench/pin_with_julia.sh /path/to/CALIPSO.jl}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
command -v julia >/dev/null || { echo "julia not found on PATH" >&2; exit 2; }
julia --project="$REF" -e 'using Pkg; Pkg.instantiate()'
julia --project="$REF" "$ROOT/bench/ref_fixtures.jl"
ls -l "$ROOT"/tests/golden/ref_kat_*.txt
cd "$ROOT"
python -m pytest tests/test_reference_fixtures.py -q -rs
echo "parity pin: tests/golden/ref_kat_*.txt written by the reference; commit them.  (B2 timing: CALIPSO_JL_PROJECT=$REF python bench.py)"
