#!/bin/bash
# throughput of the batched small-problem kernel against the threads per instance (options.threads) and the build flags in SN_EXTRA (GPU box; the plain build is restored)
#   bash bench/small_newton_nt.sh "128 256" [rate-script arguments]
R=${GRAFT_REPO_ROOT:-$(pwd)}
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result"
NTS=${1:-"128 256"}; shift
if [ -n "$SN_EXTRA" ]; then cd $R/calipso.jl_amd/csrc && hipcc $FL $SN_EXTRA -c smallnewton.hip -o smallnewton.o && hipcc --offload-arch=gfx950 -shared -fPIC -o ../libcalipso_hip.so *.o -ldl; fi
cd $R
for nt in $NTS; do echo "threads=$nt $SN_EXTRA"; SN_THREADS=$nt python bench/small_newton_rate.py "$@"; done
if [ -n "$SN_EXTRA" ]; then cd $R/calipso.jl_amd/csrc && hipcc $FL -c smallnewton.hip -o smallnewton.o && hipcc --offload-arch=gfx950 -shared -fPIC -o ../libcalipso_hip.so *.o -ldl; fi
