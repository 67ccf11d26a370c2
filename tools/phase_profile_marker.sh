# diagnostic build with in-kernel phase stamps of k_time_eliminate's first workgroup (never used for timing); built to a scratch
# path and loaded through RSBA_LIB — the packaged library is not touched
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
S=realsensecalibration_amd/csrc
T=$(mktemp -d /tmp/rsba_pp.XXXXXX); trap 'rm -rf "$T"' EXIT
python __graft_entry__.py >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -DRSBA_PROFILE_PHASES -Wno-unused-result -I include -c $S/ba_solver.hip -o $T/s.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $T/librsba_variant.so $T/s.o build/obj/ba_problem.cpp.o build/obj/ba_initial_guess.cpp.o build/obj/rsba_capi.cpp.o -L/opt/rocm/lib -lrccl -pthread -Wl,-rpath,/opt/rocm/lib
RSBA_LIB=$T/librsba_variant.so python tools/marker_chain_scale.py ${1:-8} ${2:-5000} ${3:-16} 2>&1 | grep "rsba\[mt-phases\]\|rsba\[phases\]" | tail -2
