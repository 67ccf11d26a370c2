# diagnostic build with in-kernel phase stamps of k_time_eliminate's first workgroup (never used for timing)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
S=realsensecalibration_amd/csrc
mkdir -p /tmp/pp && cp realsensecalibration_amd/librsba.so /tmp/pp/librsba.so.bak
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -DRSBA_PROFILE_PHASES -Wno-unused-result -I include -c $S/ba_solver.hip -o /tmp/pp/s.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o realsensecalibration_amd/librsba.so /tmp/pp/s.o build/obj/ba_problem.cpp.o build/obj/ba_initial_guess.cpp.o build/obj/rsba_capi.cpp.o -L/opt/rocm/lib -lrccl -pthread -Wl,-rpath,/opt/rocm/lib
touch realsensecalibration_amd/librsba.so
python tools/marker_chain_scale.py ${1:-8} ${2:-5000} ${3:-16} 2>&1 | grep "rsba\[mt-phases\]\|rsba\[phases\]" | tail -2
cp /tmp/pp/librsba.so.bak realsensecalibration_amd/librsba.so
