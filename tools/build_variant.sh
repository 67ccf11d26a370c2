# build/variants/librsba_<name>.so: the library with extra compiler flags (-D switches of an experiment), for tools/ab_variants.sh.
# usage: tools/build_variant.sh <name> [flags...]
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
name=$1; shift
mkdir -p build/variants build/obj
python __graft_entry__.py >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -mllvm -amdgpu-kernarg-preload-count=16 "$@" -Wno-unused-result -I include -c realsensecalibration_amd/csrc/ba_solver.hip -o build/variants/s_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build/variants/librsba_$name.so build/variants/s_$name.o build/obj/ba_problem.cpp.o build/obj/ba_initial_guess.cpp.o build/obj/rsba_capi.cpp.o -L/opt/rocm/lib -lrccl -pthread -Wl,-rpath,/opt/rocm/lib
rm -f build/variants/s_$name.o
echo built build/variants/librsba_$name.so
