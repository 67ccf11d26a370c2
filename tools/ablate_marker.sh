# diagnostic builds of k_time_eliminate with parts of its work left out (wrong results, never used for anything but timing):
# how much of the kernel's time each part is worth.  Built to a scratch path and loaded through RSBA_LIB: the packaged library is
# never touched (round 4 overwrote it in place and restored it on the last line only).
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
S=realsensecalibration_amd/csrc
T=$(mktemp -d /tmp/rsba_ab.XXXXXX); trap 'rm -rf "$T"' EXIT
python __graft_entry__.py >/dev/null   # (the host objects under build/obj)
for v in "" "-DRSBA_ABL_Q=1" "-DRSBA_ABL_SUMS=0" "-DRSBA_ABL_STAGE=0" "-DRSBA_ABL_Q=1 -DRSBA_ABL_SUMS=0 -DRSBA_ABL_STAGE=0"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -mllvm -amdgpu-kernarg-preload-count=16 $v -Wno-unused-result -I include -c $S/ba_solver.hip -o $T/s.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $T/librsba_variant.so $T/s.o build/obj/ba_problem.cpp.o build/obj/ba_initial_guess.cpp.o build/obj/rsba_capi.cpp.o -L/opt/rocm/lib -lrccl -pthread -Wl,-rpath,/opt/rocm/lib
  echo "variant [$v]: $(RSBA_LIB=$T/librsba_variant.so python tools/marker_chain_scale.py 8 5000 16 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['kernels_us']['k_time_eliminate'], d['iterations'])")"
done
