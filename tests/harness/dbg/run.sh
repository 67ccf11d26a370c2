cd tests/harness/dbg
echo "== B: C++ main against in-tree lib"
g++ -O1 -o main_b main.cpp -L../../../realsensecalibration_amd -lrsba -Wl,-rpath,$PWD/../../../realsensecalibration_amd -Wl,-rpath,/opt/rocm/lib && RSBA_DEBUG=1 ./main_b 2>&1 | tail -5
echo "== C: separate compile + link"
S=../../../realsensecalibration_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I ../../../include -c $S/ba_solver.hip -o s.o 2>&1 | grep -v warning | head -3
g++ -O2 -std=c++17 -fPIC -I ../../../include -c $S/ba_problem.cpp -o p.o
g++ -O2 -std=c++17 -fPIC -I ../../../include -c $S/rsba_capi.cpp -o c.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libc/librsba.so s.o p.o c.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib 2>&1 | head
mkdir -p libc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libc/librsba.so s.o p.o c.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
g++ -O1 -o main_c main.cpp -Llibc -lrsba -Wl,-rpath,$PWD/libc -Wl,-rpath,/opt/rocm/lib && RSBA_DEBUG=1 ./main_c 2>&1 | tail -5
echo "== D: no rccl link, no unsafe atomics"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I ../../../include -c $S/ba_solver.hip -o s2.o 2>&1 | grep -v warning | head -3
mkdir -p libd && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libd/librsba.so s2.o p.o c.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
g++ -O1 -o main_d main.cpp -Llibd -lrsba -Wl,-rpath,$PWD/libd -Wl,-rpath,/opt/rocm/lib && RSBA_DEBUG=1 ./main_d 2>&1 | tail -5
echo "== E: O1"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -std=c++17 -fPIC -munsafe-fp-atomics -I ../../../include -c $S/ba_solver.hip -o s3.o 2>&1 | grep -v warning | head -3
mkdir -p libe && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libe/librsba.so s3.o p.o c.o -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
g++ -O1 -o main_e main.cpp -Llibe -lrsba -Wl,-rpath,$PWD/libe -Wl,-rpath,/opt/rocm/lib && RSBA_DEBUG=1 AMD_LOG_LEVEL=2 ./main_e 2>&1 | tail -12
