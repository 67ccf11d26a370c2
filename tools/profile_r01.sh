# rocprofv3 evidence for round 1: kernel trace + stats of the default bench command, then PMC passes.
# The PMC passes run the sequential schedule: counter collection serialises kernels, and the pipelined Cholesky (which
# waits inside the kernel for the Schur kernel launched after it) would only time out and fall back.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
rm -rf gpurun_out/prof_kt gpurun_out/prof_pmc_*
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -- python3 bench.py --no-cpu-baseline > gpurun_out/prof_kt.json 2> gpurun_out/prof_kt.err
export RSBA_PIPELINE=0
rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_pmc_fetch -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/prof_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_pmc_write -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/prof_pmc_write.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/prof_pmc_sq -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/prof_pmc_sq.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d gpurun_out/prof_pmc_sq2 -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/prof_pmc_sq2.err
unset RSBA_PIPELINE
python3 tools/pmc_to_json.py gpurun_out/prof_pmc_fetch gpurun_out/prof_pmc_write gpurun_out/r01_pmc.json
python3 tools/pmc_summary.py gpurun_out/prof_pmc_fetch gpurun_out/prof_pmc_write gpurun_out/prof_pmc_sq gpurun_out/prof_pmc_sq2 > gpurun_out/r01_pmc_summary.txt
cp $(ls gpurun_out/prof_kt/*/*kernel_stats.csv | head -1) gpurun_out/r01_kernel_stats.csv
tail -1 gpurun_out/prof_kt.json > gpurun_out/r01_bench_under_rocprof.json
# marker-chain model at scale (time-block elimination): kernel trace of tools/marker_chain_scale.py
rm -rf gpurun_out/prof_mc
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_mc -- python3 tools/marker_chain_scale.py 8 5000 16 > gpurun_out/r01_marker_chain_scale.json 2> gpurun_out/prof_mc.err
cp $(ls gpurun_out/prof_mc/*/*kernel_stats.csv | head -1) gpurun_out/r01_marker_chain_kernel_stats.csv
