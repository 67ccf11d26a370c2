cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/prof_pmc1 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_pmc1.json 2> gpurun_out/prof_pmc1.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/prof_pmc2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_pmc2.json 2> gpurun_out/prof_pmc2.err
