cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_kt -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/prof_kt.json 2> gpurun_out/prof_kt.err
ls -R gpurun_out/prof_kt | head -20
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d gpurun_out/prof_pmc1 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_pmc1.json 2> gpurun_out/prof_pmc1.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_THREAD_CYCLES_VALU --output-format csv -d gpurun_out/prof_pmc2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/prof_pmc2.json 2> gpurun_out/prof_pmc2.err
ls -R gpurun_out/prof_pmc1 | head; tail -3 gpurun_out/prof_pmc1.err
