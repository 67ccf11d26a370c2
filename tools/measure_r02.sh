# round-2 measurements beyond the rocprof passes of tools/profile_r02.sh: default bench line (with the CPU baseline),
# block timeline of the Schur kernel, per-panel timeline of the Cholesky, the 256-camera shard, the RCCL path on one rank
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python bench.py > gpurun_out/r02_bench_default.json 2> gpurun_out/r02_bench_default.err
RSBA_TRACE=2 RSBA_TRACE_FILE=gpurun_out/wgtrace.txt python bench.py --no-cpu-baseline --steps 8 --warmup 2 > /dev/null 2> gpurun_out/r02_trace2.err
python tools/schur_timeline_summary.py gpurun_out/wgtrace.txt > gpurun_out/r02_schur_block_timeline.txt
RSBA_MC_TRACE=1 RSBA_TRACE=1 python bench.py --no-cpu-baseline --steps 6 --warmup 2 2>&1 | grep -E "rsba\[(mc|trace)\]" | tail -100 > gpurun_out/r02_cholesky_diag_timeline_pipelined.txt
python tools/mc_chain.py gpurun_out/r02_cholesky_diag_timeline_pipelined.txt >> gpurun_out/r02_cholesky_diag_timeline_pipelined.txt
python bench.py --config cfg5 --points 62500 --no-cpu-baseline --steps 20 > gpurun_out/r02_bench_cfg5_shard.json 2>/dev/null
python bench.py --config cfg4 --points 125000 --no-cpu-baseline --steps 30 > gpurun_out/r02_bench_cfg4_shard.json 2>/dev/null
RSBA_FORCE_COMM=1 python bench.py --no-cpu-baseline --steps 30 > gpurun_out/r02_bench_comm1rank_sequential.json 2>/dev/null
RSBA_FORCE_COMM=1 RSBA_PIPELINE_MG=1 python bench.py --no-cpu-baseline --steps 30 > gpurun_out/r02_bench_comm1rank_pipelined.json 2>/dev/null
RSBA_PIPELINE=0 python bench.py --no-cpu-baseline --steps 30 > gpurun_out/r02_bench_sequential.json 2>/dev/null
for f in default cfg5_shard cfg4_shard comm1rank_sequential comm1rank_pipelined sequential; do python -c "
import json
d=json.loads(open('gpurun_out/r02_bench_$f.json').read().strip().splitlines()[-1])
print('$f', round(d['ms_per_step'],4), {k:round(v['avg_us'],1) for k,v in d.get('kernels',{}).items()})"; done
