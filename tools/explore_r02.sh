# round-2 exploration: Cholesky workgroups, segment counts, phase profile of the diagonal factorisation
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { python bench.py --no-cpu-baseline --steps 40 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],4), {k:round(v['avg_us'],1) for k,v in d['kernels'].items() if k in ('k_schur_tiles','k_reduced_system_solve')})"; }
for w in 3 4 5 6; do RSBA_CHOL_WGS=$w run "chol_wgs=$w"; done
for sp in 6 8 10 12; do RSBA_SEG_PER_CU=$sp run "seg_per_cu=$sp"; done
for g in 4 8 16; do RSBA_GRP=$g run "grp=$g"; done
RSBA_BALANCE=64 run "balance=64"
bash tools/phase_profile.sh
RSBA_MC_TRACE=1 RSBA_TRACE=1 python bench.py --no-cpu-baseline --steps 6 --warmup 2 2>&1 | grep -E "rsba\[(mc|trace)\]" | tail -49 > gpurun_out/r02_mc_trace.txt
