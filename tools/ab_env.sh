# A/B of environment settings on the packaged library: tools/ab_env.sh "<label>:<VAR=val ...>" ...   (each twice, alternating; "base:" = no setting)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { label=$1; shift; env "$@" python bench.py --no-cpu-baseline --steps 50 --warmup 3 $BENCH_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', round(d['ms_per_step'],4), round(d.get('ms_per_step_steady') or 0,4), d['timed_region']['clean'], {k:round(v['avg_us'],1) for k,v in d['kernels'].items() if k in ('k_schur_tiles','k_reduced_system_solve','k_backsub_candidate','k_chol_tiles_persistent','k_backsub_chain')})"; }
for rep in 1 2; do
  for spec in "$@"; do
    label=${spec%%:*}; vars=${spec#*:}
    if [ -z "$vars" ]; then run $label RSBA_NOP=1; else run $label $vars; fi
  done
done
