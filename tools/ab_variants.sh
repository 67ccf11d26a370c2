# A/B of prebuilt library variants (build/variants/librsba_<name>.so, tools/build_variant.sh): bench each twice through RSBA_LIB —
# the packaged library is not touched.  "default" names the packaged build.  BENCH_ARGS: extra bench.py arguments.
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { RSBA_LIB=$2 python bench.py --no-cpu-baseline --steps 50 --warmup 3 $BENCH_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],4), round(d.get('ms_per_step_steady') or 0,4), {k:round(v['avg_us'],1) for k,v in d['kernels'].items() if k in ('k_schur_tiles','k_reduced_system_solve','k_backsub_candidate','k_chol_tiles_persistent','k_backsub_chain')})"; }
for v in "$@"; do
  lib=build/variants/librsba_$v.so; [ "$v" = default ] && lib=
  run $v $lib; run $v $lib
done
