# A/B of prebuilt library variants (build/variants/librsba_<name>.so): bench each twice, restore the default build
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
cp realsensecalibration_amd/librsba.so /tmp/librsba.default.so
run() { python bench.py --no-cpu-baseline --steps 50 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],4), {k:round(v['avg_us'],1) for k,v in d['kernels'].items() if k in ('k_schur_tiles','k_reduced_system_solve')})"; }
for v in "$@"; do
  cp build/variants/librsba_$v.so realsensecalibration_amd/librsba.so; touch realsensecalibration_amd/librsba.so
  run $v; run $v
done
cp /tmp/librsba.default.so realsensecalibration_amd/librsba.so; touch realsensecalibration_amd/librsba.so
