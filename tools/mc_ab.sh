# usage: bash tools/mc_ab.sh "VAR=val ..." ...   — the marker-chain benchmark (8 x 5000 x 16) under each environment, one line each
for v in "$@"; do
  env $v MC_RUNS=6 timeout 300 python tools/marker_chain_scale.py 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('[$v]', round(d['ms_per_iteration'],4), d['kernels_us'])"
done
