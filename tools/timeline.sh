# kernel timeline of the bench (rocprofv3 kernel trace), summarised per iteration by tools/timeline.py
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
rm -rf gpurun_out/prof_tl
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_tl -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-events > gpurun_out/prof_tl.json 2> gpurun_out/prof_tl.err
python3 tools/timeline.py gpurun_out/prof_tl
