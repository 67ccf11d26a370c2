cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r05b
RSBA_BORDER=1 RSBA_TRACE=1 RSBA_MC_TRACE=1 python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 --no-events > /dev/null 2> gpurun_out/r05b/tr2.err
grep "rsba\[trace\]" gpurun_out/r05b/tr2.err | tail -1 | cut -c1-420
grep "rsba\[mc\] border" gpurun_out/r05b/tr2.err | tail -1
python3 tools/mc_chain.py gpurun_out/r05b/tr2.err | sed -n 6,9p
