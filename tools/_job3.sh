cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "NCCL\|RCCL\|^$" | tail -30
