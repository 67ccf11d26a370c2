"""Every RSBA_* switch that selects a code path of the product, with the path it selects held to the parity bar.

The switches exist so that a fallback (sequential schedule, one-workgroup or round-robin factorisation, multi-launch
factorisation above 64 cameras, the masked search instead of the hit lists, the round-1 back-substitution, ...) can be forced
and so that two implementations of the same step can be compared; a path nobody tests is a path nobody can rely on, so each of
them solves the same problems against the oracle here (same iteration count, decisions and stop reason, every iterate's cost
to 1e-9, raw parameters to 1e-6 per block, bitwise reproducible).  They are read once per process: one child process per
setting (tests/switch_worker.py).  The reference has no counterpart: it calls ceres::Solve once, single-threaded
(Test1_BundleAdjustment/main.cpp:82-87)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SETTINGS = [
    ({}, ["c40", "c64_huber", "c70_huber", "c130", "c240"]),
    ({"RSBA_PIPELINE": "0"}, ["c40", "c64_huber"]),
    ({"RSBA_CHOL_DIAG": "0"}, ["c40", "c64_huber"]),
    ({"RSBA_CHOL_WGS": "1"}, ["c40", "c64_huber"]),
    ({"RSBA_CHOL_WGS": "3"}, ["c64_huber"]),
    ({"RSBA_BACKSUB_PROJ": "0"}, ["c40", "c64_huber", "c130"]),
    ({"RSBA_FUSED_LIN": "0"}, ["c40", "c64_huber"]),
    ({"RSBA_DECIDED_DAMP": "0"}, ["c40", "c64_huber"]),
    ({"RSBA_FIRST_STAGED": "0"}, ["c40", "c64_huber"]),
    ({"RSBA_BALANCE": "0"}, ["c40", "c64_huber"]),
    ({"RSBA_SEG_PER_CU": "4"}, ["c64_huber", "c130"]),
    ({"RSBA_SPARSE_PAIRS": "0"}, ["c70_huber", "c130"]),
    ({"RSBA_CHOL_TILES": "0"}, ["c70_huber", "c130"]),
    ({"RSBA_TILE_ORDER": "0"}, ["c130", "c240"]),
    ({"RSBA_SYS_FUSED": "0"}, ["c70_huber", "c130"]),
    ({"RSBA_BACKSUB_MULTI": "0"}, ["c70_huber", "c130"]),
    ({"RSBA_BACKSUB_MULTI": "1"}, ["c70_huber", "c130"]),
    ({"RSBA_FORCE_COMM": "1"}, ["c40", "c64_huber", "c70_huber"]),
    ({"RSBA_FORCE_COMM": "1", "RSBA_PIPELINE_MG": "1"}, ["c40", "c64_huber"]),
    ({"RSBA_FORCE_COMM": "1", "RSBA_BACKSUB_PROJ": "0"}, ["c40"]),
]


@pytest.mark.parametrize("env,cases", SETTINGS, ids=[" ".join("%s=%s" % kv for kv in e.items()) or "defaults" for e, _ in SETTINGS])
def test_switch_selects_a_path_that_matches_the_oracle(env, cases):
    child_env = dict(os.environ)
    child_env.update(env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "switch_worker.py")] + cases, env=child_env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])   # (RCCL prints its banner to stdout as well)
    for name in cases:
        r = res[name]
        assert r["iterations"] == r["iterations_ref"] and r["same_decisions"] and r["stop"] == r["stop_ref"], (name, r)
        assert r["iterate_costs"] < 1e-9 and r["final_cost_rel"] < 1e-9, (name, r)
        assert r["raw"] < 1e-6, (name, r)
        assert r["reproducible"], (name, r)
    assert "stalled" not in out.stderr, out.stderr[-2000:]
