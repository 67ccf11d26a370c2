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
    ({"RSBA_BORDER": "0"}, ["c40", "c64_huber", "c33_long"]),   # (every camera group through the diagonal-chain kernel: the default before the border)
    ({"RSBA_BORDER": "0", "RSBA_PIPELINE": "0"}, ["c40", "c64_huber"]),
    ({"RSBA_CHOL_WGS": "1"}, ["c40", "c64_huber"]),
    ({"RSBA_CHOL_WGS": "3"}, ["c64_huber"]),
    ({"RSBA_CHOL_WGS": "8"}, ["c40", "c64_huber"]),   # (seven row workgroups + the border's: nine workgroups)
    ({"RSBA_BACKSUB_PROJ": "0"}, ["c40", "c64_huber", "c130"]),
    ({"RSBA_FUSED_LIN": "0"}, ["c40", "c64_huber"]),
    ({"RSBA_DECIDED_DAMP": "0"}, ["c40", "c64_huber"]),
    ({"RSBA_FIRST_STAGED": "0"}, ["c40", "c64_huber"]),
    ({"RSBA_BALANCE": "0"}, ["c40", "c64_huber"]),
    ({"RSBA_RED_DELAY": "0"}, ["c40", "c64_huber"]),   # (a stage's reducers right behind its own compute entries: the order before round 5's last change)
    ({"RSBA_SEG_PER_CU": "4"}, ["c64_huber", "c130"]),
    ({"RSBA_SPARSE_PAIRS": "0"}, ["c70_huber", "c130"]),
    ({"RSBA_SEG_PER_CU": "1"}, ["c33_long", "c64_huber", "c8_dense", "c64_dense_huber"]),   # (long segments of several chunks; dense visibility)
    ({"RSBA_TILES_SMALL": "1", "RSBA_PIPELINE": "0"}, ["c40", "c64_huber", "c33_long"]),   # (the tiled factorisation up to 64 cameras, sequential schedule)
    ({"RSBA_CHOL_TILES": "0"}, ["c70_huber", "c130"]),
    ({"RSBA_TILE_ORDER": "0"}, ["c130", "c240"]),
    ({"RSBA_SYS_FUSED": "0"}, ["c70_huber", "c130"]),
    ({"RSBA_BACKSUB_MULTI": "0"}, ["c70_huber", "c130"]),
    ({"RSBA_BACKSUB_MULTI": "1"}, ["c70_huber", "c130"]),
    ({"RSBA_FORCE_COMM": "1"}, ["c40", "c64_huber", "c70_huber"]),
    ({"RSBA_FORCE_COMM": "1", "RSBA_PIPELINE_MG": "0"}, ["c40", "c64_huber"]),   # (the sequential multi-GPU schedule; the pipelined one is the default)
    ({"RSBA_FORCE_COMM": "1", "RSBA_TRI_PAYLOAD": "1", "RSBA_PIPELINE_MG": "0"}, ["c40", "c70_huber"]),
    ({"RSBA_FORCE_COMM": "1", "RSBA_TRI_PAYLOAD": "0"}, ["c70_huber"]),
    ({"RSBA_FORCE_COMM": "1", "RSBA_BACKSUB_PROJ": "0"}, ["c40"]),
]


# Experimental paths — the step launched ahead on the device's decision (RSBA_LAUNCH_AHEAD), the tiled factorisation gated beside the
# Schur kernel above 64 cameras (RSBA_PIPELINE_TILES), the round-robin factorisation (RSBA_CHOL_DIAG=0) — exist only in libraries built
# with -DRSBA_EXPERIMENTAL (tools/build_variant.sh exp -DRSBA_EXPERIMENTAL; RSBA_LIB points the tests at such a build):
# RSBA_TEST_EXPERIMENTAL=1 adds their settings.
if os.environ.get("RSBA_TEST_EXPERIMENTAL") == "1":
    SETTINGS += [({"RSBA_PIPELINE_TILES": "1"}, ["c70_huber", "c130", "c240"]), ({"RSBA_CHOL_DIAG": "0"}, ["c40", "c64_huber"]),
                 ({"RSBA_LAUNCH_AHEAD": "1"}, ["c40", "c64_huber", "c8_dense"])]


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


@pytest.mark.skipif(os.environ.get("RSBA_TEST_EXPERIMENTAL") != "1", reason="launch-ahead exists in -DRSBA_EXPERIMENTAL builds only")
def test_steps_launched_ahead_add_the_same_bits(tmp_path):
    """Up to 64 cameras on one GPU the NEXT step's factorisation and Schur kernel can be queued on the device's decision before the
    host has the step's result (launch_ahead in PointsStep, RSBA_LAUNCH_AHEAD=1; default: the host launches them once it has decided).
    The same kernels on the same state either way: whole solves — accepted and rejected steps, runs a tolerance ends with a
    step launched ahead still queued, runs the iteration limit ends — agree bit for bit, and so does a second run of the same
    solver (the worker's reproducibility check) behind a step that ran out unused."""
    cases = ["c40", "c64_huber", "c8_dense", "c33_long", "c64_dense_huber"]
    files = []
    for ahead in ("0", "1"):
        env = dict(os.environ, RSBA_LAUNCH_AHEAD=ahead)
        f = str(tmp_path / ("ahead%s.npz" % ahead))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "switch_worker.py"), "--dump", f] + cases, env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        assert "stalled" not in out.stderr, out.stderr[-2000:]
        files.append(f)
    import numpy as np
    a, b = np.load(files[0]), np.load(files[1])
    for name in a.files:
        assert np.array_equal(a[name], b[name]), name


def test_thousands_of_pipelined_steps_add_the_same_bits_as_the_sequential_schedule(tmp_path):
    """The in-kernel hand-overs (stage flags between the Schur kernel and the factorisation beside it, the tiled factorisation's
    data-as-flag transfers above 64 cameras) are ordered by agent-scope stores awaited before a counter or flag moves, one release
    per stage and an acquire behind every flag: a stale line read once in thousands of steps would show as a different bit.  3000
    forced LM steps at 64 cameras (pipelined) and 600 at 130 (tiled factorisation) against RSBA_PIPELINE=0, bit for bit in the
    parameters and in every column of the iteration log."""
    files = []
    for pipe, steps in (("1", "3000"), ("0", "3000")):
        # (the same segments in both schedules — the sequential one cuts four per CU by default, the pipelined one eight — so that
        #  both add in the same order)
        # (the pipelined run with every step's head launched ahead on the device's decision, RSBA_LAUNCH_AHEAD: 3000 hand-overs of
        #  the decision block to a factorisation that is already resident)
        env = dict(os.environ, RSBA_PIPELINE=pipe, SWITCH_FORCED_STEPS=steps, RSBA_SEG_PER_CU="8", RSBA_LAUNCH_AHEAD=pipe)
        f = str(tmp_path / ("pipe%s.npz" % pipe))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "switch_worker.py"), "--dump", f, "c64_long"], env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        assert "stalled" not in out.stderr, out.stderr[-2000:]
        files.append(f)
    for rep in ("a", "b"):
        env = dict(os.environ, SWITCH_FORCED_STEPS="600")
        f = str(tmp_path / ("tiled%s.npz" % rep))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "switch_worker.py"), "--dump", f, "c130_long"], env=env, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        assert "stalled" not in out.stderr, out.stderr[-2000:]
        files.append(f)
    import numpy as np
    for fa, fb in ((files[0], files[1]), (files[2], files[3])):
        a, b = np.load(fa), np.load(fb)
        for name in a.files:
            assert a[name].shape == b[name].shape and np.array_equal(a[name], b[name]), name
