"""GPU parity at the shapes BASELINE.json names (configs[2..4]) and the LM-ordering corners, through the C ABI.

configs[3] (64 cams x 1M points) and configs[4] (256 cams x 500k points, Huber) are 8-GPU jobs; what one MI355X runs
of them is a rank's shard — all cameras and a contiguous eighth of the points — and that shard is a bundle-adjustment
problem of its own, which the oracle solves on the host for comparison.  Three forced LM iterations (tolerances off)
bound the oracle's time; every camera block and EVERY point block is compared (1e-6 relative per block, BASELINE's
tolerance), as are the costs of every iterate and the accept / reject sequence.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle_lib as ol
from realsensecalibration_amd import capi
from realsensecalibration_amd import distributed as rd
from realsensecalibration_amd import synthetic as syn

pytestmark = pytest.mark.gpu
ROOT = ol.ROOT
FORCED = dict(function_tolerance=-1.0, parameter_tolerance=-1.0, gradient_tolerance=-1.0)


@pytest.fixture(scope="module", autouse=True)
def _lib():
    lib = capi.load()
    assert lib.rsba_device_count() > 0, "GPU tests need a HIP device; the product has no CPU path"
    return lib


def _block_rel(a, b, C):
    worst = 0.0
    for x, y in ((a[:6 * C].reshape(-1, 6), b[:6 * C].reshape(-1, 6)), (a[6 * C:].reshape(-1, 3), b[6 * C:].reshape(-1, 3))):
        worst = max(worst, (np.abs(x - y).max(axis=1) / np.maximum(np.abs(y).max(axis=1), 1e-12)).max())
    return worst


def _threads():
    return max(1, min(len(os.sched_getaffinity(0)), 64))


def _forced_iterations_match(oracle, prob, iters, huber=0.0):
    o_ref = oracle.options(max_num_iterations=iters, num_threads=_threads(), huber_delta=huber, **FORCED)
    ref, s_ref, log_ref = oracle.solve_points(prob, o_ref)
    got, s_got, log_got = capi.solve_points(prob, capi.default_options(max_num_iterations=iters, huber_delta=huber, **FORCED))
    assert s_got.num_iterations == s_ref.num_iterations == iters
    assert (s_got.termination_type, s_got.stop_reason) == (s_ref.termination, s_ref.stop_reason)
    assert np.array_equal(log_got[:, 7], log_ref[:, 7])                       # same accept / reject sequence
    assert np.abs(log_got[:, 1] - log_ref[:, 1]).max() / log_ref[:, 1].max() < 1e-9   # every iterate's cost
    assert abs(s_got.final_cost - s_ref.final_cost) < 1e-10 * s_ref.final_cost
    assert _block_rel(got, ref, prob["C"]) < 1e-6                             # all camera blocks, all point blocks
    # the gradient the oracle logs for an accepted iteration is the NEW point's: the last row comes from the gradient
    # evaluation the run ends with (the iteration limit was reached right after an accepted step)
    assert np.abs(log_got[:, 3] - log_ref[:, 3]).max() < 1e-7 * log_ref[:, 3].max()
    assert np.allclose(log_got[:, 6], log_ref[:, 6], rtol=1e-6)               # trust-region radius
    ss_ref = oracle.points_cost(prob, ref, num_threads=_threads())[1]
    ss_got = oracle.points_cost(prob, got, num_threads=_threads())[1]
    assert abs(np.sqrt(ss_ref / (2 * prob["N"])) - np.sqrt(ss_got / (2 * prob["N"]))) < 1e-4   # RMS px
    return s_got


def test_config3_full_size_forced_iterations(oracle):
    """BASELINE.json configs[2], the benchmark's workload: 64 cams x 100k points, 2M observations, whole problem."""
    prob = syn.make_config("cfg3")
    assert (prob["C"], prob["P"], prob["N"]) == (64, 100_000, 2_000_000)
    _forced_iterations_match(oracle, prob, 3)


def test_config3_full_size_to_its_own_termination(oracle):
    """The benchmark's workload solved with the REFERENCE's tolerances (bundle_adjustment_manager.cpp:90-92 leaves Ceres' defaults:
    function 1e-6, gradient 1e-10, parameter 1e-8) to its own end, at full size: the STOP decision at 2M observations — which
    test fires, after how many iterations — the whole iteration log column by column, every camera and point block (1e-6) and the
    RMS (1e-4 px) against the oracle.  (Round 4 compared three forced iterations only.)"""
    prob = syn.make_config("cfg3")
    ref, s_ref, log_ref = oracle.solve_points(prob, oracle.options(num_threads=_threads()))
    got, s_got, log_got = capi.solve_points(prob)
    assert (s_got.termination_type, s_got.stop_reason, s_got.num_iterations) == (s_ref.termination, s_ref.stop_reason, s_ref.num_iterations)
    assert s_ref.termination == 0 and s_ref.num_iterations >= 3, "the run is expected to converge on a tolerance"
    assert (s_got.num_successful_steps, s_got.num_unsuccessful_steps) == (s_ref.num_successful_steps, s_ref.num_unsuccessful_steps)
    _full_log_matches(log_got, log_ref)
    assert abs(s_got.final_cost - s_ref.final_cost) < 1e-10 * s_ref.final_cost
    assert _block_rel(got, ref, prob["C"]) < 1e-6
    ss_ref = oracle.points_cost(prob, ref, num_threads=_threads())[1]
    ss_got = oracle.points_cost(prob, got, num_threads=_threads())[1]
    assert abs(np.sqrt(ss_ref / (2 * prob["N"])) - np.sqrt(ss_got / (2 * prob["N"]))) < 1e-4


def test_config4_shard_forced_iterations(oracle):
    """BASELINE.json configs[3]: 64 cams x 1M points over 8 GPUs -> rank 3's shard, 125k points, 2.5M observations."""
    C, P, k, seed, outl, huber = syn.CONFIGS["cfg4"]
    prob = syn.make_config("cfg4", point_range=rd.shard_range(P, 3, 8))
    assert (prob["C"], prob["P"], prob["N"]) == (64, 125_000, 2_500_000)
    _forced_iterations_match(oracle, prob, 3)


def test_config5_shard_huber_forced_iterations(oracle):
    """BASELINE.json configs[4]: 256 cams x 500k points, Huber delta = 1 px with 5 % outliers, over 8 GPUs -> rank 5's
    shard: 62 500 points, 1.25M observations, the 1536 x 1536 reduced system (persistent tiled Cholesky + multi-workgroup
    back-substitution), corrector as Ceres applies it (corrector.cc)."""
    C, P, k, seed, outl, huber = syn.CONFIGS["cfg5"]
    prob = syn.make_config("cfg5", point_range=rd.shard_range(P, 5, 8))
    assert (prob["C"], prob["P"], prob["N"], huber, outl) == (256, 62_500, 1_250_000, 1.0, 0.05)
    _forced_iterations_match(oracle, prob, 3, huber=huber)


def test_config4_whole_problem_on_one_gpu(oracle):
    """BASELINE.json configs[3] UNSHARDED: 64 cams x 1M points, 20M observations fit one MI355X (288 GB); the same three forced
    iterations against the oracle, every block compared.  (In eight shards: tests/test_gpu_loopback.py.)"""
    prob = syn.make_config("cfg4")
    assert (prob["C"], prob["P"], prob["N"]) == (64, 1_000_000, 20_000_000)
    _forced_iterations_match(oracle, prob, 3)


def test_config5_whole_problem_on_one_gpu(oracle):
    """BASELINE.json configs[4] UNSHARDED: 256 cams x 500k points, 10M observations, Huber."""
    C, P, k, seed, outl, huber = syn.CONFIGS["cfg5"]
    prob = syn.make_config("cfg5")
    assert (prob["C"], prob["P"], prob["N"]) == (256, 500_000, 10_000_000)
    _forced_iterations_match(oracle, prob, 3, huber=huber)


def test_config5_shard_reduced_system_matches_oracle(oracle):
    """One linearisation of a (smaller) 256-camera Huber shard, stage by stage: S, rhs, step, model cost change."""
    C, P, k, seed, outl, huber = syn.CONFIGS["cfg5"]
    prob = syn.make_config("cfg5", point_range=(0, 6000))
    a = oracle.points_linearize_and_step(prob, prob["params"], 1e4, opts=oracle.options(huber_delta=huber, num_threads=_threads()))
    b = capi.points_linearize_and_step(prob, 1e4, capi.default_options(huber_delta=huber))
    assert a["solve_ok"] and b["solve_ok"]
    assert np.abs(b["S"] - a["S"]).max() < 1e-10 * np.abs(a["S"]).max()
    assert np.abs(b["rhs"] - a["rhs"]).max() < 1e-10 * np.abs(a["rhs"]).max()
    assert np.abs(b["delta"] - a["delta"]).max() < 1e-7 * np.abs(a["delta"]).max()
    assert abs(b["model_cost_change"] - a["model_cost_change"]) < 1e-8 * abs(a["model_cost_change"])


# ------------------------------------------------------------------ order of the convergence tests (Ceres 1.14)
def _full_log_matches(log_got, log_ref, tol=1e-6):
    assert log_got.shape == log_ref.shape
    assert np.array_equal(log_got[:, [0, 7]], log_ref[:, [0, 7]])
    for col, name in ((1, "cost"), (2, "cost_change"), (3, "gradient_max_norm"), (4, "step_norm"), (6, "trust_region_radius")):
        scale = np.abs(log_ref[:, col]).max()
        assert np.abs(log_got[:, col] - log_ref[:, col]).max() <= tol * scale, name
    assert np.abs(log_got[:, 5] - log_ref[:, 5]).max() < 1e-5, "relative_decrease"


@pytest.mark.parametrize("impl", [0, 1])
@pytest.mark.parametrize("C,P,k,stop_at", [(8, 1500, 6, 2), (33, 2500, 8, 1), (33, 2500, 8, 3)])
def test_gradient_tolerance_fires_after_an_accepted_step(oracle, impl, C, P, k, stop_at):
    """A noise-free problem (the optimum has zero residual, the gradient falls by orders of magnitude per iteration).  The
    gradient tolerance is put between the gradients of iterations stop_at - 1 and stop_at, the other tolerances are off:
    Ceres tests max |g| at the NEW point right after the accepted step `stop_at` and stops there, without another solve
    (trust_region_minimizer.cc; oracle/ba_oracle.hpp).  Same iteration count, same stop reason, same log — the last row
    carries the new point's cost and gradient."""
    prob = syn.make_problem(C, P, k, seed=31 + C, noise_px=0.0)
    free = dict(function_tolerance=-1.0, parameter_tolerance=-1.0)
    _, s_probe, log_probe = oracle.solve_points(prob, oracle.options(max_num_iterations=stop_at + 1, gradient_tolerance=-1.0, **free))
    g = log_probe[:, 3]
    assert np.all(log_probe[1:stop_at + 1, 7] == 3), "the probe's steps must be accepted"
    assert g[stop_at] < 0.1 * g[stop_at - 1]
    gtol = float(np.sqrt(g[stop_at] * g[stop_at - 1]))
    ref, s_ref, log_ref = oracle.solve_points(prob, oracle.options(gradient_tolerance=gtol, **free))
    assert (s_ref.termination, s_ref.stop_reason, s_ref.num_iterations) == (0, 1, stop_at)
    got, s_got, log_got = capi.solve_points(prob, capi.default_options(schur_impl=impl, gradient_tolerance=gtol, **free))
    assert (s_got.termination_type, s_got.stop_reason, s_got.num_iterations) == (capi.CONVERGENCE, 1, stop_at)
    assert (s_got.num_successful_steps, s_got.num_unsuccessful_steps) == (s_ref.num_successful_steps, s_ref.num_unsuccessful_steps)
    _full_log_matches(log_got, log_ref)
    assert log_got[-1, 3] <= gtol < log_got[-2, 3]
    assert abs(s_got.final_cost - s_ref.final_cost) <= 1e-9 * max(s_ref.final_cost, 1e-3 * s_ref.initial_cost)
    assert _block_rel(got, ref, C) < 1e-6


@pytest.mark.parametrize("impl", [0, 1])
def test_whole_log_matches_oracle_including_the_final_gradient(oracle, impl):
    """Every column of the iteration log against the oracle on a converging solve that ends on the function tolerance,
    and on a run cut by the iteration limit right after an accepted step (the last row's gradient is then evaluated
    without a solve)."""
    prob = syn.make_problem(16, 3000, 9, seed=7)
    ref, s_ref, log_ref = oracle.solve_points(prob)
    got, s_got, log_got = capi.solve_points(prob, capi.default_options(schur_impl=impl))
    assert (s_got.stop_reason, s_got.num_iterations) == (s_ref.stop_reason, s_ref.num_iterations)
    _full_log_matches(log_got, log_ref)
    ref, s_ref, log_ref = oracle.solve_points(prob, oracle.options(max_num_iterations=2))
    got, s_got, log_got = capi.solve_points(prob, capi.default_options(schur_impl=impl, max_num_iterations=2))
    assert (s_got.stop_reason, s_got.num_iterations) == (s_ref.stop_reason, s_ref.num_iterations) == (4, 2)
    assert log_ref[-1, 7] == 3
    _full_log_matches(log_got, log_ref)


def test_progress_table_has_ceres_columns(capfd):
    """minimizer_progress_to_stdout (bundle_adjustment_manager.cpp:92): Ceres' ten columns, one row per iteration."""
    prob = syn.make_problem(8, 800, 5, seed=3)
    _, s, log = capi.solve_points(prob, capi.default_options(minimizer_progress_to_stdout=1))
    capi.load()
    import ctypes
    ctypes.CDLL(None).fflush(None)
    out = capfd.readouterr().out
    lines = [ln for ln in out.splitlines() if ln.strip()]
    head = [i for i, ln in enumerate(lines) if ln.startswith("iter")]
    assert head, out
    assert lines[head[0]].split() == ["iter", "cost", "cost_change", "|gradient|", "|step|", "tr_ratio", "tr_radius", "ls_iter", "iter_time", "total_time"]
    rows = [ln.split() for ln in lines[head[0] + 1:head[0] + 2 + s.num_iterations]]
    assert len(rows) == s.num_iterations + 1 and all(len(r) == 10 for r in rows)
    assert [int(r[0]) for r in rows] == list(range(s.num_iterations + 1))
    assert abs(float(rows[-1][1]) - log[-1, 1]) < 1e-5 * log[-1, 1]
    assert [int(r[7]) for r in rows] == [0] + [1] * s.num_iterations
    tot = [float(r[9]) for r in rows]
    assert all(b >= a for a, b in zip(tot, tot[1:]))


# ------------------------------------------------------------------ two ranks over RCCL (needs two GPUs)
def test_two_rank_solve_matches_unsharded_oracle(oracle, tmp_path):
    """Two processes, one GPU each, points sharded by contiguous block, the reduced system all-reduced over RCCL: the
    trajectory must be the unsharded oracle's, both ranks must hold bit-identical camera blocks, and the communicator
    must span two ranks.  Skipped on a one-GPU box."""
    if capi.load().rsba_device_count() < 2:
        pytest.skip("needs two GPUs")
    out = tmp_path / "mg.json"
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import socket
    with socket.socket() as sk:   # a free ephemeral port, as bench.py picks one: a fixed one collides with a concurrent or stale run
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "mg_worker.py"), str(out)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    res = json.loads(out.read_text())
    assert res["rccl_nranks"] == [2, 2]
    assert res["camera_blocks_bitwise_equal"]
    for case in res["cases"]:
        assert case["same_iterations"] and case["same_stop_reason"] and case["same_accept_reject"], case
        assert case["max_block_rel"] < 1e-6 and case["cost_rel"] < 1e-9, case


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_two_processes_on_one_gpu_match_the_unsharded_oracle(oracle, tmp_path):
    """The same rank program, two REAL processes (torch.distributed.run, gloo for the plumbing), on whatever GPUs are visible —
    one is enough: the collectives go through shared memory (ShmComm) instead of ncclAllReduce, which refuses two ranks of a
    communicator on one device.  Everything else is the multi-process path as an 8-GPU node runs it: one process per rank, the
    id bootstrap through the process group, the sharded upload, three collectives per LM step, identical decisions."""
    out = tmp_path / "mg_shm.json"
    env = dict(os.environ, RSBA_MG_COMM="shm")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "mg_worker.py"), str(out)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    res = json.loads(out.read_text())
    assert res["rccl_nranks"] == [2, 2] and res["comm_kinds"] == ["shm", "shm"]
    assert res["camera_blocks_bitwise_equal"]
    for case in res["cases"]:
        assert case["same_iterations"] and case["same_stop_reason"] and case["same_accept_reject"], case
        assert case["max_block_rel"] < 1e-6 and case["cost_rel"] < 1e-9, case


def test_a_dead_rank_ends_the_job_instead_of_hanging_it():
    """Rank 1 exits behind set-up; rank 0's first collective finds nobody.  The communicator's bounded wait (4 s here) aborts the
    group and rsba_solver_run returns RSBA_ERR_COMM: the worker leaves with exit code 7 within seconds.  (Over RCCL the same
    path polls ncclCommGetAsyncError and calls ncclCommAbort: RcclComm::WaitStream — needs two GPUs to run.)"""
    import time
    env = dict(os.environ, RSBA_COMM_TIMEOUT_S="4")
    t0 = time.time()
    procs = []
    port = _free_port()
    for r in range(2):
        e = dict(env, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "mg_abort_worker.py")], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert procs[1].returncode == 0, outs[1][-2000:]
    assert procs[0].returncode == 7, (procs[0].returncode, outs[0][-2000:])
    assert time.time() - t0 < 120


def test_bench_launcher_two_ranks_on_one_gpu(tmp_path):
    """bench.py's own multi-rank launcher end to end on a one-GPU box: `--gpus 2 --comm shm` starts two fresh rank processes (children,
    never a re-exec), bootstraps the group, runs the preflight (iteration logs compared bit for bit across the ranks), the timed
    regions with barrier + max over ranks, and rank 0 prints ONE JSON line that says which schedule ran and what stalled."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--comm", "shm", "--config", "cfg2", "--steps", "4", "--warmup", "1",
                        "--preload", "0", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_nranks"] == 2 and line["comm"] == "shm" and line["steps"] == 4 and line["warmup"] == 1
    assert line["schedule"] == "sequential" and line["stalls"] == 0 and line["fallbacks"] == 0
    assert line["preflight"]["logs_bitwise_equal_across_ranks"] is True and line["preflight"]["stalls"] == 0
    assert line["value"] > 0 and line["scaling"] == "weak"


# ------------------------------------------------------------------ internal point order
@pytest.mark.parametrize("C,P,k,huber", [(24, 6000, 8, 0.0), (40, 9000, 10, 1.0), (70, 5000, 9, 0.0)])
def test_balanced_point_order_is_internal(oracle, C, P, k, huber, monkeypatch):
    """The tiled Schur kernel deals the points to its chunks so that every camera pair has about the same number of shared
    points per chunk (BalancedPointOrder).  The permutation must not be visible: same trajectory and same parameter
    blocks, in the caller's order, as with the file order (RSBA_BALANCE=0) and as the oracle; the stage-level step too."""
    prob = syn.make_problem(C, P, k, seed=90 + C, outlier_frac=0.05 if huber else 0.0)
    ref, s_ref, log_ref = oracle.solve_points(prob, oracle.options(huber_delta=huber, num_threads=_threads()))
    got, s_got, log_got = capi.solve_points(prob, capi.default_options(huber_delta=huber))
    a = capi.points_linearize_and_step(prob, 1e4, capi.default_options(huber_delta=huber))
    monkeypatch.setenv("RSBA_BALANCE", "0")
    plain, s_plain, log_plain = capi.solve_points(prob, capi.default_options(huber_delta=huber))
    b = capi.points_linearize_and_step(prob, 1e4, capi.default_options(huber_delta=huber))
    for x, sx, lx in ((got, s_got, log_got), (plain, s_plain, log_plain)):
        assert (sx.num_iterations, sx.stop_reason) == (s_ref.num_iterations, s_ref.stop_reason)
        assert np.array_equal(lx[:, 7], log_ref[:, 7])
        assert _block_rel(x, ref, C) < 1e-6
        assert abs(sx.final_cost - s_ref.final_cost) < 1e-9 * s_ref.final_cost
    assert _block_rel(got, plain, C) < 1e-8
    assert np.abs(a["delta"] - b["delta"]).max() < 1e-9 * np.abs(b["delta"]).max()
    assert np.abs(a["S"] - b["S"]).max() < 1e-11 * np.abs(b["S"]).max()


# ------------------------------------------------------------------ both several-workgroup factorisations of the reduced system
@pytest.mark.parametrize("diag", [1, 0] if os.environ.get("RSBA_TEST_EXPERIMENTAL") == "1" else [1])   # (0: -DRSBA_EXPERIMENTAL builds only)
@pytest.mark.parametrize("C,P,k,huber", [(33, 2500, 8, 0.0), (48, 3000, 9, 1.0), (64, 4000, 12, 0.0)])
def test_diagonal_and_round_robin_cholesky(oracle, C, P, k, huber, diag, monkeypatch):
    """ba_cholesky_diag.hpp (the default for 32 to 64 cameras: workgroup 0 keeps the chain of diagonal factorisations, the
    others own the rows below) and ba_cholesky_multi.hpp (RSBA_CHOL_DIAG=0: blocks dealt round-robin): same trajectory and
    blocks as the oracle, pipelined and sequential schedule bit for bit, two runs bit for bit."""
    monkeypatch.setenv("RSBA_CHOL_DIAG", str(diag))
    prob = syn.make_problem(C, P, k, seed=60 + C, outlier_frac=0.05 if huber else 0.0)
    ref, s_ref, log_ref = oracle.solve_points(prob, oracle.options(huber_delta=huber, num_threads=_threads()))
    got, s_got, log_got = capi.solve_points(prob, capi.default_options(huber_delta=huber))
    again, s_again, log_again = capi.solve_points(prob, capi.default_options(huber_delta=huber))
    assert (s_got.num_iterations, s_got.stop_reason) == (s_ref.num_iterations, s_ref.stop_reason)
    assert np.array_equal(log_got[:, 7], log_ref[:, 7])
    assert _block_rel(got, ref, C) < 1e-6
    assert abs(s_got.final_cost - s_ref.final_cost) < 1e-9 * s_ref.final_cost
    assert np.array_equal(got, again) and np.array_equal(log_got, log_again)
    monkeypatch.setenv("RSBA_PIPELINE", "0")
    monkeypatch.setenv("RSBA_SEG_TARGET", "8")   # (the pipelined default, through the same whole-chunk rounding)
    seq, s_seq, log_seq = capi.solve_points(prob, capi.default_options(huber_delta=huber))
    assert np.array_equal(got, seq) and np.array_equal(log_got, log_seq)
