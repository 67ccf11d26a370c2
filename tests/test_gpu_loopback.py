"""The multi-rank schedule on ONE GPU: N point shards = N solver objects of this process, each driven by its own host thread, their
collectives summed over the group instead of sent through ncclAllReduce (csrc/ba_comm.hpp, rsba_comm_loopback_id).

Everything of the N > 1 path except the RCCL call itself runs here: the sharded upload (all cameras, a contiguous range of the
points: realsensecalibration_amd/distributed.py), the sequential multi-GPU schedule with its three collectives per LM step,
the end-of-run gradient collectives, the stall flag summed over the ranks and the fallback every rank then takes.  The whole
problem is what the ORACLE solves, unsharded; the reference itself is single-threaded and has no counterpart
(Main_Calibration/bundle_adjustment_manager.cpp:90-92, Test1_BundleAdjustment/main.cpp:82-87).  What must hold:
  * the stitched solution equals the oracle's of the whole problem like a single-GPU solve does (trajectory, costs, raw
    parameters per block to 1e-6, RMS to 1e-4 px);
  * every rank holds bit-identical camera blocks and took bit-identical decisions (identical iteration logs);
  * the communicator really spans N ranks.
BASELINE.json's configs[3] and configs[4] — 64 cameras x 1M points and 256 cameras x 500k points with Huber loss, both 8-GPU jobs —
run here at their full size in 8 shards (three forced iterations bound the oracle's time).
"""
import os

import numpy as np
import pytest

import oracle_lib as ol
from realsensecalibration_amd import capi
from realsensecalibration_amd import distributed as rd
from realsensecalibration_amd import synthetic as syn

pytestmark = pytest.mark.gpu
FORCED = dict(function_tolerance=-1.0, parameter_tolerance=-1.0, gradient_tolerance=-1.0)


@pytest.fixture(scope="module", autouse=True)
def _lib():
    lib = capi.load()
    assert lib.rsba_device_count() > 0, "GPU tests need a HIP device; the product has no CPU path"
    return lib


def _threads():
    return max(1, min(len(os.sched_getaffinity(0)), 64))


def _block_rel(a, b, C):
    worst = 0.0
    for x, y in ((a[:6 * C].reshape(-1, 6), b[:6 * C].reshape(-1, 6)), (a[6 * C:].reshape(-1, 3), b[6 * C:].reshape(-1, 3))):
        worst = max(worst, (np.abs(x - y).max(axis=1) / np.maximum(np.abs(y).max(axis=1), 1e-12)).max())
    return worst


def _shards(C, P, k, seed, world, outlier_frac=0.0):
    return [syn.make_problem(C, P, k, seed, point_range=rd.shard_range(P, r, world), outlier_frac=outlier_frac) for r in range(world)]


def _sharded_solve_matches_oracle(oracle, whole, shards, huber=0.0, iter_cost_tol=1e-9, **optkw):
    """Runs the shards as a loopback group and the whole problem through the oracle; returns the ranks' results."""
    C, world = whole["C"], len(shards)
    ref, s_ref, log_ref = oracle.solve_points(whole, oracle.options(huber_delta=huber, num_threads=_threads(), **optkw))
    ranks = capi.solve_points_sharded_loopback(shards, dict(huber_delta=huber, **optkw))
    p0, s0, log0, _ = ranks[0]
    for r, (p, s, log, nranks) in enumerate(ranks):
        assert nranks == world
        # the camera system is replicated: identical bits and identical decisions on every rank
        assert np.array_equal(p[:6 * C], p0[:6 * C]), "rank %d's camera blocks differ from rank 0's" % r
        assert np.array_equal(log[:, :8], log0[:, :8]), "rank %d's iteration log differs from rank 0's" % r
        assert (s.termination_type, s.stop_reason, s.num_iterations, s.final_cost) == (s0.termination_type, s0.stop_reason, s0.num_iterations, s0.final_cost)
    got = np.concatenate([p0[:6 * C]] + [p[6 * C:] for p, _, _, _ in ranks])
    assert got.shape == ref.shape
    assert (s0.termination_type, s0.stop_reason, s0.num_iterations) == (s_ref.termination, s_ref.stop_reason, s_ref.num_iterations)
    assert np.array_equal(log0[:, 7], log_ref[:, 7])
    assert np.abs(log0[:, 1] - log_ref[:, 1]).max() / log_ref[:, 1].max() < iter_cost_tol
    assert abs(s0.final_cost - s_ref.final_cost) < 1e-9 * s_ref.final_cost
    assert np.allclose(log0[:, 6], log_ref[:, 6], rtol=1e-6)
    rel = _block_rel(got, ref, C)
    assert rel < 1e-6, "raw parameters differ by %.2e" % rel
    ss_ref = oracle.points_cost(whole, ref, num_threads=_threads())[1]
    ss_got = oracle.points_cost(whole, got, num_threads=_threads())[1]
    assert abs(np.sqrt(ss_ref / (2 * whole["N"])) - np.sqrt(ss_got / (2 * whole["N"]))) < 1e-4
    return ranks


@pytest.mark.parametrize("world", [2, 8])
def test_config2_sharded(oracle, world):
    """BASELINE.json configs[1] (8 cameras x 10k points, 80k observations) in 2 and in 8 shards, solved to convergence."""
    C, P, k, seed, outl, huber = syn.CONFIGS["cfg2"]
    whole = syn.make_config("cfg2")
    ranks = _sharded_solve_matches_oracle(oracle, whole, _shards(C, P, k, seed, world))
    assert ranks[0][1].num_iterations >= 3


@pytest.mark.parametrize("C,P,k,huber,world", [(24, 4000, 8, 0.0, 2), (64, 6000, 12, 1.0, 3), (40, 5000, 9, 0.0, 5), (70, 3000, 10, 1.0, 4)])
def test_sharded_solves_match_the_unsharded_oracle(oracle, C, P, k, huber, world):
    """Other shapes: uneven shard sizes (3 and 5 ranks), the robust loss, the several-workgroup factorisation (40, 64 cameras)
    and the tiled one above 64 cameras (70), all to convergence."""
    outl = 0.05 if huber else 0.0
    whole = syn.make_problem(C, P, k, 400 + C, outlier_frac=outl)
    _sharded_solve_matches_oracle(oracle, whole, _shards(C, P, k, 400 + C, world, outl), huber=huber)


@pytest.mark.parametrize("mode", ["1", "0"])
@pytest.mark.parametrize("C,P,k,world", [(40, 5000, 9, 2), (64, 6000, 12, 3)])
def test_pipelined_multi_gpu_schedule_sharded(oracle, C, P, k, world, mode):
    """The pipelined multi-GPU schedule (the default with a communicator since round 4; RSBA_PIPELINE_MG=0: the sequential one): the
    factorisation launched ahead and gated stage by stage on flags published behind each stage's all-reduce (the group's row slab
    of S, read transposed), the candidate's sums all-reduced behind the back-substitution, the decision taken by k_publish_result
    on every rank alike — with real shards, not a 1-rank communicator.  Both schedules, explicitly."""
    whole = syn.make_problem(C, P, k, 500 + C)
    os.environ["RSBA_PIPELINE_MG"] = mode
    try:
        _sharded_solve_matches_oracle(oracle, whole, _shards(C, P, k, 500 + C, world))
    finally:
        del os.environ["RSBA_PIPELINE_MG"]


@pytest.mark.parametrize("C,P,k,world", [(40, 4000, 9, 2), (70, 3000, 10, 3)])
def test_triangular_payload_adds_the_same_bits(C, P, k, world):
    """The all-reduce payload as lower triangle + vectors (the default above 64 cameras with several ranks, forced here at 40 as
    well: RSBA_TRI_PAYLOAD=1) against the full square (=0): S leaves the Schur kernel symmetric to the bit, so both must end in
    identical bits on every rank (sequential multi-GPU schedule: the pipelined one all-reduces row slabs)."""
    shards = _shards(C, P, k, 600 + C, world)
    res = {}
    os.environ["RSBA_PIPELINE_MG"] = "0"
    try:
        for tri in ("0", "1"):
            os.environ["RSBA_TRI_PAYLOAD"] = tri
            res[tri] = capi.solve_points_sharded_loopback(shards)
    finally:
        del os.environ["RSBA_TRI_PAYLOAD"], os.environ["RSBA_PIPELINE_MG"]
    for (pa, sa, la, _), (pb, sb, lb, _) in zip(res["0"], res["1"]):
        assert np.array_equal(pa, pb) and np.array_equal(la, lb) and sa.final_cost == sb.final_cost


@pytest.mark.parametrize("C,P,k,world", [(40, 1500, 7, 2), (70, 1200, 8, 2)])
def test_triangular_payload_with_the_atomic_schur_kernel(oracle, C, P, k, world):
    """schur_impl 0 (k_linearize_schur_ref: also what a shard with duplicate observations runs) fills only the UPPER block triangle
    of S.  The triangular payload (default above 64 cameras, forced at 40) must then be packed from the mirror elements — read as
    the lower triangle it carried the memset's zeros and every rank lost the coupling between cameras.
    Against the full-square payload (sums of atomics: tolerance, not bits) and against the oracle."""
    whole = syn.make_problem(C, P, k, 640 + C)
    shards = _shards(C, P, k, 640 + C, world)
    res = {}
    os.environ["RSBA_PIPELINE_MG"] = "0"
    try:
        for tri in ("0", "1"):
            os.environ["RSBA_TRI_PAYLOAD"] = tri
            res[tri] = capi.solve_points_sharded_loopback(shards, dict(schur_impl=0))
    finally:
        del os.environ["RSBA_TRI_PAYLOAD"], os.environ["RSBA_PIPELINE_MG"]
    ref, s_ref, _ = oracle.solve_points(whole, oracle.options(num_threads=_threads()))
    for tri in ("0", "1"):
        got = np.concatenate([res[tri][0][0][:6 * C]] + [p[6 * C:] for p, _, _, _ in res[tri]])
        assert res[tri][0][1].num_iterations == s_ref.num_iterations, tri
        assert _block_rel(got, ref, C) < 1e-6, tri
        for p, s, log, nranks in res[tri]:
            assert nranks == world and np.array_equal(p[:6 * C], res[tri][0][0][:6 * C])


def test_a_stall_on_one_rank_is_everybodys_stall(oracle, capfd):
    """Step 2 of rank 1 reports a stalled factorisation (test hook RSBA_TEST_STALL_STEP / _RANK: what an in-kernel wait that ran
    out of its budget leaves in the result block).  The flag travels in the candidate's sum all-reduce, so BOTH ranks repeat the
    step with the one-workgroup factorisation and keep issuing the same collectives; the solve ends where the oracle's does."""
    C, P, k, seed = 40, 5000, 9, 77
    whole = syn.make_problem(C, P, k, seed)
    os.environ["RSBA_TEST_STALL_STEP"], os.environ["RSBA_TEST_STALL_RANK"] = "2", "1"
    try:
        _sharded_solve_matches_oracle(oracle, whole, _shards(C, P, k, seed, 2))
    finally:
        del os.environ["RSBA_TEST_STALL_STEP"], os.environ["RSBA_TEST_STALL_RANK"]
    err = capfd.readouterr().err
    assert err.count("multi-workgroup Cholesky stalled; using one workgroup") == 2, err


def test_config4_full_size_in_eight_shards(oracle):
    """BASELINE.json configs[3] at its full size: 64 cameras x 1M points, 20M observations, 8 ranks of 125k points."""
    C, P, k, seed, outl, huber = syn.CONFIGS["cfg4"]
    whole = syn.make_config("cfg4")
    assert (whole["C"], whole["P"], whole["N"]) == (64, 1_000_000, 20_000_000)
    shards = _shards(C, P, k, seed, 8)
    assert [s["P"] for s in shards] == [125_000] * 8
    ranks = _sharded_solve_matches_oracle(oracle, whole, shards, max_num_iterations=3, **FORCED)
    assert ranks[0][1].num_iterations == 3


def test_config5_full_size_in_eight_shards(oracle):
    """BASELINE.json configs[4] at its full size: 256 cameras x 500k points, 10M observations, Huber delta = 1 px with 5 %
    outliers, 8 ranks of 62.5k points; every rank factors the 1536 x 1536 sum with the persistent tiled Cholesky."""
    C, P, k, seed, outl, huber = syn.CONFIGS["cfg5"]
    whole = syn.make_config("cfg5")
    assert (whole["C"], whole["P"], whole["N"], huber, outl) == (256, 500_000, 10_000_000, 1.0, 0.05)
    shards = _shards(C, P, k, seed, 8, outl)
    ranks = _sharded_solve_matches_oracle(oracle, whole, shards, huber=huber, max_num_iterations=3, **FORCED)
    assert ranks[0][1].num_iterations == 3


def test_max_solver_time_with_several_ranks_stops_every_rank_on_the_same_iteration():
    """Solver::Options::max_solver_time_in_seconds with world_size > 1 (round 6; refused until then): rank 0's clock, as of the launch
    of a step, rides through that step's all-reduce of the candidate scalars (RES_TIME_UP), so every rank takes the same decision.
    A budget of zero: every rank returns the start untouched after iteration 0 (NO_CONVERGENCE, 'maximum solver time'), as one rank
    does (tests/test_gpu_parity.py::test_max_solver_time_ends_the_run_like_ceres); a generous budget changes nothing, bit for bit."""
    C, P, k, world = 24, 3000, 8, 3
    shards = _shards(C, P, k, 17, world)
    free = capi.solve_points_sharded_loopback(shards, dict(max_num_iterations=6, **FORCED))
    zero = capi.solve_points_sharded_loopback(shards, dict(max_num_iterations=6, max_solver_time_in_seconds=0.0, **FORCED))
    for r, (p, s, log, nranks) in enumerate(zero):
        assert nranks == world
        assert (s.num_iterations, s.termination_type, s.stop_reason) == (0, capi.NO_CONVERGENCE, 8), (r, s.num_iterations, s.stop_reason)
        assert np.array_equal(p, shards[r]["params"]) and s.final_cost == s.initial_cost
    ample = capi.solve_points_sharded_loopback(shards, dict(max_num_iterations=6, max_solver_time_in_seconds=1e6, **FORCED))
    for (p, s, log, _), (q, t, log2, _) in zip(ample, free):
        assert s.num_iterations == t.num_iterations == 6 and np.array_equal(p, q) and np.array_equal(log[:, :8], log2[:, :8])
