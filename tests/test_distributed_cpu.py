"""world_size = 2 over gloo, on CPU: the point-block sharding protocol of the multi-GPU path.

Each rank builds ITS shard of a synthetic problem (same generator bench.py uses), forms the per-rank payload the
HIP kernels produce — from the oracle's AutoDiff Jacobian blocks — all-reduces it exactly as the solver does
over RCCL, finishes the reduced system redundantly, back-substitutes its own points and all-reduces the
candidate scalars.  The result must equal the oracle's single-process step on the whole problem.
"""
import os
import sys

import numpy as np
import pytest

import oracle_lib as ol
from realsensecalibration_amd import distributed as rd
from realsensecalibration_amd import synthetic as syn

torch = pytest.importorskip("torch")
import torch.distributed as dist  # noqa: E402
import torch.multiprocessing as mp  # noqa: E402

C, P, K, SEED, RADIUS = 6, 240, 4, 77, 1e3
LO, HI = 1e-6, 1e32


def _rank_step(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = ol.load(build=False)
    lo, hi = rd.shard_range(P, rank, world)
    prob = syn.make_problem(C, P, K, SEED, point_range=(lo, hi))
    nc, Pl = 6 * C, prob["P"]
    cams, pts = prob["params"][:nc].reshape(C, 6), prob["params"][nc:].reshape(Pl, 3)
    intr = prob["intr"].reshape(C, 4)
    # ---- local linearisation: the payload of k_point_pass + k_schur_tiles
    S, gc, corr, diagU = np.zeros((nc, nc)), np.zeros(nc), np.zeros(nc), np.zeros(nc)
    cost = xn2 = gmax = 0.0
    per_point = []
    for j in range(Pl):
        sel = np.nonzero(prob["pt_idx"] == j)[0]
        V, gp, rows = np.zeros((3, 3)), np.zeros(3), []
        for i in sel:
            c = prob["cam_idx"][i]
            r, jc, jp = o.point_residual_jacobian(cams[c], pts[j], intr[c], prob["obs"][2 * i:2 * i + 2])
            cost += r @ r
            V += jp.T @ jp
            gp += jp.T @ r
            sl = slice(6 * c, 6 * c + 6)
            S[sl, sl] += jc.T @ jc
            gc[sl] += jc.T @ r
            diagU[sl] += np.diag(jc.T @ jc)
            rows.append((c, r, jc, jp))
        sp = 1.0 / (1.0 + np.sqrt(np.diag(V)))          # iteration-0 Jacobi scale of the point columns
        Vs = np.diag(sp) @ V @ np.diag(sp)
        M = Vs + np.diag(np.clip(np.diag(Vs), LO, HI) / RADIUS)
        Vi = np.diag(sp) @ np.linalg.inv(M) @ np.diag(sp)   # effective inverse in unscaled coordinates
        for (ca, ra, jca, jpa) in rows:
            Wa = jca.T @ jpa
            corr[6 * ca:6 * ca + 6] -= Wa @ Vi @ gp
            for (cb, rb, jcb, jpb) in rows:
                S[6 * ca:6 * ca + 6, 6 * cb:6 * cb + 6] -= Wa @ Vi @ (jcb.T @ jpb).T
        xn2 += pts[j] @ pts[j]
        gmax = max(gmax, np.abs(gp).max())
        per_point.append((V, gp, Vi, rows))
    payload = np.concatenate([S.ravel(), gc, corr, diagU, [cost, xn2, 0.0, 0, 0, 0, 0, 0]])
    assert payload.size == rd.payload_size(C)
    t = torch.from_numpy(payload)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)                         # collective 1
    g = torch.tensor([gmax], dtype=torch.float64)
    dist.all_reduce(g, op=dist.ReduceOp.MAX)                         # collective 2
    S, gc, corr, diagU = payload[:nc * nc].reshape(nc, nc), payload[nc * nc:nc * nc + nc], payload[nc * nc + nc:nc * nc + 2 * nc], payload[nc * nc + 2 * nc:nc * nc + 3 * nc]
    # ---- every rank: scale, damp, factor the identical system (k_reduced_system_solve)
    sc = 1.0 / (1.0 + np.sqrt(diagU))
    Ss = S * np.outer(sc, sc) + np.diag(np.clip(sc * sc * diagU, LO, HI) / RADIUS)
    y = np.linalg.solve(Ss, sc * (gc + corr))
    dcam = -sc * y
    # ---- local back-substitution + candidate (k_backsub_candidate)
    mcc = cost_c = dp2 = 0.0
    dpts = np.zeros((Pl, 3))
    cams_c = cams + dcam.reshape(C, 6)
    for j, (V, gp, Vi, rows) in enumerate(per_point):
        b = np.zeros(3)
        a1 = a2 = 0.0
        for (c, r, jc, jp) in rows:
            e = jc @ dcam[6 * c:6 * c + 6]
            b += jp.T @ e
            a1 += e @ r
            a2 += e @ e
        dp = -Vi @ (gp + b)
        dpts[j] = dp
        mcc -= a1 + dp @ gp + 0.5 * a2 + dp @ b + 0.5 * dp @ V @ dp
        dp2 += dp @ dp
        sel = np.nonzero(prob["pt_idx"] == j)[0]
        for i in sel:
            c = prob["cam_idx"][i]
            r, _, _ = o.point_residual_jacobian(cams_c[c], pts[j] + dp, intr[c], prob["obs"][2 * i:2 * i + 2])
            cost_c += r @ r
    small = torch.tensor([mcc, cost_c, dp2], dtype=torch.float64)
    dist.all_reduce(small, op=dist.ReduceOp.SUM)                     # collective 3
    out[rank] = dict(lo=lo, hi=hi, dcam=dcam, dpts=dpts, cost=0.5 * payload[nc * nc + 3 * nc], gmax=float(max(g.item(), np.abs(gc).max())),
                     mcc=float(small[0]), cost_c=0.5 * float(small[1]), step2=float(small[2]) + dcam @ dcam)
    dist.destroy_process_group()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_sharded_step_equals_single_process_oracle(oracle):
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_rank_step, args=(world, _free_port(), out), nprocs=world, join=True)
    full = syn.make_problem(C, P, K, SEED)
    ref = oracle.points_linearize_and_step(full, full["params"], RADIUS)
    nc = 6 * C
    r0, r1 = out[0], out[1]
    # identical camera step on both ranks, equal to the unsharded one
    assert np.array_equal(r0["dcam"], r1["dcam"])
    scale = np.abs(ref["delta"]).max()
    assert np.abs(r0["dcam"] - ref["delta"][:nc]).max() < 1e-9 * scale
    for r in (r0, r1):
        want = ref["delta"][nc + 3 * r["lo"]:nc + 3 * r["hi"]].reshape(-1, 3)
        assert np.abs(r["dpts"] - want).max() < 1e-9 * scale
        assert abs(r["cost"] - ref["cost"]) < 1e-12 * ref["cost"]
        assert abs(r["mcc"] - ref["model_cost_change"]) < 1e-8 * abs(ref["model_cost_change"])
        assert abs(r["gmax"] - ref["gradient_max_norm"]) < 1e-12 * ref["gradient_max_norm"]
        assert abs(np.sqrt(r["step2"]) - np.linalg.norm(ref["delta"])) < 1e-9 * np.linalg.norm(ref["delta"])
    cand, _ = oracle.points_cost(full, full["params"] + ref["delta"])
    assert abs(r0["cost_c"] - cand) < 1e-7 * cand and r0["cost_c"] == r1["cost_c"]


def test_shard_ranges_cover_everything_once():
    for P_, w in ((100_000, 8), (7, 3), (10, 1), (5, 8)):
        seen = []
        for r in range(w):
            lo, hi = rd.shard_range(P_, r, w)
            seen += list(range(lo, hi))
        assert seen == list(range(P_))


def test_shards_reassemble_the_global_problem():
    full = syn.make_problem(C, P, K, SEED)
    obs, params = [], [full["params"][:6 * C]]
    for r in range(3):
        lo, hi = rd.shard_range(P, r, 3)
        sh = syn.make_problem(C, P, K, SEED, point_range=(lo, hi))
        assert np.array_equal(sh["params"][:6 * C], full["params"][:6 * C])  # cameras replicated
        obs.append(sh["obs"])
        params.append(sh["params"][6 * C:])
    assert np.array_equal(np.concatenate(obs), full["obs"])
    assert np.array_equal(np.concatenate(params), full["params"])
