"""Rank program of tests/test_gpu_configs.py::test_two_rank_solve_matches_unsharded_oracle (torch.distributed.run starts
one per GPU).  Each rank builds ITS shard of a synthetic point-model problem (all cameras, a contiguous block of the
points), solves it through the C ABI with a shared RCCL communicator, and rank 0 compares the stitched result with the
oracle's solve of the WHOLE problem.  Test infrastructure: the oracle is the checker."""
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def main():
    out_path = sys.argv[1]
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    import torch.distributed as dist
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)   # plumbing only: ships the RCCL id and the results
    from realsensecalibration_amd import capi
    from realsensecalibration_amd import distributed as rd
    from realsensecalibration_amd import synthetic as syn
    # RSBA_MG_COMM=shm: the ranks are processes SHARING the visible GPUs, their collectives staged through shared memory (ShmComm,
    # csrc/ba_comm.hpp) — everything of this worker but the RCCL call itself, on a one-GPU box
    shm = os.environ.get("RSBA_MG_COMM") == "shm"
    if shm:
        import torch
        local = local % torch.cuda.device_count()
        box = ["mgw_%d_%s" % (os.getppid(), os.path.basename(out_path).replace(".", "_")) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        uid = ctypes.create_string_buffer(capi.comm_shm_id(box[0]), 128)
    else:
        uid = rd.broadcast_unique_id(dist, capi, rank)
    cases, nranks, cams_equal = [], None, True
    for (C, P, k, seed, huber, outl) in [(24, 4000, 8, 11, 0.0, 0.0), (64, 6000, 12, 12, 1.0, 0.05), (8, 3000, 6, 13, 0.0, 0.0)]:
        lo, hi = rd.shard_range(P, rank, world)
        shard = syn.make_problem(C, P, k, seed, point_range=(lo, hi), outlier_frac=outl)
        o = capi.default_options(device=local, rank=rank, world_size=world, huber_delta=huber)
        o.comm_unique_id = ctypes.cast(uid, ctypes.c_void_p)
        problem = capi.Problem.points(shard)
        sv = capi.Solver(problem, o)
        nranks = sv.comm_nranks()
        kind = sv.schedule_info()["comm_kind"]
        s = sv.run()
        sv.download()
        log = sv.iterations()
        mine = dict(params=np.array(problem.params, copy=True), iters=int(s.num_iterations), stop=int(s.stop_reason), cost=float(s.final_cost),
                    log=log, nranks=nranks, lo=lo, hi=hi, kind=kind)
        sv.close()
        problem.close()
        box = [None] * world if rank == 0 else None
        dist.gather_object(mine, box, dst=0)
        if rank == 0:
            import oracle_lib
            whole = syn.make_problem(C, P, k, seed, outlier_frac=outl)
            orc = oracle_lib.load()
            ref, s_ref, log_ref = orc.solve_points(whole, orc.options(huber_delta=huber, num_threads=min(len(os.sched_getaffinity(0)), 32)))
            got = np.concatenate([box[0]["params"][:6 * C]] + [b["params"][6 * C:] for b in box])
            for b in box[1:]:
                cams_equal = cams_equal and bool(np.array_equal(b["params"][:6 * C], box[0]["params"][:6 * C]))
                cams_equal = cams_equal and bool(np.array_equal(b["log"], box[0]["log"]))
            worst = 0.0
            for x, y in ((got[:6 * C].reshape(-1, 6), ref[:6 * C].reshape(-1, 6)), (got[6 * C:].reshape(-1, 3), ref[6 * C:].reshape(-1, 3))):
                worst = max(worst, float((np.abs(x - y).max(axis=1) / np.maximum(np.abs(y).max(axis=1), 1e-12)).max()))
            cases.append(dict(C=C, P=P, same_iterations=box[0]["iters"] == s_ref.num_iterations, same_stop_reason=box[0]["stop"] == s_ref.stop_reason,
                              same_accept_reject=bool(np.array_equal(box[0]["log"][:, 7], log_ref[:, 7])), max_block_rel=worst,
                              cost_rel=abs(box[0]["cost"] - s_ref.final_cost) / s_ref.final_cost,
                              nranks=[b["nranks"] for b in box], comm_kinds=[b["kind"] for b in box]))
    if rank == 0:
        json.dump(dict(rccl_nranks=cases[-1]["nranks"], comm_kinds=cases[-1]["comm_kinds"], camera_blocks_bitwise_equal=cams_equal, cases=cases), open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
