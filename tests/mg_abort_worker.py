"""Rank program of tests/test_gpu_configs.py::test_a_dead_rank_ends_the_job_instead_of_hanging_it: rank 1 leaves right after
set-up, rank 0 goes on into the first collective of its run.  The bounded waits of the communicator (csrc/ba_comm.hpp; shared-memory
group here, RSBA_COMM_TIMEOUT_S=4) must turn that into an error return — RSBA_ERR_COMM, exit code 7 — not into a hang."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    import torch
    import torch.distributed as dist
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    from realsensecalibration_amd import capi
    from realsensecalibration_amd import distributed as rd
    from realsensecalibration_amd import synthetic as syn
    box = ["abort_%d" % os.getppid() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    uid = ctypes.create_string_buffer(capi.comm_shm_id(box[0]), 128)
    C, P, k, seed = 24, 3000, 8, 5
    shard = syn.make_problem(C, P, k, seed, point_range=rd.shard_range(P, rank, world))
    o = capi.default_options(device=local % torch.cuda.device_count(), rank=rank, world_size=world)
    o.comm_unique_id = ctypes.cast(uid, ctypes.c_void_p)
    problem = capi.Problem.points(shard)
    sv = capi.Solver(problem, o)   # (both ranks meet in the set-up collectives)
    dist.barrier()
    if rank == 1:
        os._exit(0)                # gone without a word
    try:
        sv.run()
    except capi.RsbaError as e:
        print("rank 0: %s" % e, flush=True)
        os._exit(7 if e.code == capi.ERR_COMM else 8)
    os._exit(9)                    # the run must not succeed with half of the points missing


if __name__ == "__main__":
    main()
