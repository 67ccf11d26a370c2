"""Multi-GPU plumbing: point-block sharding and RCCL bootstrap through torch.distributed.

The path shards by point block: every rank holds all cameras and a contiguous range of
points with all their observations.  Per LM iteration the ranks exchange exactly three things:

  1. sum all-reduce of the packed reduced camera system
         [ S (6C x 6C, camera-unscaled, undamped; the point blocks are already damped locally)
         | g_c = J_c' r | corr = -sum W V^-1 g_p | diag(J_c' J_c) | cost, |X|^2, failed point blocks ]
  2. max all-reduce of max|g_p|
  3. sum all-reduce of the candidate scalars [model cost change, candidate cost, |delta_p|^2, |X+delta|^2, sum r^2]

after (1)+(2) every rank Jacobi-scales, damps and factors the identical 6C x 6C system, so the camera step
needs no broadcast, and after (3) every rank takes the identical accept/reject decision.
"""
import ctypes


def shard_range(num_points, rank, world_size):
    """Contiguous, nearly equal point ranges; with k views per point these are observation-balanced."""
    lo = (num_points * rank) // world_size
    hi = (num_points * (rank + 1)) // world_size
    return lo, hi


def payload_size(num_cameras):
    nc = 6 * num_cameras
    return nc * nc + 3 * nc + 8


def broadcast_unique_id(dist, capi, rank):
    """ncclGetUniqueId on rank 0, shipped to everyone through the already-initialised process group."""
    box = [capi.comm_unique_id() if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return ctypes.create_string_buffer(box[0], 128)
