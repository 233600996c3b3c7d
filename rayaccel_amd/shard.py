"""Ray-batch sharding across the GPUs of one node (one process per GPU).

The path shards naturally: rays never interact and the scene is read-only (Scene.cpp:342-346
CL_MEM_READ_ONLY), so each rank traces a contiguous slice with NO data-path collective.  The only
exchange that can follow is an all-gather of the 16-byte Result records over RCCL/xGMI, needed only when a
GPU-side consumer wants every hit (north_star: "only when the renderer oversubscribes one GPU").
"""
import numpy as np


def shard_range(count, rank, world):
    """Contiguous slice [begin, end) of a `count`-ray batch owned by `rank`; remainders go to the low ranks."""
    base, extra = divmod(int(count), int(world))
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def allgather_results(local, count, world, dist, torch):
    """All-gather ragged Result shards (N_r x 4 float32/uint32 words) into the full [count,4] batch on every
    rank.  Uses one all_gather_into_tensor over equal-size padded shards (a single large message per
    rank: ring collectives over xGMI are per-link bound, so fewer and larger beats many small)."""
    per = -(-count // world)
    padded = torch.zeros((per, 4), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    out = torch.empty((world * per, 4), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded)
    pieces = []
    for r in range(world):
        b, e = shard_range(count, r, world)
        pieces.append(out[r * per: r * per + (e - b)])
    return torch.cat(pieces, 0)
