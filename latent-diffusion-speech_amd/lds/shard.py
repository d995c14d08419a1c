"""Utterance-batch sharding across the GPUs of one node (one process per GPU).

The sampler path is embarrassingly parallel over utterances (every op is per-sample, SURVEY.md
8e), so the only communication is moving inputs to the ranks and results back: RCCL
broadcast / scatter / gather over xGMI through torch.distributed (backend "nccl" on ROCm;
"gloo" on CPU in the tests).  No collective runs inside the sampler loop."""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous, balanced split of n_items over `world` ranks -> (start, stop) of `rank`."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def scatter_batch(full, rank, world, src=0, shape=None):
    """Rank `src` holds `full` [N, ...]; every rank returns its shard [n_r, ...].
    One broadcast of the batch (RCCL broadcast over xGMI on the GPU box) + a local slice: the same code path for
    even and ragged splits, and only collectives every backend implements.  `shape` = the full shape when every rank
    knows it (saves the object broadcast)."""
    if world == 1:
        return full
    if shape is None:
        box = [tuple(full.shape) if rank == src else None]
        dist.broadcast_object_list(box, src=src)
        shape = box[0]
    shp = tuple(shape)
    buf = full.contiguous() if rank == src else torch.empty(shp, dtype=torch.float32, device=full.device)
    dist.broadcast(buf, src=src)
    lo, hi = shard_range(shp[0], rank, world)
    return buf[lo:hi].contiguous()


def gather_batch(local, rank, world, dst=0, sizes=None):
    """Concatenate per-rank results [n_r, ...] on rank `dst` (equal n_r -> all_gather_into_tensor).
    `sizes` = the per-rank n_r when the caller knows them (saves the object collective and its host synchronisation)."""
    if world == 1:
        return local
    if sizes is None:
        sizes = [None] * world
        dist.all_gather_object(sizes, int(local.shape[0]))
    if len(set(sizes)) == 1:
        out = torch.empty((sum(sizes),) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous())
        return out if rank == dst else None
    mx = max(sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    outs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad)
    if rank != dst:
        return None
    return torch.cat([o[:s] for o, s in zip(outs, sizes)])
