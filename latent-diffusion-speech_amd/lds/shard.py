"""Utterance-batch sharding across the GPUs of one node (one process per GPU).

The sampler path is embarrassingly parallel over utterances (every op is per-sample, SURVEY.md
8e), so the only communication is moving inputs to the ranks and results back: an RCCL scatter of
the input shards and a gather of the results over xGMI through torch.distributed (backend "nccl"
on ROCm; "gloo" on CPU in the tests).  No collective runs inside the sampler loop."""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous, balanced split of n_items over `world` ranks -> (start, stop) of `rank`."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def _meta(full, rank, src, shape, dtype):
    """(shape, dtype) of the full batch on every rank; one object broadcast unless the caller supplied both."""
    if shape is not None and dtype is not None:
        return tuple(shape), dtype
    box = [(tuple(full.shape), full.dtype) if rank == src else None]
    dist.broadcast_object_list(box, src=src)
    return tuple(box[0][0]), box[0][1]


def scatter_batch(full, rank, world, src=0, shape=None, dtype=None, device=None, force=False):
    """Rank `src` holds `full` [N, ...]; every rank returns its shard [n_r, ...] (contiguous balanced split).
    One scatter: each rank receives only its own rows (42 MB of units per rank at B=16 x 512 frames, instead of the
    whole batch).  Ragged splits pad every shard to the largest one.  `shape` / `dtype` = those of the full batch when
    every rank knows them (saves the object broadcast); `device` = where non-src ranks allocate (default: full.device).
    `force` runs the collective even for a single rank (a one-GPU rehearsal of the RCCL call pattern)."""
    if world == 1 and not force:
        return full
    shp, dt = _meta(full, rank, src, shape, dtype)
    dev = device if device is not None else full.device
    n = shp[0]
    mx = -(-n // world)
    lo, hi = shard_range(n, rank, world)
    recv = torch.empty((mx,) + shp[1:], dtype=dt, device=dev)
    parts = None
    if rank == src:
        if full.dtype != dt or tuple(full.shape) != shp:
            raise ValueError(f"scatter_batch: full is {tuple(full.shape)} {full.dtype}, announced {shp} {dt}")
        parts = []
        for r in range(world):
            a, b = shard_range(n, r, world)
            p = full[a:b]
            if b - a < mx:                                 # ragged: pad to the common shard size
                p = torch.cat([p, p.new_zeros((mx - (b - a),) + shp[1:])])
            parts.append(p.contiguous())
    dist.scatter(recv, parts, src=src)
    return recv[: hi - lo].contiguous() if hi - lo < mx else recv


def gather_batch(local, rank, world, dst=0, sizes=None, force=False):
    """Concatenate per-rank results [n_r, ...] on rank `dst` (None elsewhere): one gather to `dst`.
    `sizes` = the per-rank n_r when the caller knows them (saves the object collective and its host synchronisation)."""
    if world == 1 and not force:
        return local
    if sizes is None:
        sizes = [None] * world
        dist.all_gather_object(sizes, int(local.shape[0]))
    mx = max(sizes)
    send = local.contiguous()
    if send.shape[0] < mx:
        send = torch.cat([send, send.new_zeros((mx - send.shape[0],) + tuple(send.shape[1:]))])
    outs = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, outs, dst=dst)
    if rank != dst:
        return None
    if len(set(sizes)) == 1:
        return torch.cat(outs)
    return torch.cat([o[:s] for o, s in zip(outs, sizes)])


def broadcast_state(state, rank, world, src=0, device=None, force=False):
    """Weights of rank `src` to every rank with ONE broadcast (north_star: "RCCL broadcast/gather"; 530 MB of UNet + front end, 56 MB of vocoder).
    state: an ordered {name: tensor} (torch tensors, all float32; e.g. nn.Module.state_dict()) with the SAME keys and shapes on every rank -- the
    values on ranks other than `src` are overwritten in place (build them under lds.init_weights.deferred()).  The tensors are flattened into one
    buffer on `device` (default: where the first tensor lives; RCCL needs the GPU), broadcast, and copied back.  Returns the number of bytes moved."""
    if world == 1 and not force:
        return 0
    items = [(k, v) for k, v in state.items() if torch.is_tensor(v) and v.dtype == torch.float32]
    if not items:
        return 0
    dev = torch.device(device) if device is not None else items[0][1].device
    n = sum(v.numel() for _, v in items)
    flat = torch.empty(n, dtype=torch.float32, device=dev)
    if rank == src:
        off = 0
        for _, v in items:
            flat[off:off + v.numel()].copy_(v.reshape(-1))
            off += v.numel()
    dist.broadcast(flat, src=src)
    if rank != src:
        off = 0
        with torch.no_grad():
            for _, v in items:
                v.copy_(flat[off:off + v.numel()].reshape(v.shape))
                off += v.numel()
    return 4 * n
