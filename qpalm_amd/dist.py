"""Multi-GPU batch sharding.

QPs are independent units: a batch is partitioned over the ranks of one node (one process per GPU,
torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests), every rank
solves its shard with no data-path collective, and ONE gather at the end brings solution.x,
solution.y and the info records to rank 0.  A single QP is never split across GPUs.
"""
import numpy as np

# QPALMInfo (include/types.h:76-95) as one fp64 row per QP; the status string is derived from status_val on arrival
INFO_FIELDS = ("iter", "iter_out", "status_val", "pri_res_norm", "dua_res_norm", "dua2_res_norm", "objective", "dual_objective",
               "setup_time", "solve_time", "run_time")


class _DeviceArray:
    """__cuda_array_interface__ wrapper of a device array of the batch (fp64, C order)."""

    def __init__(self, ptr, shape):
        self.__cuda_array_interface__ = {"shape": tuple(int(v) for v in shape), "typestr": "<f8", "data": (int(ptr), False), "version": 3}


def device_view(batch, name, shape, device):
    """Zero-copy torch view of a named HBM array of the batch (e.g. "solution_x", [B][n]): what the
    RCCL gather of bench.py sends, no staging through the host."""
    import torch
    ptr, nbytes = batch.device_ptr(name)
    assert int(np.prod(shape)) * 8 <= nbytes, (name, shape, nbytes)
    return torch.as_tensor(_DeviceArray(ptr, shape), device=device)


def shard_indices(nqp, world, rank, sizes=None):
    """Round-robin assignment: per-instance cost varies by >10x inside one problem family
    (simulations/results/journal_paper/randomMPCsequential2.tex:32-61), so contiguous shards are
    avoided; rank r owns QPs r, r + world, ...  With `sizes` (one (n, m) per QP: a mixed-size batch, e.g. a directory of
    QPS files) the round robin runs over the QPs sorted by size ("size-sorted round-robin"), so
    that every rank gets its share of the large ones; equal sizes give the plain round robin."""
    if sizes is None:
        return np.arange(rank, nqp, world)
    order = sorted(range(nqp), key=lambda k: (int(sizes[k][0]), int(sizes[k][1]), k))
    return np.array(order[rank::world], dtype=np.int64)


def info_matrix(batch):
    """[B][len(INFO_FIELDS)] fp64: the QPALMInfo records of the batch (one device-to-host copy), ready for the gather"""
    from .capi import Info
    rec = np.frombuffer(batch.infos(), dtype=np.dtype(Info), count=batch.B)   # structured view of the ctypes array: no Python loop
    out = np.empty((batch.B, len(INFO_FIELDS)))
    for j, k in enumerate(INFO_FIELDS):
        out[:, j] = rec[k]
    return out


pack_info = info_matrix


def _solve_local(problems, make_batch):
    """this rank's QPs, as size buckets when the sizes differ (members of a bucket are padded to its largest member on
    the device only); returns per-QP x, y (own lengths) and the info rows"""
    from .qps import bucket_by_size
    xs, ys, infos = [None] * len(problems), [None] * len(problems), np.zeros((len(problems), len(INFO_FIELDS)))
    uniform = len({(p.n, p.m) for p in problems}) <= 1
    buckets = [list(range(len(problems)))] if uniform else bucket_by_size(problems)
    for bucket in buckets:
        if not bucket:
            continue
        bt = make_batch([problems[k] for k in bucket])
        bt.solve()
        info = pack_info(bt)
        X, Y = bt.solution()   # ONE device-to-host copy per bucket (solution_of() copies the whole batch on every call)
        dims = getattr(bt, "dims", None)
        for pos, k in enumerate(bucket):
            nk, mk = dims[pos] if dims is not None else (int(problems[k].n), int(problems[k].m))
            xs[k], ys[k], infos[k] = X[pos, :nk].copy(), Y[pos, :mk].copy(), info[pos]
        if hasattr(bt, "close"):
            bt.close()
    return xs, ys, infos


def solve_sharded(problems, make_batch, dist=None, device=None):
    """Solve `problems` (the full list, same on every rank) sharded over the process group.
    make_batch(list_of_problems) -> QpalmBatch.  Returns (x, y, info) on rank 0, None elsewhere; x is [nqp][n_max] (rows of
    smaller members are zero-padded), y likewise."""
    import torch
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    nqp = len(problems)
    sizes = [(int(p.n), int(p.m)) for p in problems]
    mixed = len(set(sizes)) > 1
    n, m = max(s[0] for s in sizes), max(s[1] for s in sizes)
    shard = lambda r: shard_indices(nqp, world, r, sizes if mixed else None)
    mine = shard(rank)
    per = (nqp + world - 1) // world
    # equal padded shards so that one gather moves everything (24 KB per QP at n=1000, m=2000)
    payload = np.zeros((per, n + m + len(INFO_FIELDS)))
    if len(mine):
        xs, ys, info = _solve_local([problems[i] for i in mine], make_batch)
        for pos in range(len(mine)):
            payload[pos, :len(xs[pos])] = xs[pos]
            payload[pos, n:n + len(ys[pos])] = ys[pos]
        payload[:len(mine), n + m:] = info
    if world == 1:
        blk = payload[:len(mine)]
        X, Y, I = np.zeros((nqp, n)), np.zeros((nqp, m)), np.zeros((nqp, len(INFO_FIELDS)))
        X[mine], Y[mine], I[mine] = blk[:, :n], blk[:, n:n + m], blk[:, n + m:]
        return X, Y, I
    t = torch.from_numpy(payload)
    if device is not None:
        t = t.to(device)
    bufs = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
    dist.gather(t, bufs, dst=0)
    if rank != 0:
        return None
    X, Y, I = np.zeros((nqp, n)), np.zeros((nqp, m)), np.zeros((nqp, len(INFO_FIELDS)))
    for r in range(world):
        idx = shard(r)
        blk = bufs[r].cpu().numpy()[:len(idx)]
        X[idx], Y[idx], I[idx] = blk[:, :n], blk[:, n:n + m], blk[:, n + m:]
    return X, Y, I
