"""Multi-GPU sharding of one MSM: contiguous shards, one process per GPU, no data-path collective
except the final exchange of one 128-byte partial sum per rank -- the rayon `par_chunks` + `reduce(identity, +)` of
/root/reference/src/provider/msm.rs:566-571,667-673 turned into an RCCL all_gather over xGMI.  RCCL has no
elliptic-curve reduce op, so the "reduce of partial bucket sums" is all_gather(raw bytes) + nmx_point_sum.
The message is 128 B per rank (latency-bound); never exchange raw bucket arrays.
"""
import numpy as np


def shard_range(n, rank, world):
    """Contiguous shard [lo, hi) of n pairs for `rank` (matches how bases are a prefix of one resident key)."""
    assert 0 <= rank < world
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


_BUFFERS = {}  # (device, world) -> (send, recv): the exchange happens once per MSM, so its buffers are kept


def combine_partials(group, partial128, pg=None, device=None):
    """all_gather this rank's 128-byte partial (an NMX_OUT_PARTIAL result) and sum all of them.
    Works on any torch.distributed backend: 'nccl' (= RCCL, CUDA tensors) or 'gloo' (CPU tensors, tests)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(pg)
    nccl = dist.get_backend(pg) == "nccl"
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if nccl else torch.device("cpu")
    assert len(partial128) == 128
    key = (str(device), world)
    if key not in _BUFFERS:
        _BUFFERS[key] = (torch.zeros(128, dtype=torch.uint8, device=device),
                         torch.zeros(128 * world, dtype=torch.uint8, device=device))
    send, recv = _BUFFERS[key]
    send.copy_(torch.frombuffer(bytearray(partial128), dtype=torch.uint8))
    if nccl:
        dist.all_gather_into_tensor(recv, send, group=pg)  # one RCCL call on one flat buffer
    else:
        dist.all_gather(list(recv.view(world, 128).unbind(0)), send, group=pg)
    return group.point_sum(np.ascontiguousarray(recv.cpu().numpy().reshape(world, 128)))


def sharded_msm(group, ck_shard, scalars_shard, pg=None, mont=False, timing=None, msm_fn=None):
    """Each rank: full single-GPU MSM over its shard (key shard resident in HBM) -> 128-byte partial -> combine.
    timing: a one-element list that accumulates the seconds spent in the exchange + point sum alone (bench.py's
    `combine_ms`).  msm_fn(scalars_shard, ck_shard) -> 128-byte partial replaces the per-shard MSM (the CPU tests
    inject the oracle there: there is no GPU in the build container)."""
    import time
    if msm_fn is None:
        part = group.vartime_multiscalar_mul(scalars_shard, ck_shard, mont=mont, partial=True).xy
    else:
        part = msm_fn(scalars_shard, ck_shard)
    t0 = time.perf_counter()
    out = combine_partials(group, part, pg)
    if timing is not None:
        timing[0] += time.perf_counter() - t0
    return out


def round_robin_batch_msm(group, ck, scalar_vecs, pg=None, mont=False, batch_fn=None):
    """`batch_vartime_multiscalar_mul` (src/provider/traits.rs:82-90) across the GPUs of a node: WHOLE vectors are dealt
    round-robin to the ranks (short MSMs are not worth sharding), every rank holds the full key, runs
    its share as one batch and the k affine results (65 bytes each) are all-gathered.  Returns the k Commitments in
    input order on every rank.  batch_fn(vecs, ck) -> [(xy64, is_inf)] replaces the per-rank batch in the CPU tests."""
    import torch
    import torch.distributed as dist
    from .provider import Commitment
    world, rank = dist.get_world_size(pg), dist.get_rank(pg)
    k = len(scalar_vecs)
    mine = list(range(rank, k, world))
    if batch_fn is None:
        res = [(c.xy, int(c.is_inf)) for c in group.batch_vartime_multiscalar_mul([scalar_vecs[j] for j in mine], ck, mont)]
    else:
        res = batch_fn([scalar_vecs[j] for j in mine], ck)
    per = (k + world - 1) // world
    nccl = dist.get_backend(pg) == "nccl"
    device = torch.device("cuda", torch.cuda.current_device()) if nccl else torch.device("cpu")
    send = torch.zeros(65 * per, dtype=torch.uint8)
    for slot, (xy, inf) in enumerate(res):
        send[65 * slot: 65 * slot + 64] = torch.frombuffer(bytearray(xy), dtype=torch.uint8)
        send[65 * slot + 64] = inf
    send = send.to(device)
    recv = torch.zeros(65 * per * world, dtype=torch.uint8, device=device)
    if nccl:
        dist.all_gather_into_tensor(recv, send, group=pg)
    else:
        dist.all_gather(list(recv.view(world, 65 * per).unbind(0)), send, group=pg)
    flat = recv.cpu().numpy().reshape(world, per, 65)
    return [Commitment(flat[j % world, j // world, :64].tobytes(), bool(flat[j % world, j // world, 64])) for j in range(k)]
