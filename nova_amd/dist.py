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


def sharded_msm(group, ck_shard, scalars_shard, pg=None, mont=False):
    """Each rank: full single-GPU MSM over its shard (key shard resident in HBM) -> partial -> combine."""
    part = group.vartime_multiscalar_mul(scalars_shard, ck_shard, mont=mont, partial=True)
    return combine_partials(group, part.xy, pg)
