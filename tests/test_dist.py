"""N > 1 path on CPU: world_size-2 (and 3) gloo processes drive nova_amd.dist.sharded_msm / round_robin_batch_msm
themselves (shard_range, the all_gather of 128-byte partials, nmx_point_sum -- host code, needs no GPU) with the
per-shard MSM injected (the oracle: there is no GPU in this container).  On the GPU box the same functions run with
the HIP MSM inside an `nccl` process group: tests/test_gpu_dist.py and bench.py --gpus N."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, cid, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from nova_amd import DlogGroup
    from nova_amd.dist import round_robin_batch_msm, shard_range, sharded_msm
    from oracle import cref
    from oracle import pyref as R
    from tests import util
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = R.CURVES_BY_ID[cid]
    bases = cref.sequential_bases(c, 17, n)
    sc = util.random_scalars(cid, n, seed=3)
    lo, hi = shard_range(n, rank, world)

    def shard_msm(s, b):          # stands in for the HIP MSM of this rank's shard: same contract, 128-byte partial
        xy, inf = cref.msm(cid, s, b, len(b))
        return util.affine_to_partial(c.p, xy, inf)

    timing = [0.0]
    got = sharded_msm(DlogGroup(cid), bases[lo:hi], sc[lo:hi], timing=timing, msm_fn=shard_msm)
    exp = cref.msm(cid, sc, bases, n)
    ok = (got.xy, int(got.is_inf)) == exp and timing[0] > 0
    # batch_msm across ranks: whole vectors round-robin, ragged lengths incl. an empty vector
    lens = [n, 0, n // 2, 1, n - 1][: max(2, world + 2)]
    vecs = [util.random_scalars(cid, m, seed=40 + j) for j, m in enumerate(lens)]
    res = round_robin_batch_msm(DlogGroup(cid), bases, vecs,
                                batch_fn=lambda vs, b: cref.msm_batch(cid, [v.tobytes() for v in vs], b, n))
    ok = ok and [(r.xy, int(r.is_inf)) for r in res] == cref.msm_batch(cid, [v.tobytes() for v in vecs], bases, n)
    q.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n,cid", [(2, 101, 0), (3, 2, 2)])
def test_sharded_combine_gloo(world, n, cid):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, cid, q)) for r in range(world)]
    [p.start() for p in procs]
    res = [q.get(timeout=180) for _ in range(world)]
    [p.join(timeout=60) for p in procs]
    assert sorted(r for r, _ in res) == list(range(world))
    assert all(ok for _, ok in res)


def test_shard_range_covers():
    from nova_amd.dist import shard_range
    for n in (0, 1, 7, 8, 1 << 20, (1 << 24) + 5):
        for w in (1, 2, 3, 8):
            rs = [shard_range(n, r, w) for r in range(w)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in rs) - min(h - l for l, h in rs) <= 1
