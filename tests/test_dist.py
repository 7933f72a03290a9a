"""N > 1 path on CPU: world_size-2 (and 3) gloo processes run shard_range + combine_partials (all_gather of
128-byte partials + nmx_point_sum, which is host code and needs no GPU).  The per-shard MSM itself is stood in by
the oracle here (no GPU in this container); on the GPU box the same exchange runs over RCCL in bench.py."""
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
    from nova_amd.dist import combine_partials, shard_range
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
    xy, inf = cref.msm(cid, sc[lo:hi], bases[lo:hi], hi - lo)
    part = util.affine_to_partial(c.p, xy, inf)
    got = combine_partials(DlogGroup(cid), part)
    exp = cref.msm(cid, sc, bases, n)
    q.put((rank, (got.xy, int(got.is_inf)) == exp))
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
