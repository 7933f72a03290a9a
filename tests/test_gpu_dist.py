"""The N > 1 code path with the real HIP MSM inside a real RCCL process group (-m gpu): backend "nccl" (= RCCL on ROCm),
world_size 1 -- the GPU boxes of the test tier have one GPU; the driver's SCALE run exercises 2/4/8.  sharded_msm /
round_robin_batch_msm are the functions bench.py --gpus N runs (the reference's par_chunks + reduce,
/root/reference/src/provider/msm.rs:564-574, and batch_vartime_multiscalar_mul, traits.rs:82-90)."""
import os
import socket

import numpy as np
import pytest

from oracle import cref
from oracle import pyref as R
from tests import util

pytestmark = pytest.mark.gpu


def test_sharded_and_round_robin_over_rccl_world1(nmx):
    import torch
    import torch.distributed as dist
    from nova_amd.dist import round_robin_batch_msm, shard_range, sharded_msm
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        c = R.BN254_G1
        g = nmx.DlogGroup(c.cid)
        n = 1 << 16
        lo, hi = shard_range(n, 0, 1)
        ck = nmx.CommitmentKey.generate(c.cid, n, k0=1 + lo)
        bases = ck.read(0, n)
        sc = util.random_scalars(c.cid, n, seed=77)
        dsc = torch.from_numpy(sc).cuda()
        timing = [0.0]
        got = sharded_msm(g, ck, dsc, timing=timing)
        assert (got.xy, int(got.is_inf)) == cref.msm(c.cid, sc, bases, n)
        assert 0 < timing[0] < 5.0
        lens = [n, 3, n // 2, 0]
        vecs = [util.random_scalars(c.cid, m, seed=5 + j) for j, m in enumerate(lens)]
        res = round_robin_batch_msm(g, ck, vecs)
        assert [(r.xy, int(r.is_inf)) for r in res] == cref.msm_batch(c.cid, [v.tobytes() for v in vecs], bases, n)
        ck.close()
    finally:
        dist.destroy_process_group()
