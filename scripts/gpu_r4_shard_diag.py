"""Round 4 diagnosis: bench.py --gpus 2 (oversubscribed, 2^22) reported gpu_matches_cpu = false.  Which form is wrong, at
which size, and does it depend on the two shards running at the same time on one physical GPU?"""
import os
import sys
import threading

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import nova_amd  # noqa: E402
from nova_amd import _lib  # noqa: E402
from oracle import cref  # noqa: E402

L = _lib.lib()
assert L.nmx_init(0) == 0
cid = 0
cref.set_threads(os.cpu_count())


def draw(cnt, seed):
    g = torch.Generator(device="cuda:0")
    g.manual_seed(seed)
    w = torch.randint(0, 1 << 31, (cnt, 8), dtype=torch.int64, device="cuda:0", generator=g)
    w = (w * 2 + torch.randint(0, 2, (cnt, 8), dtype=torch.int64, device="cuda:0", generator=g)).to(torch.int32)
    w[:, 7] &= 0x1FFFFFFF
    return w.view(torch.uint8).reshape(cnt, 32).contiguous()


def pt(c):
    return (c.xy, int(c.is_inf))


g = nova_amd.DlogGroup(cid)
for lg in (14, 21, 22, 23):
    total = 1 << lg
    for k in (1, 2, 3):
        assert nova_amd.init_devices(k, oversubscribe=True) == k
        assert L.nmx_set_option(b"shard_min_n", 1024) == 0
        ck = nova_amd.CommitmentKey.generate(cid, total, k0=1)
        plan = ck.shard_plan(0, total)
        pieces = [draw(cnt, 77 + dev) for dev, _o, cnt in plan]
        torch.cuda.synchronize()
        one = torch.cat(pieces).contiguous()
        torch.cuda.synchronize()
        host = one.cpu().numpy()
        print(f"  [2^{lg} k={k}] key + scalars ready", flush=True)
        hb2 = ck.read(0, total)
        if k == 1:
            hb = hb2
        exp = cref.msm(cid, host, hb, total)
        print(f"  [2^{lg} k={k}] oracle done", flush=True)
        row = {}
        for mode in (1, 2):
            assert L.nmx_set_option(b"combine", mode) == 0
            row[f"sharded/c{mode}"] = [pt(g.vartime_multiscalar_mul(pieces, ck)) == exp for _ in range(3)]
            print(f"  [2^{lg} k={k}] sharded combine {mode} done", flush=True)
        assert L.nmx_set_option(b"combine", 0) == 0
        row["device"] = [pt(g.vartime_multiscalar_mul(one, ck)) == exp for _ in range(3)]
        print(f"  [2^{lg} k={k}] device done", flush=True)
        row["host"] = [pt(g.vartime_multiscalar_mul(host, ck)) == exp for _ in range(2)]
        assert L.nmx_set_option(b"force_peer_copy", 1) == 0
        row["device/peer"] = [pt(g.vartime_multiscalar_mul(one, ck)) == exp for _ in range(2)]
        assert L.nmx_set_option(b"force_peer_copy", 0) == 0
        # the key itself
        row["key_ok"] = bool(np.array_equal(hb2, hb))
        print(f"2^{lg} k={k}", row, flush=True)
        ck.close()

# two unsharded keys, two host threads at once on one GPU
assert nova_amd.init_devices(1) == 1
for lg in (18, 20, 21):
    n = 1 << lg
    cks = [nova_amd.CommitmentKey.generate(cid, n, k0=1 + 1000 * i) for i in range(2)]
    scs = [draw(n, 5 + i) for i in range(2)]
    torch.cuda.synchronize()
    exps = [cref.msm(cid, scs[i].cpu().numpy(), cks[i].read(0, n), n) for i in range(2)]
    oks = [[], []]

    def work(i):
        for _ in range(6):
            oks[i].append(pt(g.vartime_multiscalar_mul(scs[i], cks[i])) == exps[i])
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    print(f"concurrent unsharded 2^{lg}", oks, flush=True)
    [c.close() for c in cks]
print("done")
