"""CPU model of the single-pass suffix-Horner scan (nova_amd/csrc/fieldvec.hip: k_horner_scan) -- the same steps in big-int
arithmetic, lane by lane and tile by tile, against the oracle's suffix Horner.  It pins the ALGEBRA the kernel relies on
(per-lane constants U, V, the tile / group weights W, X, the sub-tile powers, the aggregate -> inclusive hand-over across
groups, look-back rounds of fewer than 64 groups) independently of the GPU; the kernel itself is compared with the oracle
bit for bit in tests/test_gpu_fieldvec.py.  Geometry is scaled down (K coefficients per lane, G tiles per group) so that many
groups and rounds fit in a few thousand coefficients; K = 8, G = 64 is the kernel's."""
import random

import numpy as np
import pytest

from oracle import cref
from tests import fv_common as C

LANES = 64


def scan_model(f, u, p, K, J, G, window, rng):
    """out[i] = sum_{k >= i} f[k] u^(k - i) the way k_horner_scan computes it.  Tiles are visited in dispatch order (last
    tile first); which of the group values behind a tile are already inclusive is drawn at random among the legal states."""
    n = len(f)
    sub = LANES * K                      # coefficients per sub-tile
    T = J * sub                          # per tile
    nt = (n + T - 1) // T
    ng = (nt + G - 1) // G
    uK = pow(u, K, p)
    U = [pow(uK, l, p) for l in range(LANES)]                      # u^(K l)
    V = [pow(pow(uK, -1, p), l + 1, p) for l in range(LANES)]      # u^(-K (l + 1))
    uS = pow(u, sub, p)
    uT = pow(u, T, p)
    W = [pow(uT, m, p) for m in range(G + 1)]                      # tiles
    X = [pow(pow(uT, G, p), m, p) for m in range(window + 1)]      # groups
    fz = list(f) + [0] * (nt * T - n)
    S = {}                                                         # (tile, sub-tile) -> per-lane suffix sums
    agg = [None] * nt
    for tile in range(nt - 1, -1, -1):                             # phase 1 of every tile
        A = 0
        for j in range(J):
            e = tile * T + j * sub
            head = [sum(fz[e + K * l + k] * pow(u, k, p) for k in range(K)) % p for l in range(LANES)]
            H = [head[l] * U[l] % p for l in range(LANES)]
            s = [sum(H[l:]) % p for l in range(LANES)] + [0]
            S[tile, j] = s
            A = (A + pow(uS, j, p) * s[0]) % p
        agg[tile] = A
    gagg = [sum(W[m] * agg[g * G + m] for m in range(min(G, nt - g * G))) % p for g in range(ng)]
    ginc = [None] * ng
    ginc[ng - 1] = gagg[ng - 1]
    out = [0] * (nt * T)
    for tile in range(nt - 1, -1, -1):
        g, pos = divmod(tile, G)
        gt = min(G, nt - g * G)
        C_ = sum(W[m] * agg[tile + 1 + m] for m in range(gt - 1 - pos)) % p     # inside the group
        if g != ng - 1:
            GC, scale, base = 0, 1, g + 1
            while True:                                                           # rounds of `window` groups
                states = []
                for m in range(window):
                    gg = base + m
                    if gg >= ng:
                        states.append((2, 0))
                    elif ginc[gg] is not None and (gg == ng - 1 or rng.random() < 0.5):
                        states.append((2, ginc[gg]))
                    else:
                        states.append((1, gagg[gg]))
                fi = next((m for m, (st, _) in enumerate(states) if st == 2), 64)
                GC = (GC + scale * sum(X[m] * v for m, (_, v) in enumerate(states) if m <= fi)) % p
                if fi < 64:
                    break
                scale = scale * X[window] % p
                base += window
            C_ = (C_ + W[G - 1 - pos] * GC) % p
            if ginc[g] is None:                                                   # (the closer's job; any tile of the group knows GC)
                ginc[g] = (gagg[g] + X[1] * GC) % p
        t = C_
        for j in range(J - 1, -1, -1):                                            # phase 3, from the top sub-tile down
            TC = uS * t % p
            e = tile * T + j * sub
            for l in range(LANES):
                c = (S[tile, j][l + 1] + TC) * V[l] % p
                for k in range(K - 1, -1, -1):
                    c = (fz[e + K * l + k] + u * c) % p
                    out[e + K * l + k] = c
                if l == 0:
                    t = c                                                         # lane 0 ends on the sub-tile's first coefficient
    return out[:n]


@pytest.mark.parametrize("fid", [1, 2])
@pytest.mark.parametrize("K,J,G,window,n", [(1, 1, 4, 64, 64 * 4 * 7 + 13), (1, 2, 4, 2, 64 * 2 * 4 * 9), (2, 4, 2, 1, 64 * 8 * 2 * 5 + 1),
                                            (1, 1, 4, 3, 64 * 4 * 30 + 1), (8, 1, 64, 64, 512 * 64 + 700)])
def test_scan_model_matches_the_oracle(fid, K, J, G, window, n):
    p = C.FIELDS[fid]
    rng = random.Random(1000 * fid + n)
    f = C.edge_vectors(fid, n, 3)
    u = C.rand_vec(fid, 1, 4)
    exp = np.frombuffer(cref.suffix_horner(fid, f, n, u), np.uint8).reshape(n, 32)
    got = scan_model(C.ints(f), C.ints(u)[0], p, K, J, G, window, rng)
    assert got == C.ints(exp)
