"""Regenerates tests/golden/keys/*: a tiny .ptau (BN254, power 2, G1 = (7+i)*G for i < 4, 128 opaque G2 bytes) and a tiny
PEDERSEN_KEY (Pallas, h = 5*G, ck = (6+i)*G for i < 4), written by the oracle's restatement of the reference writers
(oracle/keyfiles.py: write_ptau = src/provider/ptau.rs:205-268, save_setup = src/provider/pedersen.rs:383-393).  The
reference stores no key file; these pin OUR reader/writer pair against regressions, not the upstream byte layout."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import keyfiles as K
from oracle import pyref as R

here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "keys")
os.makedirs(here, exist_ok=True)
open(os.path.join(here, "tiny_bn254.ptau"), "wb").write(K.write_ptau(R.BN254_G1, R.sequential_bases(R.BN254_G1, 7, 4), bytes(range(128)), power=2))
pts = R.sequential_bases(R.PALLAS, 5, 5)
open(os.path.join(here, "tiny_pallas.key"), "wb").write(K.write_pedersen_key(R.PALLAS, pts[0], pts[1:]))
