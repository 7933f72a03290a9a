"""nova_amd -- MI355X (gfx950) commitment / MSM provider for microsoft/Nova.

The product is the C-ABI shared library `libnova_mi355x.so` (include/nova_mi355x.h).  This package is the
Python host-side mirror of the reference's provider interface for that path (DlogGroupExt /
CommitmentEngineTrait, /root/reference/src/provider/traits.rs:77-117, src/traits/commitment.rs:52-195) used by
tests and bench.py; it never computes a group operation itself.
"""
from . import _lib  # noqa: F401
from .provider import (  # noqa: F401
    BN254_G1, GRUMPKIN, PALLAS, VESTA, CURVE_NAMES, Commitment, CommitmentEngine, CommitmentKey, DlogGroup,
    NmxError, ShardedVector, init_devices, ipa_prove, shard_plan, svec_map,
)
