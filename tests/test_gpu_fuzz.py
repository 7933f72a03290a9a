"""-m gpu: a fixed-seed slice of scripts/gpu_fuzz.py -- random curves, key sizes (with identity and duplicate bases),
prefixes / interior slices, scalar distributions, host and HBM-resident scalars, registered and one-shot keys -- every
result bit-exact against the oracle."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12])
def test_randomised_differential(nmx, seed):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gpu_fuzz.py"), "120", str(seed)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "fuzz ok" in r.stdout, (r.stdout[-500:], r.stderr[-500:])


def test_randomised_differential_ipa(nmx):
    """a fixed-seed slice of scripts/gpu_fuzz_ipa.py: nmx_ipa_prove over random curves, sizes, placements, layouts, key forms, structured
    witnesses and forced challenges against the oracle's key-folding restatement (and the reference's verifier every few cases)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gpu_fuzz_ipa.py"), "60", "21"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ipa fuzz ok" in r.stdout, (r.stdout[-500:], r.stderr[-500:])

