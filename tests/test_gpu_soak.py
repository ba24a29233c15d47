"""A 60 s x 4-engine parity soak inside the driver's suite (tools/soak_parity.py is the long form: 120 s x 6 engines x 8 seeds,
profiles/r06_soak.md): random dense play -- notes, releases, pedal, depth changes -- every block of every engine against its CPU oracle
engine under the dense-play bar (1e-5 relative, 1e-3 of the block's peak, ABS_FLOOR_DENSE), voice counts equal after every block."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_sixty_second_soak(hiplib, oracle):
    import soak_parity
    r = soak_parity.soak(seconds=60.0, n=4, seed=2026, verbose=True)
    assert r["blocks"] == 5625
    assert r["worst"] <= 1.0, r            # (soak() raises on the first block outside the bar; this is the margin on record)
    print("soak margin:", r)
