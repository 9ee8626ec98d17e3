"""Bounded, seeded run of the randomised GPU-vs-oracle differential sweep (tests/parity_sweep.py): random rows / k (2 .. 16384,
incl. the folded sizes) / batch, field-corner values, both commit entry points, forced pipeline chunking, openings and
the quadratic sub-proof polynomial -- every case asserted bit-exact against the oracle."""
import pytest

pytestmark = pytest.mark.gpu


def test_seeded_parity_sweep():
    from parity_sweep import sweep
    cases = sweep(45.0, 20261003)
    print(f"parity sweep: {cases} random cases bit-exact")
    assert cases >= 15
