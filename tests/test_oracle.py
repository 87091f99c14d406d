"""Pins the oracles: oracle/port (the C restatement) and, when present, oracle/_ref (the reference's
own C path) must reproduce the committed digests of tests/golden/dsp_golden.json, which were
generated from the reference C functions (tests/golden/make_golden.py)."""
import json
import os

import pytest

import util
import golden_cases

GOLD = json.load(open(os.path.join(util.GOLDEN, "dsp_golden.json")))["sha256"]
CASES = golden_cases.all_cases()


def test_golden_file_covers_every_case():
    assert set(GOLD) == set(CASES)


@pytest.mark.parametrize("name", sorted(CASES))
def test_port_matches_golden(name):
    assert CASES[name](util.Oracle("port")) == GOLD[name]


@pytest.mark.parametrize("name", sorted(CASES))
def test_reference_build_matches_golden(name):
    if util.ref_lib() is None:
        pytest.skip("oracle/_ref not available")
    assert CASES[name](util.Oracle("ref")) == GOLD[name]


@pytest.mark.parametrize("name", sorted(CASES))
def test_release_build_of_the_reference_matches_golden(name):
    """oracle/_ref_release (dav1d's own release flags: -O3 -DNDEBUG -fomit-frame-pointer -ffast-math, the CPU peer bench.py times) gives
    the digests of the asserts-on build that checks parity: the peer that is timed computes what the checker computes."""
    if util.ref_release_lib() is None:
        pytest.skip("oracle/_ref_release not available")
    assert CASES[name](util.Oracle("ref_release")) == GOLD[name]
