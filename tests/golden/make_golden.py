#!/usr/bin/env python3
"""Regenerates tests/golden/dsp_golden.json from the reference's own C path (oracle/_ref, built
from /root/reference by oracle/Makefile).  Run in the build container only:
    python tests/golden/make_golden.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import util            # noqa: E402
import golden_cases    # noqa: E402

oracle = util.Oracle("ref")
out = {name: fn(oracle) for name, fn in sorted(golden_cases.all_cases().items())}
with open(os.path.join(HERE, "dsp_golden.json"), "w") as f:
    json.dump({"generator": "oracle/_ref (reference C path, dav1d 1.5.4)", "sha256": out}, f, indent=1, sort_keys=True)
print("wrote %d digests" % len(out))
