#!/usr/bin/env python3
"""Runs only the intra wavefront pass of the bench frame (enqueued launches), a few times; for rocprofv3 --kernel-trace.
usage: python tools/intra_probe.py [--width W --height H --reps N]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from dav1d_amd import api
import synth_frames as synth  # noqa: E402
import test_postchain  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=7680)
ap.add_argument("--height", type=int, default=4320)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--graph", type=int, default=0)
ap.add_argument("--flow", type=int, default=0)
ap.add_argument("--mode", type=int, default=-1, help="force this prediction mode on every block (timing experiments)")
ap.add_argument("--no-tx", type=int, default=0, help="drop the residuals")
a = ap.parse_args()
ctx = api.Context(0)
frame = synth.make_frame(a.width, a.height, 10, seed=0xD1D)
intra = synth.make_intra_pass(frame, seed=0x1A7)
if a.mode >= 0:
    for p_, t_ in intra.batches:
        p_["mode"] = np.where((p_["tw"] > 8) & (a.mode == 13), 12, a.mode)
        p_["angle"] = 0
if a.no_tx:
    intra.batches = [(p_, t_[:0]) for p_, t_ in intra.batches]
pic = ctx.picture(a.width, a.height, api.LAYOUT_I420, 10)
import time
ms = []
for _ in range(a.reps):
    t0 = time.perf_counter()
    try:
        test_postchain.hip_intra(ctx, intra, pic, timed=False, graph=bool(a.graph), flow=bool(a.flow))
    except AssertionError:
        pass
    ms.append(round((time.perf_counter() - t0) * 1e3, 2))
print("intra pass ms:", ms, "steps", len(intra.batches), "blocks", intra.n_blocks)
