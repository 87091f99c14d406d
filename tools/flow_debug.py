#!/usr/bin/env python3
"""Bounded experiments on the intra dataflow launch: python tools/flow_debug.py <case>"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from dav1d_amd import api
import synth_frames as synth  # noqa: E402

case = sys.argv[1]
ctx = api.Context(0)
w, h, bpc = 1024, 576, 10
frame = synth.make_frame(w, h, bpc, seed=101)
ip = synth.make_intra_pass(frame, seed=27)
b = ip.batches
if case == "one":
    b = [(b[0][0][:1], b[0][1][:0])]            # one prediction, no residual
elif case == "onepair":
    p = b[-1][0][:1]
    t = b[-1][1][:1]
    b = [(p, t)]
elif case == "batch0":
    b = b[:1]
elif case == "two":
    b = b[:2]
elif case == "predonly":
    b = [(x, y[:0]) for x, y in b]
print(case, "batches", len(b), "units", sum(len(x) for x, _ in b), flush=True)
pic = ctx.picture(w, h, api.LAYOUT_I420, bpc)
coef = ctx.buffer_from(ip.coef)
fl = ctx.intra_flow(b)
print("created", fl.n_units, flush=True)
t0 = time.perf_counter()
fl.run(pic, coef)
print("enqueued", flush=True)
st = fl.status()
print("status", st, "ms %.2f" % ((time.perf_counter() - t0) * 1e3), flush=True)
