"""The row_progress leg of bench.py on its own (DAV1D_HIP_TRACE_FRAME=1 adds the section times of frame_run and frame_lr_banded on
stderr): python tools/row_progress_probe.py [threads] [frames]"""
import sys, os, json
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
import lister_util as lu
from dav1d_amd import api
thr = int(sys.argv[1]) if len(sys.argv) > 1 else 64
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 7
ctx = api.Context(0)
print(json.dumps(lu.row_progress_cost(ctx, 7680, 4320, 10, 16, 8, threads=thr, frames=frames)))
