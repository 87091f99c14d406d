"""The dav1d_task_loop_real_pass1 leg of bench.py on its own: python tools/stream_task_loop_probe.py [frames] [seg_pin] [threads]"""
import sys, time, json
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
import stream_util as su
from dav1d_amd import _lib
t = time.time()
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 16
pin = int(sys.argv[2]) if len(sys.argv) > 2 else 0
thr = int(sys.argv[3]) if len(sys.argv) > 3 else 64
r = su.task_loop_rate(os.environ.get("DAV1D_HIP_LIB") or _lib.DEFAULT_PATH, 7680, 4320, 10, tiles_log2=(2, 0), threads=thr, frame_delay=8, frames=frames, seg_pin=pin)
r["total_s"] = round(time.time() - t, 1)
print(json.dumps(r))
