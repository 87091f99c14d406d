import sys, time, json
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
import stream_util as su
from dav1d_amd import _lib
t=time.time()
r = su.task_loop_rate(_lib.DEFAULT_PATH, 7680, 4320, 10, tiles_log2=(2,0), threads=64, frame_delay=8, frames=int(sys.argv[1]) if len(sys.argv)>1 else 16)
print(json.dumps(r)); print("total %.1fs" % (time.time()-t))
