"""What a row-progress listener costs at 8K (bench.py's row_progress leg on its own): python tools/progress_probe.py"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from dav1d_amd import api
import lister_util as lu

ctx = api.Context(0)
print(json.dumps(lu.row_progress_cost(ctx, 7680, 4320, 10, 16, 8, threads=24, frames=9)))
