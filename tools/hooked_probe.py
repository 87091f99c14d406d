"""The dav1d task loop leg on its own: python tools/hooked_probe.py [--width W --height H --bpc B --threads T --frames N]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=7680); ap.add_argument("--height", type=int, default=4320); ap.add_argument("--bpc", type=int, default=10)
ap.add_argument("--threads", type=int, default=64); ap.add_argument("--frames", type=int, default=24); ap.add_argument("--delay", type=int, default=8)
ap.add_argument("--tile-cols", type=int, default=4); ap.add_argument("--tile-rows", type=int, default=1); ap.add_argument("--check-frames", type=int, default=3); ap.add_argument("--intra-pct", type=int, default=10)
a = ap.parse_args()
import hooked_util as hk
from dav1d_amd import _lib
print(json.dumps(hk.task_loop_rate(_lib.DEFAULT_PATH, a.width, a.height, a.bpc, (a.tile_cols, a.tile_rows), a.threads, a.delay, a.frames, a.check_frames, intra_pct=a.intra_pct)))
