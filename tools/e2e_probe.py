"""Frames-in-flight legs of bench.py on their own (no headline step): python tools/e2e_probe.py [--frames N] [--threads T] [--depth D]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--depth", type=int, default=2)
    ap.add_argument("--tiles", default="16x8")
    ap.add_argument("--no-full", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    a = ap.parse_args()
    from dav1d_amd import api
    import e2e
    import lister_util as lu
    ctx = api.Context(0)
    tc, tr = (int(v) for v in a.tiles.split("x"))
    chk = None if a.no_check else (lambda ho, planes, refs: lu.check_handoff_against_reference(ho, planes, refs))
    out = {"recon": e2e.run_sustained(ctx, 7680, 4320, 10, frames=a.frames, threads=a.threads or None, tile_cols=tc, tile_rows=tr, depth=a.depth, check=chk)}
    if not a.no_full:
        out["full_table"] = lu.full_route_sustained(ctx, 7680, 4320, 10, tc, tr, threads=a.threads or None, frames=a.frames, depth=a.depth)
    for v in out.values():
        v.pop("workload", None)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
