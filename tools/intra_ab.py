"""A/B of library builds on the intra wavefront (one process per library, one box): 8K key frame, inter frame with 10 % intra blocks, key frame
with 40 % intra block copies, end to end (frame_end_ms = gather + launches + sync).  python tools/intra_ab.py [lib.so] [--check]"""
import json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from dav1d_amd import api
import e2e
import lister_util as lu
lib = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else None
ctx = api.Context(0, lib_path=lib) if lib else api.Context(0); ctx.backend = "hip"
check = "--check" in sys.argv
out = {"lib": os.path.basename(lib) if lib else "tree"}
for name, kw, inter in (("key_frame", dict(key_frame=True, seed=0xE2F), False), ("inter_10pct_intra", dict(intra_pct=10, seed=0xE30), True),
                        ("key_frame_40pct_intrabc", dict(key_frame=True, intrabc_pct=40, seed=0xE31), False)):
    chk = (lambda ho, planes, refs, inter=inter: lu.check_handoff_against_reference(ho, planes, refs, is_inter=inter)) if check else None
    r = e2e.run(ctx, 7680, 4320, 10, frames=5, threads=64, tile_cols=16, tile_rows=8, check=chk, **kw)
    out[name] = {k: r.get(k) for k in ("frame_end_ms", "parity")}
print(json.dumps(out))
