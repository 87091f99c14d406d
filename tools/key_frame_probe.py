"""8K key frame (and an inter frame with 10 % intra blocks) end to end, with the superblocks waiting per block (intra_sb_fine 1) and per
superblock (0); each checked against the reference's pass 2.  python tools/key_frame_probe.py [--no-check]"""
import json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from dav1d_amd import api
import e2e
import lister_util as lu
ctx = api.Context(0, lib_path=os.environ["DAV1D_HIP_LIB"]) if os.environ.get("DAV1D_HIP_LIB") else api.Context(0); ctx.backend = "hip"
out = {}
check = "--no-check" not in sys.argv
for fine in (1, 0):
    assert ctx.lib.dav1d_hip_set_option(ctx.h, b"intra_sb_fine", fine) == 0
    for name, kw, inter in (("key_frame", dict(key_frame=True, seed=0xE2F), False), ("inter_10pct_intra", dict(intra_pct=10, seed=0xE30), True),
                            ("key_frame_40pct_intrabc", dict(key_frame=True, intrabc_pct=40, seed=0xE31), False)):
        chk = (lambda ho, planes, refs, inter=inter: lu.check_handoff_against_reference(ho, planes, refs, is_inter=inter)) if check else None
        r = e2e.run(ctx, 7680, 4320, 10, frames=4, threads=64, tile_cols=16, tile_rows=8, check=chk, **kw)
        out["%s_fine%d" % (name, fine)] = {k: r.get(k) for k in ("ms_per_frame", "frame_end_ms", "list_ms", "parity", "value")}
print(json.dumps(out))
