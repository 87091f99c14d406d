"""Where the cycles of the intra wavefront go ("nobody has yet measured what the 3 us consist of").  A -DDV_PHASES variant
of intra_sb.hip (tools/build_variant.py phases "-DDV_PHASES" intra_sb.hip) marks every wave's phases with the shader clock; this runs an
8K key frame and an inter frame with 10 % intra blocks end to end through it and prints cycles per wave / per unit / per group.
    python tools/intra_phase_probe.py dav1d_amd/build/variants/phases.so"""
import ctypes as C
import json
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
from dav1d_amd import api   # noqa: E402
import e2e                  # noqa: E402

lib = sys.argv[1]
ctx = api.Context(0, lib_path=lib)
ctx.backend = "hip"
raw = C.CDLL(lib)
NAMES = ["start_up", "wait_for_neighbour_superblocks", "prediction", "transform", "store", "group_end_stores_acked_barrier_publish", "wave_total"]
out = {}
for name, kw in (("key_frame", dict(key_frame=True, seed=0xE2F)), ("inter_10pct_intra", dict(intra_pct=10, seed=0xE30))):
    e2e.run(ctx, 7680, 4320, 10, frames=2, threads=64, tile_cols=16, tile_rows=8, **kw)          # warm-up: pools, first-use allocations
    assert raw.dav1d_hip_debug_phases_intra_sb(None, 1) == 0
    n_frames = 3
    r = e2e.run(ctx, 7680, 4320, 10, frames=n_frames + 1, threads=64, tile_cols=16, tile_rows=8, **kw)
    buf = (C.c_ulonglong * 1024)()
    assert raw.dav1d_hip_debug_phases_intra_sb(buf, 0) == 0
    v = list(buf)
    waves, units, groups, idle = v[907], v[908], v[909], v[910]
    d = {"frame_end_ms": r.get("frame_end_ms"), "frames_measured": n_frames + 1, "waves": waves, "units": units, "wave_turns_through_a_group": groups,
         "turns_without_a_unit": idle}
    d["cycles_per_wave"] = {NAMES[k]: round(v[900 + k] / max(1, waves)) for k in range(7)}
    d["share_of_wave"] = {NAMES[k]: round(v[900 + k] / max(1, v[906]), 3) for k in range(6)}
    d["cycles_per_unit"] = {NAMES[k]: round(v[900 + k] / max(1, units)) for k in (1, 2, 3, 4)}
    d["cycles_per_group_turn"] = {NAMES[5]: round(v[905] / max(1, groups))}
    # the transform body's own marks (itx_body.h, slots 512 + tx * 16 + 8: the PRED_LDS form the units use)
    tx = {}
    for t in range(19):
        b = 512 + t * 16 + 8
        if v[b + 5]:
            tx["tx%d" % t] = dict(zip(("loads_landed", "rows_in_regs", "row_pass", "column_pass_store", "body"), [round(v[b + k] / v[b + 5]) for k in range(5)]), n=v[b + 5])
    d["transform_bodies"] = tx
    out[name] = d
print(json.dumps(out, indent=1))
