"""The end-to-end route of bench.py (dav1d_amd/e2e.py): pass-1 hand-off arrays -> pass-2 lister on host threads -> chunks ->
device, for an inter frame and for a key frame (every block through the intra wavefront), checked against the reference's
own pass 2 on a real Dav1dFrameContext."""
import pytest

import lister_util as lu
from dav1d_amd import e2e


@pytest.mark.parametrize("key_frame", [False, True], ids=["inter", "key"])
def test_end_to_end_route_matches_reference_pass2(ctx, key_frame, twin_refs):
    if lu.ref_lib() is None:
        pytest.skip("no reference build (oracle/_ref)")
    w, h = (384, 256) if ctx.backend == "emu" else (1920, 1080)
    out = e2e.run(ctx, w, h, 10, frames=2, threads=3, tile_cols=2, tile_rows=2, seed=77, key_frame=key_frame,
                  check=lambda ho, planes, refs: lu.check_handoff_against_reference(ho, planes, refs, is_inter=not key_frame))
    assert out["parity"].startswith("bit-exact"), out["parity"]
    assert out["wavefront_steps"] >= (2 if key_frame else 0)


def test_full_route_with_filters_matches_the_reference(ctx):
    """lister_util.full_route_rate (bench.py's end_to_end_full_table leg): hand-off arrays + pass 1's filter inputs -> library threads
    (dav1d_hip_lister_run, dav1d_hip_lister_filter_run) -> one dav1d_hip_frame_end with every stage; the helper raises when the
    filtered picture differs from dav1d_decode_tile_sbrow + dav1d_filter_sbrow."""
    import lister_util as lu
    if lu.ref_lib() is None:
        pytest.skip("no reference build (oracle/_ref)")
    w, h = (384, 264) if ctx.backend == "emu" else (1920, 1080)
    out = lu.full_route_rate(ctx, w, h, 10, tile_cols=2, tile_rows=2, threads=4, frames=3)
    assert out and out["parity"].startswith("bit-exact") and out["total_ms"] > 0
