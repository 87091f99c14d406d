"""The N > 1 path of bench.py on CPU: world_size-2 gloo run of the frame sharding + timing reduce."""
import os
import subprocess
import sys
import textwrap

import util


def test_frame_parallel_world2_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent("""
        import sys, json
        sys.path.insert(0, %r)
        import torch.distributed as dist
        from dav1d_amd import dist as dd
        rank, local, world = dd.env()
        dist.init_process_group("gloo")
        mine = dd.frames_of_rank(9, rank, world)
        dd.barrier(world)
        t = dd.max_over_ranks(1.0 + rank, world)          # slowest rank wins
        v = dd.job_throughput(100.0, 4, t, world)
        out = [None] * world
        dist.all_gather_object(out, mine)
        if rank == 0:
            print(json.dumps({"t": t, "v": v, "frames": out}))
        dist.destroy_process_group()
    """ % util.ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29517", str(script)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["t"] == 2.0 and d["v"] == 2 * 100.0 * 4 / 2.0
    assert d["frames"] == [[0, 2, 4, 6, 8], [1, 3, 5, 7]]


TILE_COL_WORKER = """
import sys, json
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
import numpy as np
import torch, torch.distributed as dist
import util, test_frame
from dav1d_amd import api, dist as dd
import synth_frames as synth

rank, local, world = dd.env()
dist.init_process_group("gloo")
W, H, BPC = 512, 128, 10
ctx = util.make_context("emu")
cols = dd.tile_columns(W, world)
rng = np.random.default_rng(11)
ref_host = [synth.make_planes(rng, W, H, BPC) for _ in range(3)]
dst0 = synth.make_planes(rng, W, H, BPC, smooth=False)
frames = [synth.make_frame(W, H, BPC, seed=900 + n, mv_range_px=48) for n in range(2)]

# every rank holds all references; the picture being reconstructed becomes reference 0 of the next frame
refs = []
for r in ref_host:
    p = dd.SharedPicture(ctx, W, H, api.LAYOUT_I420, BPC, "cpu")
    for pl in range(3):
        p.upload(pl, r[pl])
    refs.append(p)
prev = None
for n, frame in enumerate(frames):
    cur = dd.SharedPicture(ctx, W, H, api.LAYOUT_I420, BPC, "cpu")
    for pl in range(3):
        cur.upload(pl, dst0[pl])
    stride_px = [cur.view.stride_px(pl) for pl in range(3)]
    mine = dd.tasks_by_column(frame.mc, frame.comp, frame.itx, stride_px, cols)[rank]
    rlist = [prev.view if (prev is not None and k == 0) else refs[k].view for k in range(3)]
    prep = ctx.buffer(frame.prep_elems * 2); prep.zero()
    coef = ctx.buffer_from(frame.coef)
    ctx.mc_batch(cur.view, rlist, frame.mc[mine[0]], prep)
    if len(mine[1]):
        ctx.comp_batch(cur.view, frame.comp[mine[1]], prep, None)
    ctx.itx_add_batch(cur.view, frame.itx[mine[2]], coef)
    ctx.sync()
    own = [cur.download(pl).copy() for pl in range(3)]
    dd.allgather_tile_columns(cur, cols, rank, world)
    prev = cur
    counts = [len(mine[0]), len(mine[1]), len(mine[2])]

# the oracle reconstructs both frames whole, on one rank
if rank == 0:
    oracle = util.default_oracle()
    want_prev = None
    for n, frame in enumerate(frames):
        rl = [want_prev if (want_prev is not None and k == 0) else ref_host[k] for k in range(3)]
        want, _, _ = test_frame.oracle_frame(oracle, frame, dst0, rl)
        want_prev = want
    got = [prev.download(pl) for pl in range(3)]
    ok = all(np.array_equal(got[pl], want[pl]) for pl in range(3))
    # what this rank alone produced must differ from the whole picture outside its column (the gather did the work)
    partial = any(not np.array_equal(own[pl], want[pl]) for pl in range(3))
    print(json.dumps({"ok": bool(ok), "partial": bool(partial), "cols": cols, "counts": counts}))
dist.barrier()
dist.destroy_process_group()
"""


def test_tile_columns_world2_gloo(tmp_path):
    """Tile-column sharding end to end on two CPU ranks: each reconstructs its column with the
    SIMT-emulated kernels, one all-gather per frame rebuilds the picture, the second frame predicts from the gathered
    picture; the result equals the oracle's whole-frame replay."""
    import json
    script = tmp_path / "tc.py"
    script.write_text(TILE_COL_WORKER % {"root": util.ROOT, "tests": os.path.join(util.ROOT, "tests")})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29519", str(script)],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["ok"] and d["partial"], d
    assert d["cols"] == [[0, 256], [256, 512]]
    assert all(c > 0 for c in d["counts"])


POST_WORKER = """
import sys, json
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
import numpy as np
import torch, torch.distributed as dist
import util, test_frame, test_postchain
from dav1d_amd import api, dist as dd
import synth_frames as synth

rank, local, world = dd.env()
dist.init_process_group("gloo")
W, H, BPC = 512, 192, 10
ctx = util.make_context("emu")
cols = dd.tile_columns(W, world)
rng = np.random.default_rng(21)
ref_host = [synth.make_planes(rng, W, H, BPC) for _ in range(3)]
dst0 = synth.make_planes(rng, W, H, BPC, smooth=False)
frame = synth.make_frame(W, H, BPC, seed=933, mv_range_px=48)
post = synth.make_post_filters(frame, seed=41)

refs = []
for r in ref_host:
    p = dd.SharedPicture(ctx, W, H, api.LAYOUT_I420, BPC, "cpu")
    for pl in range(3):
        p.upload(pl, r[pl])
    refs.append(p)
cur, cdf, res = (dd.SharedPicture(ctx, W, H, api.LAYOUT_I420, BPC, "cpu") for _ in range(3))
for pl in range(3):
    cur.upload(pl, dst0[pl])
stride_px = [cur.view.stride_px(pl) for pl in range(3)]
mine = dd.tasks_by_column(frame.mc, frame.comp, frame.itx, stride_px, cols)[rank]
prep = ctx.buffer(frame.prep_elems * 2); prep.zero()
coef = ctx.buffer_from(frame.coef)
ctx.mc_batch(cur.view, [r.view for r in refs], frame.mc[mine[0]], prep)
if len(mine[1]):
    ctx.comp_batch(cur.view, frame.comp[mine[1]], prep, None)
ctx.itx_add_batch(cur.view, frame.itx[mine[2]], coef)
ctx.sync()
# the in-loop filters of this rank's column: neighbours' edge columns first, then deblocking (in place), CDEF, restoration
dd.exchange_halo(cur, cols, rank, world)
lf_c, cdef_c, lr_c = dd.post_tasks_of_column(post.lf, post.cdef, post.lr, stride_px, cols[rank])
lvl = ctx.buffer_from(post.lvl)
ctx.lf_batch(cur.view, lf_c, lvl, post.b4_stride, post.lut_e, post.lut_i)
for pl in range(3):
    cdf.planes[pl].copy_(cur.planes[pl])
ctx.cdef_batch(cdf.view, cur.view, cdef_c, post.cdef_damping)
for pl in range(3):
    res.planes[pl].copy_(cdf.planes[pl])
ctx.lr_batch(res.view, cdf.view, cur.view, lr_c)
ctx.sync()
dd.allgather_tile_columns(res, cols, rank, world)
share = [len(lf_c) / len(post.lf), len(cdef_c) / len(post.cdef), len(lr_c) / max(1, len(post.lr))]
if rank == 0:
    oracle = util.default_oracle()
    want, _, _ = test_frame.oracle_frame(oracle, frame, dst0, ref_host)
    d, c, r, _ = test_postchain.oracle_post(oracle, post, want, W, H, BPC, with_grain=False)
    got = [res.download(pl) for pl in range(3)]
    bad = []
    for pl in range(3):
        vh, vw = (H, W) if pl == 0 else (H // 2, W // 2)
        if not np.array_equal(got[pl][:vh, :vw], r[pl][:vh, :vw]):
            yy, xx = np.nonzero(got[pl][:vh, :vw] != r[pl][:vh, :vw])
            bad.append((pl, int(yy[0]), int(xx[0]), int(len(yy)), int(xx.min()), int(xx.max())))
    changed = any(np.any(r[pl] != want[pl]) for pl in range(3))
    print(json.dumps({"ok": not bad, "bad": bad, "cols": cols, "share": share, "filters_changed_pixels": bool(changed)}))
dist.barrier()
dist.destroy_process_group()
"""


def test_tile_columns_with_in_loop_filters_world2_gloo(tmp_path):
    """Tile-column sharding with deblocking, CDEF and restoration across the tile edge: each rank reconstructs its
    column, receives 16 luma columns of its neighbour's reconstruction, filters its own column (+ the margin the chain
    needs), and the gathered result equals the oracle's whole-frame chain on one rank."""
    import json
    script = tmp_path / "post.py"
    script.write_text(POST_WORKER % {"root": util.ROOT, "tests": os.path.join(util.ROOT, "tests")})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29523", str(script)],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["ok"] and d["filters_changed_pixels"], d
    assert all(0.5 <= s < 0.75 for s in d["share"]), d          # a little more than half the tasks per rank: the margins


DEP_WORKER = """
import sys, json
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
import numpy as np
import torch, torch.distributed as dist
import util, test_frame
from dav1d_amd import api, dist as dd
import synth_frames as synth

rank, local, world = dd.env()
dist.init_process_group("gloo")
W, H, BPC, N = 256, 128, 10, 4
ctx = util.make_context("emu")
rng = np.random.default_rng(31)
ref_host = [synth.make_planes(rng, W, H, BPC) for _ in range(3)]
dst0 = synth.make_planes(rng, W, H, BPC, smooth=False)
frames = [synth.make_frame(W, H, BPC, seed=700 + n, mv_range_px=32) for n in range(N)]
refs = []
for r in ref_host:
    p = dd.SharedPicture(ctx, W, H, api.LAYOUT_I420, BPC, "cpu")
    for pl in range(3):
        p.upload(pl, r[pl])
    refs.append(p)
# frame n is reconstructed by rank n %% world and predicts (reference 0) from frame n - 1, which another rank produced:
# the owner publishes the finished picture, everybody receives it
pics = [dd.SharedPicture(ctx, W, H, api.LAYOUT_I420, BPC, "cpu") for _ in range(N)]
mine = dd.frames_of_rank(N, rank, world)
for n, frame in enumerate(frames):
    cur = pics[n]
    if n in mine:
        for pl in range(3):
            cur.upload(pl, dst0[pl])
        rlist = [pics[n - 1].view if (n and k == 0) else refs[k].view for k in range(3)]
        prep = ctx.buffer(frame.prep_elems * 2); prep.zero()
        coef = ctx.buffer_from(frame.coef)
        ctx.mc_batch(cur.view, rlist, frame.mc, prep)
        if len(frame.comp):
            ctx.comp_batch(cur.view, frame.comp, prep, None)
        ctx.itx_add_batch(cur.view, frame.itx, coef)
        ctx.sync()
    dd.broadcast_picture(cur, n %% world, world)
if rank == 0:
    oracle = util.default_oracle()
    want_prev = None
    for n, frame in enumerate(frames):
        rl = [want_prev if (want_prev is not None and k == 0) else ref_host[k] for k in range(3)]
        want, _, _ = test_frame.oracle_frame(oracle, frame, dst0, rl)
        want_prev = want
    got = [pics[N - 1].download(pl) for pl in range(3)]
    print(json.dumps({"ok": bool(all(np.array_equal(got[pl], want[pl]) for pl in range(3))), "mine": mine}))
dist.barrier()
dist.destroy_process_group()
"""


def test_frame_parallel_dependent_frames_world2_gloo(tmp_path):
    """Frame-parallel sharding where every frame predicts from the previous one (decoded on the other rank): the owner
    broadcasts the finished picture (dist.broadcast_picture), the chain of four frames equals the oracle's."""
    import json
    script = tmp_path / "dep.py"
    script.write_text(DEP_WORKER % {"root": util.ROOT, "tests": os.path.join(util.ROOT, "tests")})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29527", str(script)],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["ok"] and d["mine"] == [0, 2], d


C4_WORKER = """
import sys, json
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
import numpy as np
import torch, torch.distributed as dist
import util, test_frame, test_postchain
from dav1d_amd import api, dist as dd
import synth_frames as synth

rank, local, world = dd.env()
dist.init_process_group("gloo")
W, H, BPC, STEPS = 256, 128, 10, 3
DEP = %(dependent)d
ctx = util.make_context("emu")
rng = np.random.default_rng(41)
ref_host = [synth.make_planes(rng, W, H, BPC) for _ in range(3)]
dst0 = synth.make_planes(rng, W, H, BPC, smooth=False)
# every rank has a frame of its own (BASELINE configs[4]: frames in flight, one per GPU); the lists of all ranks are built everywhere
# so that rank 0 can replay the whole job on the oracle
frames = [synth.make_frame(W, H, BPC, seed=1200 + r, mv_range_px=32) for r in range(world)]
posts = [synth.make_post_filters(frames[r], seed=50 + r) for r in range(world)]
wl = dd.C4Workload(ctx, frames[rank], posts[rank], ref_host, dst0, rank, world, "cpu", bool(DEP))
for s in range(STEPS):
    coef = ctx.buffer_from(frames[rank].coef)
    wl.step(coef)
    ctx.sync()
    coef.free()
res, grn = wl.outputs()
mine = {"res": [p.tolist() for p in res], "grn": [p.tolist() for p in grn]} if False else None
# rank 0 replays every rank's chain on the oracle and compares its own outputs; the other ranks send theirs over
got = [None] * world
dist.all_gather_object(got, (rank, [p.copy() for p in res], [p.copy() for p in grn]))
if rank == 0:
    oracle = util.default_oracle()
    prev = [None] * world                       # restoration output of rank r in the previous step
    for s in range(STEPS):
        cur = []
        for r in range(world):
            rl = list(ref_host)
            if DEP and s:
                rl[0] = prev[(r - 1) %% world]
            rec, _, _ = test_frame.oracle_frame(oracle, frames[r], dst0, rl)
            d, c, rs, g = test_postchain.oracle_post(oracle, posts[r], rec, W, H, BPC)
            cur.append((rs, g))
        prev = [c[0] for c in cur]
    ok = True
    for r, res_r, grn_r in got:
        for pl in range(3):
            vh, vw = (H, W) if pl == 0 else (H // 2, W // 2)
            ok &= bool(np.array_equal(res_r[pl][:vh, :vw], cur[r][0][pl][:vh, :vw]))
            ok &= bool(np.array_equal(grn_r[pl][:vh, :vw], cur[r][1][pl][:vh, :vw]))
    differs = bool(any(not np.array_equal(got[0][1][pl], got[1][1][pl]) for pl in range(3)))
    print(json.dumps({"ok": ok, "ranks_differ": differs, "steps": STEPS, "dependent": DEP}))
dist.barrier()
dist.destroy_process_group()
"""


import pytest


@pytest.mark.parametrize("dependent", [0, 1], ids=["independent", "dependent"])
def test_c4_full_table_and_film_grain_world2_gloo(tmp_path, dependent):
    """BASELINE configs[4] on two CPU ranks (dav1d_amd.dist.C4Workload, what `bench.py --config c4 [--dependent]` times on GPUs): every rank
    takes a frame of its own through reconstruction, deblocking, CDEF, restoration and film grain, step after step; in the dependent
    flavour reference 0 of a rank's frame is the picture the OTHER rank produced one step earlier (broadcast_picture).  Restoration and
    grain outputs of both ranks after three steps equal the oracle's replay of the whole job."""
    import json
    script = tmp_path / "c4.py"
    script.write_text(C4_WORKER % {"root": util.ROOT, "tests": os.path.join(util.ROOT, "tests"), "dependent": dependent})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", str(29531 + dependent), str(script)],
                       capture_output=True, text=True, env=env, timeout=1500)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["ok"] and d["ranks_differ"], d


RCCL_ONE_RANK = """
import sys, ctypes as C
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
import util
from dav1d_amd import api
ctx = util.make_context("hip")
ident = (C.c_uint8 * 128)()
assert ctx.lib.dav1d_hip_peer_unique_id(ident) == 0 and any(ident)
h = C.c_void_p()
rc = ctx.lib.dav1d_hip_peer_open(ctx.h, C.byref(h), ident, 0, 1)
assert rc == 0, rc
assert ctx.lib.dav1d_hip_peer_rank(h) == 0 and ctx.lib.dav1d_hip_peer_world(h) == 1
pic = ctx.picture(256, 128, api.LAYOUT_I420, 10)
x0, x1 = (C.c_int * 1)(0), (C.c_int * 1)(256)
assert ctx.lib.dav1d_hip_peer_broadcast_picture(h, C.byref(pic.pic), 0) == 0
assert ctx.lib.dav1d_hip_peer_allgather_columns(h, C.byref(pic.pic), x0, x1) == 0
assert ctx.lib.dav1d_hip_peer_exchange_halo(h, C.byref(pic.pic), x0, x1, 16) == 0
assert ctx.lib.dav1d_hip_peer_broadcast_picture(h, C.byref(pic.pic), 3) == -22
ctx.lib.dav1d_hip_peer_close(h)
pic.free()
ctx.close()
print("RCCL_ONE_RANK_OK")
"""


@pytest.mark.gpu
def test_peer_opens_on_rccl_with_one_rank(tmp_path):
    """The C peer entry points on the real thing, in a process of their own as every rank of a multi-GPU job is: librccl.so is found and
    resolved at run time, ncclGetUniqueId / ncclCommInitRank work on the GPU box (one rank: the collectives themselves are no-ops), the
    picture calls keep their contracts."""
    script = tmp_path / "one.py"
    script.write_text(RCCL_ONE_RANK % {"root": util.ROOT, "tests": os.path.join(util.ROOT, "tests")})
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_ONE_RANK_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-2500:]


def test_tile_column_split_covers_every_task_once():
    import numpy as np
    from dav1d_amd import dist as dd
    import synth_frames as synth
    frame = synth.make_frame(1024, 256, 8, seed=3)
    stride_px = [g[0] for g in synth.plane_geometry(1024, 256, 8, 1)]
    for n in (1, 2, 3, 4, 8):
        cols = dd.tile_columns(1024, n)
        assert len(cols) == n and cols[0][0] == 0 and cols[-1][1] == 1024
        assert all(a[1] == b[0] for a, b in zip(cols, cols[1:])) and all(x0 % 128 == 0 for x0, _ in cols)
        parts = dd.tasks_by_column(frame.mc, frame.comp, frame.itx, stride_px, cols)
        for k, total in enumerate((len(frame.mc), len(frame.comp), len(frame.itx))):
            idx = np.concatenate([p[k] for p in parts])
            assert len(idx) == total and len(np.unique(idx)) == total
        # destination rectangles stay inside their column
        for c, (mi, ci, ii) in enumerate(parts):
            t = frame.itx[ii]
            x = (t["dst_off"].astype(np.int64) % np.asarray(stride_px)[t["plane"]]) << (t["plane"] > 0)
            assert ((x >= cols[c][0]) & (x < cols[c][1])).all()
    assert dd.uniform_tile_columns(7680, 8) == [(k * 1024, min((k + 1) * 1024, 7680)) for k in range(8)]   # 8,8,...,4 sb128


def test_bench_py_starts_its_own_launcher_and_reports_three_legs_world2_gloo():
    """`python bench.py --gpus 2` with NO launcher around it (it died on an assert): bench.py re-executes itself under
    torch.distributed.run, one process per rank, and the N > 1 line carries three legs — frame-parallel replicas, tile columns with
    in-loop filters (C3: halo exchange + all-gather through dav1d_hip_peer_*), dependent frames with the full table and film grain
    (C4: picture broadcasts) — each with its own parity string against the oracle and the number of ranks the collective saw.
    --emu: the same code on the SIMT-emulated kernels over gloo (tiny frame; the timings mean nothing)."""
    import json
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(util.ROOT, "bench.py"), "--gpus", "2", "--emu", "--width", "256", "--height", "128",
                        "--steps", "2", "--warmup", "2", "--cpu-seconds", "0.1"], capture_output=True, text=True, env=env, timeout=1500, cwd="/tmp")
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["n_ranks_seen"] == 2 and d["scaling"] == "weak"
    legs = d["legs"]
    assert set(legs) == {"replicas", "c3_tile_columns_with_in_loop_filters", "c4_dependent_frames_full_table_film_grain"}
    for name, leg in legs.items():
        assert leg["parity"].startswith("bit-exact"), (name, leg["parity"])
        assert leg["n_ranks_seen"] == 2 and leg["value"] >= 0
    assert legs["c3_tile_columns_with_in_loop_filters"]["scaling"] == "strong"


def test_bench_py_n_gpus_runs_dav1ds_loop_over_the_devices_of_one_process_world2_gloo():
    """`bench.py --gpus N` also reports dav1d's OWN loop over the N devices of one process (dav1d_task_loop_n_gpus: rank 0 runs
    tools/task_loop_n_devices.py in a child, the other ranks wait on a gloo barrier).  Here: two ranks over gloo, the child on two emulated
    devices ($DAV1D_BENCH_N_DEVICES_LEG switches the leg on under --emu)."""
    import json
    import hooked_util as hk
    if hk.lib() is None:
        pytest.skip("needs oracle/_ref_hooked")
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["DAV1D_BENCH_N_DEVICES_LEG"] = "1"
    r = subprocess.run([sys.executable, os.path.join(util.ROOT, "bench.py"), "--gpus", "2", "--emu", "--width", "256", "--height", "128",
                        "--steps", "2", "--warmup", "2", "--cpu-seconds", "0.1"], capture_output=True, text=True, env=env, timeout=1500, cwd="/tmp")
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and len(lines[0]) < 8000, lines
    leg = json.loads(lines[0])["legs"].get("dav1d_task_loop_n_gpus")
    assert leg and leg.get("parity", "").startswith("bit-exact") and leg.get("fps"), leg
    full = json.load(open(os.path.join(util.ROOT, "bench_legs.json")))["dav1d_task_loop_n_gpus"]
    assert full["devices_2"]["devices"]["n"] == 2 and full["devices_1"]["devices"]["n"] == 1, full
    st = full["devices_2"]["devices"]["frames_ended_and_reference_pictures_copied_in_by_device"]
    assert len(st) == 2 and st[0][0] and st[1][0] and st[0][1] + st[1][1] > 0, st


def test_peer_calls_check_their_columns():
    """a column narrower than the halo, odd or overlapping columns, columns outside the plane are refused with -EINVAL
    (they made the strip kernels read or write out of bounds) — on one emulated rank, before anything is exchanged."""
    import ctypes as C
    from dav1d_amd import api
    ctx = util.make_context("emu")
    try:
        ident = (C.c_uint8 * 128)()
        assert ctx.lib.dav1d_hip_peer_unique_id(ident) == 0
        h = C.c_void_p()
        assert ctx.lib.dav1d_hip_peer_open(ctx.h, C.byref(h), ident, 0, 1) == 0
        pic = ctx.picture(256, 128, api.LAYOUT_I420, 10)

        def cols(a, b):
            return (C.c_int * 1)(a), (C.c_int * 1)(b)
        for a, b, halo, want in ((0, 256, 16, 0), (0, 8, 16, -22), (1, 255, 16, -22), (-16, 128, 16, -22), (0, 512, 16, -22), (128, 64, 16, -22)):
            x0, x1 = cols(a, b)
            assert ctx.lib.dav1d_hip_peer_exchange_halo(h, C.byref(pic.pic), x0, x1, halo) == want, (a, b)
            assert ctx.lib.dav1d_hip_peer_allgather_columns(h, C.byref(pic.pic), x0, x1) == (0 if (a, b) == (0, 8) else want), (a, b)
        assert ctx.lib.dav1d_hip_peer_wait(h, 0) == 0 and ctx.lib.dav1d_hip_peer_wait(h, 9) == -22
        ctx.lib.dav1d_hip_peer_close(h)
        pic.free()
    finally:
        ctx.close()
