"""Wall clock against device time of the batch calls of one 8K frame's in-loop filters (where the host side of a frame goes)."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
from dav1d_amd import api
import synth_frames as synth

w, h, bpc = 7680, 4320, 10
ctx = api.Context(0)
frame = synth.make_frame(w, h, bpc, seed=1, n_refs=1)
post = synth.make_post_filters(frame, seed=2)
rng = np.random.default_rng(3)
planes = synth.make_planes(rng, w, h, bpc)
pics = [ctx.picture(w, h, api.LAYOUT_I420, bpc) for _ in range(4)]
for pl in range(3):
    pics[0].upload(pl, planes[pl])
lvl = ctx.buffer_from(post.lvl)
g = ctx.fg_prepare(post.fg, bpc, api.LAYOUT_I420)
ctx.sync()
for rep in range(3):
    rows = []
    for name, fn in (("lf", lambda: ctx.lf_batch(pics[0], post.lf, lvl, post.b4_stride, post.lut_e, post.lut_i)),
                     ("cdef", lambda: ctx.cdef_batch(pics[1], pics[0], post.cdef, post.cdef_damping)),
                     ("lr", lambda: ctx.lr_batch(pics[2], pics[1], pics[0], post.lr)),
                     ("fg", lambda: ctx.fg_apply_prepared(pics[3], pics[2], g))):
        t0 = time.perf_counter()
        fn()
        wall = (time.perf_counter() - t0) * 1e3
        rows.append("%s wall %.3f dev %.3f" % (name, wall, ctx.last_kernel_ms()))
    print(" | ".join(rows))
# raw upload rates: pageable numpy -> device
for mb in (1, 8):
    a = np.zeros(mb << 20, np.uint8)
    b = ctx.buffer(mb << 20)
    t0 = time.perf_counter()
    for _ in range(5):
        b.upload(a)
    print("upload %d MiB pageable: %.3f ms" % (mb, (time.perf_counter() - t0) / 5 * 1e3))
