"""Multi-GPU launch plumbing: one process per GPU.  The data-path collectives — picture broadcast for frames decoded
elsewhere, the all-gather of tile columns, the halo exchange before the in-loop filters — are C entry points of the library
(dav1d_hip_peer_*, csrc/peer.hip, RCCL over xGMI; a shared-memory stand-in in the SIMT-emulated build); torch.distributed (RCCL on GPUs,
gloo in CPU tests) only carries the rendezvous of the peer id, the barrier and the max-over-ranks of the timed region."""
import os


def env():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def frames_of_rank(n_frames, rank, world):
    """Frame n belongs to GPU n mod world (config C4 of SURVEY §8d)."""
    return list(range(rank, n_frames, world))


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()


def max_over_ranks(seconds, world, device="cpu"):
    if world == 1:
        return float(seconds)
    import torch
    import torch.distributed as dist
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def job_throughput(units_per_rank_step, steps, seconds_max, world):
    """Whole-job rate: units all ranks processed / slowest rank's time (weak scaling)."""
    return world * units_per_rank_step * steps / seconds_max


# ---- tile-column mode (config C3 of SURVEY §8d/§8e): GPU g reconstructs tile column g of every frame, one
# all-gather per frame rebuilds the whole picture (the next frame's reference) on every GPU ----------------------------

def uniform_tile_columns(w, n_cols, sb=128):
    """Luma pixel ranges [(x0, x1), ...] of AV1's uniform tile spacing (reference src/obu.c:637-644:
    tile_w = 1 + ((sbw - 1) >> log2_cols) superblocks, the last column takes what is left).  n_cols is rounded up to
    the power of two the syntax can express; fewer columns come back when the picture is too narrow for that many."""
    sbw = (w + sb - 1) // sb
    log2 = max(0, (int(n_cols) - 1).bit_length())
    tile_w = 1 + ((sbw - 1) >> log2)
    return [(sbx * sb, min((sbx + tile_w) * sb, sbw * sb)) for sbx in range(0, sbw, tile_w)]


def tile_columns(w, n_cols, sb=128):
    """One tile column per rank: the uniform spacing when it yields exactly n_cols columns, else explicit widths
    (reference src/obu.c:656-664) as even as superblocks allow."""
    cols = uniform_tile_columns(w, n_cols, sb)
    if len(cols) == n_cols:
        return cols
    sbw = (w + sb - 1) // sb
    if n_cols > sbw:
        raise ValueError("%d tile columns do not fit %d superblock columns" % (n_cols, sbw))
    cuts = [(sbw * k) // n_cols for k in range(n_cols + 1)]
    return [(cuts[k] * sb, cuts[k + 1] * sb) for k in range(n_cols)]


def tasks_by_column(mc, comp, itx, stride_px, cols, ss_hor=1):
    """Index arrays (mc_idx, comp_idx, itx_idx) per tile column.  A task belongs to the column its destination rectangle
    lies in (blocks never straddle a tile edge, reference src/decode.c:2620-2635); a PREP record belongs to the column of
    the compound record that consumes its block of the prep arena."""
    import numpy as np
    edges = np.array([c[0] for c in cols[1:]], np.int64)
    stride = np.asarray(stride_px, np.int64)

    def col(off, plane):
        x = (off.astype(np.int64) % stride[plane]) << np.where(plane > 0, ss_hor, 0)
        return np.searchsorted(edges, x, side="right")

    itx_col = col(itx["dst_off"], itx["plane"])
    comp_col = col(comp["dst_off"], comp["plane"]) if len(comp) else np.zeros(0, np.int64)
    mc_col = col(mc["dst_off"], mc["plane"])
    tmp = mc["kind"] != 0                                  # PREP / PUT_TMP write the arena, not the picture
    if tmp.any():
        offs = np.concatenate([comp["tmp1_off"], comp["tmp2_off"]]).astype(np.int64)
        owner = np.concatenate([comp_col, comp_col])
        order = np.argsort(offs, kind="stable")
        at = np.searchsorted(offs[order], mc["dst_off"][tmp].astype(np.int64))
        if (at >= len(offs)).any() or (offs[order][np.minimum(at, len(offs) - 1)] != mc["dst_off"][tmp]).any():
            raise ValueError("a PREP record is not consumed by any compound record of this frame")
        mc_col[tmp] = owner[order][at]
    return [(np.flatnonzero(mc_col == c), np.flatnonzero(comp_col == c), np.flatnonzero(itx_col == c)) for c in range(len(cols))]


class SharedPicture:
    """Picture planes held in ONE torch tensor (so that torch.distributed can move them) and described to the C ABI by
    an ordinary Dav1dHipPicture pointing into it — the caller owns picture memory at this boundary
    (include/dav1d_hip.h, Dav1dHipPicture).  Geometry = the library's own (reference src/picture.c:46-78)."""

    def __init__(self, ctx, w, h, layout, bpc, device):
        import ctypes as C
        import torch
        from . import api
        probe = ctx.picture(w, h, layout, bpc)             # ask the library for strides / padded heights, then let go
        self.w, self.h, self.layout, self.bpc = w, h, layout, bpc
        self.tdtype = torch.uint8 if bpc == 8 else torch.int16      # int16 carries the uint16 bit patterns
        bps = 1 if bpc == 8 else 2
        geo = []
        for pl in range(probe.n_planes):
            rows, cols_ = probe.padded_shape(pl)
            geo.append((rows, cols_, probe.pic.p[pl].stride // bps, probe.pic.p[pl].w, probe.pic.p[pl].h))
        probe.free()
        total = sum(g[0] * g[2] for g in geo) + 64
        self.store = torch.zeros(total, dtype=self.tdtype, device=device)
        base = self.store.data_ptr()
        skip = (-base) % 64 // bps                          # DAV1D_PICTURE_ALIGNMENT
        self.planes, off = [], skip
        self.pic = api.Picture()
        self.pic.bpc, self.pic.layout = bpc, layout
        for pl, (rows, cols_, stride, vw, vh) in enumerate(geo):
            t = self.store[off:off + rows * stride].view(rows, stride)
            self.planes.append(t)
            self.pic.p[pl].data = t.data_ptr()
            self.pic.p[pl].stride = stride * bps
            self.pic.p[pl].w, self.pic.p[pl].h = vw, vh
            off += rows * stride
        self.view = api.DevicePicture.view(ctx, self.pic, w, h, layout, bpc)

    def upload(self, plane, arr):
        import torch
        t = torch.from_numpy(arr.view("int16") if self.bpc > 8 else arr)
        self.planes[plane][:, :t.shape[1]].copy_(t)

    def download(self, plane):
        a = self.planes[plane].cpu().numpy()
        a = a.view("uint16") if self.bpc > 8 else a
        return a[:, :self.view.padded_shape(plane)[1]]


# ---- the data-path collectives live behind the C ABI (csrc/peer.hip: dav1d_hip_peer_* on RCCL; one Dav1dHipPeer per process) -------
# What stays here is launch plumbing: the rendezvous of the 128-byte id over torch.distributed and thin wrappers.

_peers = {}


def peer_of(ctx, rank, world):
    """The process's Dav1dHipPeer for this context (made on first use: rank 0's id travels through torch.distributed once)."""
    import ctypes as C
    key = (id(ctx), rank, world)
    p = _peers.get(key)
    if p is None:
        import torch.distributed as dist
        ident = (C.c_uint8 * 128)()
        if rank == 0:
            rc = ctx.lib.dav1d_hip_peer_unique_id(ident)
            if rc:
                raise RuntimeError("dav1d_hip_peer_unique_id: %d" % rc)
        box = [bytes(ident)]
        dist.broadcast_object_list(box, src=0)
        ident = (C.c_uint8 * 128).from_buffer_copy(box[0])
        h = C.c_void_p()
        rc = ctx.lib.dav1d_hip_peer_open(ctx.h, C.byref(h), ident, rank, world)
        if rc:
            raise RuntimeError("dav1d_hip_peer_open: %d" % rc)
        p = h
        _peers[key] = (h, ctx)
        return p
    return p[0]


def close_peers():
    """dav1d_hip_peer_close for every peer this process opened (before the contexts go)"""
    for h, ctx in list(_peers.values()):
        ctx.lib.dav1d_hip_peer_close(h)
    _peers.clear()


def _pic_of(pic):
    return pic.pic if hasattr(pic, "pic") else pic


def _ctx_of(pic):
    return pic.view.ctx if hasattr(pic, "view") else pic.ctx


def wait_gathers(pic, rank, world, lag=0):
    """dav1d_hip_peer_wait: the context's stream waits for the overlapped gathers issued so far (all but the `lag` most recent)"""
    if world == 1:
        return
    ctx = _ctx_of(pic)
    rc = ctx.lib.dav1d_hip_peer_wait(peer_of(ctx, rank, world), lag)
    if rc:
        raise RuntimeError("dav1d_hip_peer_wait: %d" % rc)


def allgather_tile_columns(pic, cols, rank, world, ss_hor=1, overlap=False):
    """After rank g reconstructed column g of `pic`: ONE all-gather per frame (all planes of a strip packed into one message by a
    strided copy kernel, strips padded to the widest column — SURVEY 8e) and every rank holds the whole picture
    (dav1d_hip_peer_allgather_columns)."""
    import ctypes as C
    if world == 1:
        return
    assert len(cols) == world, "one tile column per rank"
    ctx = _ctx_of(pic)
    x0 = (C.c_int * world)(*[c[0] for c in cols])
    x1 = (C.c_int * world)(*[c[1] for c in cols])
    # overlap: on the peer's side stream, next to what is enqueued afterwards (the next frame's reconstruction); wait_gathers() before
    # anything reads the gathered picture
    fn = ctx.lib.dav1d_hip_peer_allgather_columns_async if overlap else ctx.lib.dav1d_hip_peer_allgather_columns
    rc = fn(peer_of(ctx, rank, world), C.byref(_pic_of(pic)), x0, x1)
    if rc:
        raise RuntimeError("dav1d_hip_peer_allgather_columns: %d" % rc)


# ---- tile-column mode, in-loop filters: deblocking, CDEF and loop restoration read across the tile edge ------------
#
# What rank g needs from its neighbours so that everything it produces INSIDE its column [x0, x1) is exact:
#   loop restoration at x >= x0 reads CDEF output from x0 - 3;
#   CDEF output at x0 - 3 lies in the 8x8 unit [x0 - 8, x0): its direction search reads the unit's deblocked pixels, its taps
#     deblocked pixels from x0 - 5;
#   deblocked pixels in [x0 - 8, x0) come from the vertical edges at x0 - 8 (at most the 8-wide filter there: a 16-wide one needs
#     16-pixel transforms on both sides, i.e. an edge at a multiple of 16), x0 - 4 and x0, which read reconstructed pixels from
#     x0 - 12 on; horizontal edges work down a column and spread nothing sideways.
# So HALO = 16 luma columns (8 chroma) of RECONSTRUCTED pixels per side are exchanged once per frame, every rank then runs the
# filter tasks that touch [x0 - 8, x1 + 8) (deblocking, CDEF) or are cropped to [x0, x1) (restoration) — a few per cent of work
# done twice instead of a second and third exchange — and the final all-gather moves finished columns only.
HALO = 16

def exchange_halo(pic, cols, rank, world, ss_hor=1, halo=HALO):
    """After rank g reconstructed column g of `pic`: its neighbours' outermost `halo` luma columns (all planes) arrive next to it
    (dav1d_hip_peer_exchange_halo: neighbour to neighbour, an ncclSend / ncclRecv pair per side inside one group; 16 columns of an 8K
    frame are 0.2 MB per side)."""
    import ctypes as C
    if world == 1:
        return
    assert len(cols) == world
    ctx = _ctx_of(pic)
    x0 = (C.c_int * world)(*[c[0] for c in cols])
    x1 = (C.c_int * world)(*[c[1] for c in cols])
    rc = ctx.lib.dav1d_hip_peer_exchange_halo(peer_of(ctx, rank, world), C.byref(_pic_of(pic)), x0, x1, halo)
    if rc:
        raise RuntimeError("dav1d_hip_peer_exchange_halo: %d" % rc)


def post_tasks_of_column(lf, cdef, lr, stride_px, col, ss_hor=1):
    """The in-loop filter tasks rank g runs for column `col` = (x0, x1) (see the derivation above): deblocking and CDEF tasks
    that touch [x0 - 8, x1 + 8), restoration units cropped to [x0, x1).  Returns (lf tasks, cdef tasks, lr tasks)."""
    import numpy as np
    x0, x1 = col
    stride = np.asarray(stride_px, np.int64)
    pl = lf["plane"].astype(np.int64)
    sh = np.where(pl > 0, ss_hor, 0)
    x = (lf["dst_off"].astype(np.int64) % stride[pl]) << sh              # luma x of the task's first unit
    span = np.where(lf["dir"] == 0, 1, 128 << sh)                        # dir 1: a line of up to 32 units runs across
    keep = (x + span > x0 - 9) & (x <= x1 + 8)               # edges at x0 - 8 .. x1 + 8; lines that cross that range
    lf_c = lf[keep]
    cx = cdef["bx"].astype(np.int64) * 8
    raw = (cdef["flags"] & 1) != 0
    cdef_c = cdef[~raw & (cx + 8 > x0 - 8) & (cx < x1 + 8)]
    out = []
    for t in lr:
        s = ss_hor if t["plane"] else 0
        a, b = x0 >> s, x1 >> s
        lo, hi = max(int(t["x"]), a), min(int(t["x"]) + int(t["w"]), b)
        if lo >= hi:
            continue
        u = t.copy()
        e = int(t["edges"])
        if lo > int(t["x"]):
            e |= 1                                                       # cropped on the left: the pixels beyond exist
        if hi < int(t["x"]) + int(t["w"]):
            e |= 2
        u["x"], u["w"], u["edges"] = lo, hi - lo, e
        out.append(u)
    lr_c = np.array(out, dtype=lr.dtype) if out else lr[:0]
    return lf_c, cdef_c, lr_c


# ---- frame-parallel mode with dependent frames ------------------------------------------------------------------------------------
# dav1d's frame threads let frame n + 1 start while frame n is still being decoded and make each of its blocks wait until the
# reference rows it reads have been published (src/thread_task.c:416-433, progress per superblock row).  Across GPUs the unit
# of publication is the whole picture: the rank that finished frame n sends it once, every rank that predicts from it receives
# it into its own copy (RCCL broadcast over xGMI; one 8K 10-bit picture is 100 MB = about 0.7 ms per link).

def broadcast_picture(pic, src_rank, world, rank=None):
    """Every rank ends up with rank `src_rank`'s planes of `pic` (dav1d_hip_peer_broadcast_picture: one ncclBroadcast for a picture of
    the library's allocator, one per plane for caller-owned planes)."""
    import ctypes as C
    if world == 1:
        return
    ctx = _ctx_of(pic)
    if rank is None:
        rank = env()[0]
    rc = ctx.lib.dav1d_hip_peer_broadcast_picture(peer_of(ctx, rank, world), C.byref(_pic_of(pic)), src_rank)
    if rc:
        raise RuntimeError("dav1d_hip_peer_broadcast_picture: %d" % rc)


# ---- config C4 of SURVEY 8d / BASELINE configs[4]: frames in flight one per GPU, the FULL table (reconstruction, deblocking, CDEF,
# restoration) and film grain on every frame.  One step = every rank takes its next frame through all of it.  Two flavours:
#   independent (closed GOPs): no data-path collective at all;
#   dependent: reference 0 of a rank's frame in step s is the picture rank - 1 produced in step s - 1 (the publication rule of
#   src/thread_task.c:416-433 at picture granularity): after its frame is through, every owner broadcasts the picture later frames
#   predict from (the restoration output; film grain is output-only, src/lib.c:311-329) and every rank keeps a copy.
# bench.py --config c4 [--dependent] runs it on GPUs over RCCL, tests/test_dist.py on two CPU ranks over gloo (SIMT-emulated kernels).

class C4Workload:
    def __init__(self, ctx, frame, post, ref_host, dst_host, rank, world, device, dependent):
        from . import api
        self.ctx, self.frame, self.post, self.rank, self.world, self.dependent = ctx, frame, post, rank, world, dependent
        w, h, bpc = frame.w, frame.h, frame.bpc
        self.w, self.h, self.bpc = w, h, bpc

        def pic():
            return SharedPicture(ctx, w, h, api.LAYOUT_I420, bpc, device)
        self.refs = []
        for rp in ref_host:
            p = pic()
            for pl in range(3):
                p.upload(pl, rp[pl])
            self.refs.append(p)
        self.cur, self.cdf, self.grn = pic(), pic(), pic()
        if dst_host is not None:           # what lies in the picture before the frame (every visible pixel is written by the frame)
            for pl in range(3):
                self.cur.upload(pl, dst_host[pl])
        # the restoration outputs: one per owner in dependent mode (ring[o] = the latest picture of rank o, kept on every rank)
        self.ring = [pic() for _ in range(world if dependent else 1)]
        self.have_prev = False
        self.lvl = ctx.buffer_from(post.lvl)
        self.prep = ctx.buffer(frame.prep_elems * 2 + 64)
        self.recon = ctx.recon_list(self.cur.view, frame.mc, frame.comp, frame.itx)
        self.coefs = None

    def refs_now(self):
        views = [r.view for r in self.refs]
        if self.dependent and self.have_prev:
            views[0] = self.ring[(self.rank - 1) % self.world].view
        return views

    def step(self, coef):
        """coef: a pristine DEVICE coefficient arena for this step (pointer or Buffer)"""
        ctx, post = self.ctx, self.post
        cur, cdf = self.cur, self.cdf
        res = self.ring[self.rank if self.dependent else 0]
        self.prep.zero()
        self.recon.run(cur.view, self.refs_now(), self.prep, coef)
        ctx.lf_batch(cur.view, post.lf, self.lvl, post.b4_stride, post.lut_e, post.lut_i)
        for pl in range(3):
            cdf.planes[pl].copy_(cur.planes[pl])
        ctx.cdef_batch(cdf.view, cur.view, post.cdef, post.cdef_damping)
        for pl in range(3):
            res.planes[pl].copy_(cdf.planes[pl])
        ctx.lr_batch(res.view, cdf.view, cur.view, post.lr)
        ctx.fg_apply(self.grn.view, res.view, post.fg)
        if self.dependent:
            for o in range(self.world):
                broadcast_picture(self.ring[o], o, self.world)
            self.have_prev = True

    def outputs(self):
        """(restored planes, grain planes) of this rank's latest frame"""
        res = self.ring[self.rank if self.dependent else 0]
        return [res.download(pl) for pl in range(3)], [self.grn.download(pl) for pl in range(3)]
