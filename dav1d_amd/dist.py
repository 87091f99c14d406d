"""Multi-GPU plumbing of the frame-parallel mode: one process per GPU, frames
sharded round-robin, no data-path collective.  torch.distributed (RCCL on GPUs, gloo in CPU tests)
carries only the barrier and the max-over-ranks of the timed region."""
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


def allgather_tile_columns(pic, cols, rank, world, ss_hor=1):
    """After rank g reconstructed column g of `pic`: ONE all-gather per frame (all planes of a strip packed into one
    message, strips padded to the widest column — SURVEY §8e) and every rank holds the whole picture."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return
    assert len(cols) == world, "one tile column per rank"
    wmax = max(x1 - x0 for x0, x1 in cols)
    shapes = [(p.shape[0], wmax >> (ss_hor if pl else 0)) for pl, p in enumerate(pic.planes)]
    per = sum(r * c for r, c in shapes)
    send = torch.zeros(per, dtype=pic.tdtype, device=pic.store.device)

    def strips(buf, k):
        out, off = [], 0
        x0, x1 = cols[k]
        for pl, (r, c) in enumerate(shapes):
            s = ss_hor if pl else 0
            out.append((buf[off:off + r * c].view(r, c)[:, :(x1 - x0) >> s], x0 >> s, x1 >> s))
            off += r * c
        return out

    for pl, (v, a, b) in enumerate(strips(send, rank)):
        v.copy_(pic.planes[pl][:, a:b])
    recv = torch.empty(world * per, dtype=pic.tdtype, device=pic.store.device)
    dist.all_gather_into_tensor(recv.view(torch.uint8), send.view(torch.uint8))    # bytes: every backend moves uint8
    for k in range(world):
        if k == rank:
            continue
        for pl, (v, a, b) in enumerate(strips(recv[k * per:(k + 1) * per], k)):
            pic.planes[pl][:, a:b].copy_(v)
