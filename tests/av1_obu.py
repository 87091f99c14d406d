"""AV1 bitstreams for dav1d's REAL front end.  TEST INFRASTRUCTURE (tests/, bench.py's checker legs): nothing in dav1d_amd/ imports it.

There are no AV1 streams and no encoder in either box, but an arithmetic decoder turns ANY byte string into valid syntax (that is
how the reference's own fuzzer works, tests/libfuzzer/dav1d_fuzzer.c).  This module writes the part of a stream that is not
arithmetic-coded — temporal delimiters, the sequence header, frame headers with every coding tool switched on at random, tile
group framing — bit by bit in the order the reference parses it (src/obu.c: parse_seq_hdr :71-303, parse_frame_hdr :406-1151,
parse_tile_hdr :1153-1167, dav1d_parse_obus :1169 ff.; bit readers src/getbits.c), and fills the tile payloads with seeded random
bytes.  dav1d's own dav1d_parse_obus / dav1d_submit_frame / msac / decode_b then produce Av1Block, cbi, cf, palettes, the
deblocking masks and levels, cdef indices, restoration units, refmvs, segmentation maps and CDF updates exactly as for a real
stream; what the symbols decode to follows the default CDFs, i.e. the statistics of real content.

The writer keeps the decoder-side state a header depends on (the eight reference slots: order hints, sizes, global motion
parameters, loop filter deltas, segmentation data, film grain) so that every conditional syntax element is present exactly when
the parser expects it.  A payload the decoder rejects (4:2:2 forbids some partitions, an intra block copy can point nowhere,
src/decode.c:2150-2156, 1337-1341) is repaired by the caller: tests/stream_util.py re-rolls the bytes from where the tile failed."""
import numpy as np

KEY, INTER, INTRA_ONLY, SWITCH = 0, 1, 2, 3
OBU_SEQ_HDR, OBU_TD, OBU_FRAME_HDR, OBU_TILE_GRP, OBU_FRAME = 1, 2, 3, 4, 6
LAYOUT_I400, LAYOUT_I420, LAYOUT_I422, LAYOUT_I444 = 0, 1, 2, 3
PRIMARY_REF_NONE = 7
WM_IDENTITY, WM_TRANSLATION, WM_ROT_ZOOM, WM_AFFINE = 0, 1, 2, 3
RESTORATION_NONE, RESTORATION_SWITCHABLE, RESTORATION_WIENER, RESTORATION_SGRPROJ = 0, 1, 2, 3      # enum Dav1dRestorationType == the 2 raw bits


class BitWriter:
    def __init__(self):
        self.acc, self.n = 0, 0

    def f(self, n, v):
        """n bits, most significant first (dav1d_get_bits)"""
        v = int(v)
        assert n >= 0 and 0 <= v < (1 << n), (n, v)
        self.acc = (self.acc << n) | v
        self.n += n

    def bit(self, v):
        self.f(1, 1 if v else 0)

    def su(self, n, v):
        """n bits, two's complement (dav1d_get_sbits(gb, n))"""
        assert -(1 << (n - 1)) <= v < (1 << (n - 1)), (n, v)
        self.f(n, v & ((1 << n) - 1))

    def uniform(self, mx, v):
        """dav1d_get_uniform(gb, mx): v in [0, mx), mx > 1"""
        assert mx > 1 and 0 <= v < mx
        l = mx.bit_length()
        m = (1 << l) - mx
        if v < m:
            self.f(l - 1, v)
        else:
            x = v + m
            self.f(l - 1, x >> 1)
            self.f(1, x & 1)

    def _subexp_u(self, x, ref, n):
        def recenter(r, x):
            return x if x > 2 * r else (2 * (x - r) if x >= r else 2 * (r - x) - 1)
        v = recenter(ref, x) if ref * 2 <= n else recenter(n - ref, n - x)
        mk, i = 0, 0
        while True:
            b = 3 if i == 0 else 3 + i - 1
            if n < mk + 3 * (1 << b):
                self.uniform(n - mk + 1, v - mk)
                return
            if v < mk + (1 << b):
                self.bit(0)
                self.f(b, v - mk)
                return
            self.bit(1)
            mk += 1 << b
            i += 1

    def subexp(self, x, ref, n):
        """dav1d_get_bits_subexp(gb, ref, n) == x, x and ref in [-(1 << n), 1 << n]"""
        assert -(1 << n) <= x <= (1 << n) and -(1 << n) <= ref <= (1 << n), (x, ref, n)
        self._subexp_u(x + (1 << n), ref + (1 << n), 2 << n)

    def align(self):
        if self.n & 7:
            self.f(8 - (self.n & 7), 0)

    def trailing(self):
        self.bit(1)
        self.align()

    def tobytes(self):
        assert not self.n & 7
        return self.acc.to_bytes(self.n >> 3, "big") if self.n else b""


def leb128(v):
    out = bytearray()
    while True:
        b = v & 0x7f
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def obu(obu_type, payload):
    return bytes([obu_type << 3 | 2]) + leb128(len(payload)) + payload          # forbidden 0, type, no extension, has_size_field, reserved 0


def tile_log2(sz, tgt):
    k = 0
    while (sz << k) < tgt:
        k += 1
    return k


def poc_diff(bits, a, b):
    """get_poc_diff, reference src/env.h"""
    m = 1 << (bits - 1)
    d = a - b
    return (d & (m - 1)) - (d & m)


class Seq:
    """The sequence header's choices; every tool the frame headers may use is enabled here."""

    def __init__(self, w, h, layout=LAYOUT_I420, bpc=8, sb128=True, film_grain=True, super_res=True, screen_content=2, cdef=True,
                 restoration=True, order_hint_bits=7, filter_intra=True, intra_edge_filter=True, separate_uv_delta_q=True):
        self.max_w, self.max_h, self.layout, self.bpc, self.sb128 = w, h, layout, bpc, int(sb128)
        self.hbd = {8: 0, 10: 1, 12: 2}[bpc]
        self.monochrome = int(layout == LAYOUT_I400)
        self.profile = 1 if layout == LAYOUT_I444 and bpc < 12 else 2 if (layout in (LAYOUT_I422, LAYOUT_I444) or bpc == 12) else 0
        self.ss_hor = int(layout != LAYOUT_I444)
        self.ss_ver = int(layout in (LAYOUT_I420, LAYOUT_I400))
        self.film_grain, self.super_res, self.cdef, self.restoration = int(film_grain), int(super_res), int(cdef), int(restoration)
        self.screen_content_tools = screen_content          # 0, 1, 2 = adaptive (per frame)
        self.force_integer_mv = 2                           # adaptive
        self.order_hint_bits = order_hint_bits
        self.filter_intra, self.intra_edge_filter = int(filter_intra), int(intra_edge_filter)
        self.inter_intra = self.masked_compound = self.warped_motion = self.dual_filter = 1
        self.jnt_comp = self.ref_frame_mvs = 1
        self.separate_uv_delta_q = int(separate_uv_delta_q and not self.monochrome)
        self.width_bits = max(1, (w - 1).bit_length())
        self.height_bits = max(1, (h - 1).bit_length())

    def write(self):
        b = BitWriter()
        b.f(3, self.profile)
        b.bit(0)                      # still_picture
        b.bit(0)                      # reduced_still_picture_header
        b.bit(0)                      # timing_info_present
        b.bit(0)                      # display_model_info_present
        b.f(5, 0)                     # operating_points_cnt_minus_1
        b.f(12, 0)                    # idc
        b.f(3, 7)                     # major level - 2 -> 9 (> 3: a tier bit follows)
        b.f(2, 3)
        b.bit(0)                      # tier
        b.f(4, self.width_bits - 1)
        b.f(4, self.height_bits - 1)
        b.f(self.width_bits, self.max_w - 1)
        b.f(self.height_bits, self.max_h - 1)
        b.bit(0)                      # frame_id_numbers_present
        b.bit(self.sb128)
        b.bit(self.filter_intra)
        b.bit(self.intra_edge_filter)
        b.bit(self.inter_intra)
        b.bit(self.masked_compound)
        b.bit(self.warped_motion)
        b.bit(self.dual_filter)
        b.bit(1)                      # order_hint
        b.bit(self.jnt_comp)
        b.bit(self.ref_frame_mvs)
        if self.screen_content_tools == 2:
            b.bit(1)
        else:
            b.bit(0)
            b.bit(self.screen_content_tools)
        if self.screen_content_tools:
            b.bit(1)                  # force_integer_mv: adaptive
        b.f(3, self.order_hint_bits - 1)
        b.bit(self.super_res)
        b.bit(self.cdef)
        b.bit(self.restoration)
        # color config
        b.bit(1 if self.hbd else 0)
        if self.profile == 2 and self.hbd:
            b.bit(self.hbd == 2)
        if self.profile != 1:
            b.bit(self.monochrome)
        b.bit(0)                      # color_description_present
        if self.monochrome:
            b.bit(0)                  # color_range
        else:
            b.bit(0)                  # color_range
            if self.profile == 2 and self.hbd == 2:
                b.bit(self.ss_hor)
                if self.ss_hor:
                    b.bit(self.ss_ver)
            if self.ss_hor and self.ss_ver:
                b.f(2, 0)             # chroma sample position
        if not self.monochrome:
            b.bit(self.separate_uv_delta_q)
        b.bit(self.film_grain)
        b.trailing()
        return b.tobytes()


class Slot:
    """what dav1d keeps of the frame in a reference slot (c->refs[i].p.p.frame_hdr) and a later header depends on"""

    def __init__(self):
        self.valid = False
        self.frame_offset = 0
        self.w1 = self.h = self.render_w = self.render_h = 0          # width[1] (upscaled), height
        self.frame_type = KEY
        self.gmv = [(WM_IDENTITY, [0, 0, 1 << 16, 0, 0, 1 << 16])] * 7
        self.ref_delta = [1, 0, 0, 0, -1, 0, -1, -1]
        self.mode_delta = [0, 0]
        self.seg = None               # segmentation data (list of 8 dicts) or None
        self.grain = None
        self.showable = False


DEFAULT_GMV = (WM_IDENTITY, [0, 0, 1 << 16, 0, 0, 1 << 16])


class Knobs:
    """probabilities (0..1) of the per-frame header choices; the defaults switch everything on some of the time"""

    def __init__(self, **kw):
        self.intra_only = 0.08
        self.hidden = 0.12                 # show_frame = 0, shown later by show_existing_frame
        self.super_res = 0.2
        self.scaled = 0.15                 # inter frame of another size than its references
        self.screen_content = 0.3
        self.intrabc = 0.5                 # of intra frames with screen content tools
        self.force_integer_mv = 0.1
        self.segmentation = 0.4
        self.delta_q = 0.4
        self.delta_lf = 0.6
        self.lossless = 0.06               # base_q_idx = 0 with no deltas
        self.qm = 0.3
        self.lf_off = 0.1
        self.lf_delta_update = 0.4
        self.cdef = 0.85
        self.restoration = 0.8
        self.tx_largest = 0.15
        self.reduced_txtp_set = 0.2
        self.single_ref = 0.15             # switchable_comp_refs = 0
        self.gmv = 0.35
        self.film_grain = 0.6
        self.disable_cdf_update = 0.1
        self.primary_ref = 0.6             # inherit CDFs / deltas / segmentation / gmv from a reference
        self.error_resilient = 0.1
        self.non_uniform_tiles = 0.2
        self.short_signaling = 0.1
        self.fixed_filter = 0.2
        self.no_switchable_motion = 0.15
        self.no_hp = 0.3
        self.no_warp = 0.1
        self.use_ref_frame_mvs = 0.6
        self.max_tiles = (2, 2)            # log2 of the most tile columns / rows a frame gets
        self.bytes_per_pixel = 1.0         # tile payload (the decoder must never run dry: src/decode.c:2743)
        self.seg_pin = 0                   # inter frames: segmentation on with this many of the 8 segments carrying the reference feature
                                           # (their blocks are inter blocks of that reference whatever the payload says, src/decode.c:
                                           # 1241-1262): random payloads alone decode to ~3/4 intra blocks (the intra flag's context feeds itself)
        self.__dict__.update(kw)


class Frame:
    """one frame header's decisions (filled by StreamWriter.next_frame; tests read them for the histogram)"""
    pass


class StreamWriter:
    def __init__(self, seq, seed, knobs=None):
        self.seq, self.k = seq, knobs or Knobs()
        self.rng = np.random.default_rng(seed)
        self.slots = [Slot() for _ in range(8)]
        self.n = 0                     # frames written (decode order)
        self.order = 0                 # display order counter
        self.hidden = []               # slots holding frames not shown yet
        self.units = []                # temporal units: dict(data=bytearray, tiles=[(offset, size)], frame=Frame or None)

    # ---- random helpers
    def p(self, prob):
        return bool(self.rng.random() < prob)

    def ri(self, lo, hi):
        return int(self.rng.integers(lo, hi + 1))

    # ---- header pieces
    def _frame_size(self, b, fr):
        seq = self.seq
        if fr.frame_size_override:
            b.f(seq.width_bits, fr.w1 - 1)
            b.f(seq.height_bits, fr.h - 1)
        self._superres(b, fr)
        b.bit(0)                       # have_render_size

    def _superres(self, b, fr):
        if self.seq.super_res:
            b.bit(fr.super_res)
            if fr.super_res:
                b.f(3, fr.sr_denom - 9)

    def _pick_size(self, fr, refs_wh):
        """width[1] / height / super-resolution of the frame; refs_wh: (w, h) of the references an inter frame predicts from"""
        seq, k = self.seq, self.k
        fr.frame_size_override = 0
        fr.w1, fr.h = seq.max_w, seq.max_h
        if refs_wh is not None and self.p(k.scaled) and not fr.error_resilient:
            for _ in range(8):
                w = max(16, int(seq.max_w * self.rng.uniform(0.55, 1.0)))
                h = max(16, int(seq.max_h * self.rng.uniform(0.55, 1.0)))
                if all(2 * w >= rw and 2 * h >= rh and w <= 16 * rw and h <= 16 * rh for rw, rh in refs_wh):
                    fr.frame_size_override, fr.w1, fr.h = 1, w, h
                    break
        fr.super_res, fr.sr_denom = 0, 8
        fr.w0 = fr.w1
        if seq.super_res and self.p(k.super_res) and not fr.allow_intrabc_wanted:
            d = self.ri(9, 16)
            w0 = max((fr.w1 * 8 + (d >> 1)) // d, min(16, fr.w1))
            ok = refs_wh is None or all(2 * w0 >= rw and w0 <= 16 * rw for rw, _ in refs_wh)
            if ok and w0 != fr.w1:
                fr.super_res, fr.sr_denom, fr.w0 = 1, d, w0
        if refs_wh is not None and not fr.frame_size_override:
            # the sequence's size against references of another size
            if not all(2 * fr.w0 >= rw and 2 * fr.h >= rh and fr.w0 <= 16 * rw and fr.h <= 16 * rh for rw, rh in refs_wh):
                fr.super_res, fr.sr_denom, fr.w0 = 0, 8, fr.w1

    def _tiles(self, b, fr):
        seq, k = self.seq, self.k
        sb_log2 = 6 + seq.sb128
        sbw = (fr.w0 + (1 << sb_log2) - 1) >> sb_log2
        sbh = (fr.h + (1 << sb_log2) - 1) >> sb_log2
        max_tile_width_sb = 4096 >> sb_log2
        max_tile_area_sb = 4096 * 2304 >> (2 * sb_log2)
        min_log2_cols = tile_log2(max_tile_width_sb, sbw)
        max_log2_cols = tile_log2(1, min(sbw, 64))
        max_log2_rows = tile_log2(1, min(sbh, 64))
        min_log2_tiles = max(tile_log2(max_tile_area_sb, sbw * sbh), min_log2_cols)
        uniform = not self.p(k.non_uniform_tiles)
        want_c, want_r = getattr(fr, "want_tiles_log2", (self.ri(0, k.max_tiles[0]), self.ri(0, k.max_tiles[1])))
        b.bit(uniform)
        if uniform:
            log2_cols = min(max(want_c, min_log2_cols), max_log2_cols)
            for _ in range(min_log2_cols, log2_cols):
                b.bit(1)
            if log2_cols < max_log2_cols:
                b.bit(0)
            tile_w = 1 + ((sbw - 1) >> log2_cols)
            cols = list(range(0, sbw, tile_w))
            min_log2_rows = max(min_log2_tiles - log2_cols, 0)
            log2_rows = min(max(want_r, min_log2_rows), max_log2_rows)
            for _ in range(min_log2_rows, log2_rows):
                b.bit(1)
            if log2_rows < max_log2_rows:
                b.bit(0)
            tile_h = 1 + ((sbh - 1) >> log2_rows)
            rows = list(range(0, sbh, tile_h))
        else:
            cols, sbx, widest = [], 0, 0
            n_want = 1 << want_c
            while sbx < sbw and len(cols) < 64:
                mx = min(sbw - sbx, max_tile_width_sb)
                left_tiles = max(1, n_want - len(cols))
                tw = 1 if mx <= 1 else min(mx, max(1, self.ri(1, max(1, 2 * (sbw - sbx) // left_tiles))))
                if len(cols) + 1 >= n_want:
                    tw = mx if (sbw - sbx) <= max_tile_width_sb else tw
                if mx > 1:
                    b.uniform(mx, tw - 1)
                cols.append(sbx)
                sbx += tw
                widest = max(widest, tw)
            log2_cols = tile_log2(1, len(cols))
            area = sbw * sbh
            if min_log2_tiles:
                area >>= min_log2_tiles + 1
            max_tile_height_sb = max(area // widest, 1)
            rows, sby = [], 0
            n_want = 1 << want_r
            while sby < sbh and len(rows) < 64:
                mx = min(sbh - sby, max_tile_height_sb)
                left_tiles = max(1, n_want - len(rows))
                th = 1 if mx <= 1 else min(mx, max(1, self.ri(1, max(1, 2 * (sbh - sby) // left_tiles))))
                if len(rows) + 1 >= n_want:
                    th = mx
                if mx > 1:
                    b.uniform(mx, th - 1)
                rows.append(sby)
                sby += th
            log2_rows = tile_log2(1, len(rows))
        fr.tile_cols, fr.tile_rows = len(cols), len(rows)
        fr.col_start_sb, fr.row_start_sb = cols + [sbw], rows + [sbh]
        fr.n_bytes = 4
        if log2_cols or log2_rows:
            fr.context_update_tile = self.ri(0, fr.tile_cols * fr.tile_rows - 1)
            b.f(log2_cols + log2_rows, fr.context_update_tile)
            b.f(2, fr.n_bytes - 1)

    def _quant(self, b, fr):
        seq, k = self.seq, self.k
        fr.lossless_wanted = self.p(k.lossless)
        fr.base_q_idx = 0 if fr.lossless_wanted else self.ri(1, 255)
        b.f(8, fr.base_q_idx)

        def delta(allow):
            v = self.ri(-40, 40) if allow and self.p(0.3) else 0
            b.bit(v != 0)
            if v:
                b.su(7, v)
            return v
        allow = not fr.lossless_wanted
        fr.ydc = delta(allow)
        fr.udc = fr.uac = fr.vdc = fr.vac = 0
        if not seq.monochrome:
            diff_uv = int(seq.separate_uv_delta_q and self.p(0.5))
            if seq.separate_uv_delta_q:
                b.bit(diff_uv)
            fr.udc = delta(allow)
            fr.uac = delta(allow)
            if diff_uv:
                fr.vdc = delta(allow)
                fr.vac = delta(allow)
            else:
                fr.vdc, fr.vac = fr.udc, fr.uac
        fr.qm = int(self.p(k.qm))
        b.bit(fr.qm)
        if fr.qm:
            b.f(4, self.ri(0, 15))
            b.f(4, self.ri(0, 15))
            if seq.separate_uv_delta_q:
                b.f(4, self.ri(0, 15))

    def _segmentation(self, b, fr, pri):
        k = self.k
        pin = k.seg_pin if fr.frame_type == INTER else 0
        fr.seg_enabled = int(self.p(k.segmentation) or pin > 0)
        b.bit(fr.seg_enabled)
        fr.seg = None
        if not fr.seg_enabled:
            return
        if fr.primary_ref_frame == PRIMARY_REF_NONE:
            update_map, update_data, temporal = 1, 1, 0
        else:
            update_map = int(self.p(0.7))
            b.bit(update_map)
            temporal = 0
            if update_map:
                temporal = int(self.p(0.5))
                b.bit(temporal)
            update_data = int(self.p(0.7) or pri.seg is None or pin > 0)
            b.bit(update_data)
        fr.seg_update_map, fr.seg_temporal, fr.seg_update_data = update_map, temporal, update_data
        if update_data:
            seg = []
            for i in range(8):
                d = dict(delta_q=0, lf=[0, 0, 0, 0], ref=-1, skip=0, globalmv=0)
                on = self.p(0.5)

                def feat(bits, lo, hi, prob=0.3):
                    if on and self.p(prob):
                        v = self.ri(lo, hi)
                        b.bit(1)
                        b.su(bits, v)
                        return v
                    b.bit(0)
                    return 0
                d["delta_q"] = feat(9, -255, 255)
                for j in range(4):
                    d["lf"][j] = feat(7, -63, 63)
                if i < pin or (on and self.p(0.15) and not pin):
                    d["ref"] = 1 + i % 7 if i < pin else self.ri(0, 7)
                    b.bit(1)
                    b.f(3, d["ref"])
                else:
                    b.bit(0)
                d["skip"] = int(on and self.p(0.1))
                b.bit(d["skip"])
                d["globalmv"] = int(on and self.p(0.1))
                b.bit(d["globalmv"])
                seg.append(d)
            fr.seg = seg
        else:
            fr.seg = pri.seg           # copied from the primary reference (src/obu.c:786-793)

    def _gmv(self, b, fr, pri):
        k = self.k
        out = []
        for i in range(7):
            if not self.p(k.gmv):
                b.bit(0)
                out.append(DEFAULT_GMV)
                continue
            t = [WM_TRANSLATION, WM_ROT_ZOOM, WM_AFFINE][self.ri(0, 2)]
            b.bit(1)
            b.bit(t == WM_ROT_ZOOM)
            if t != WM_ROT_ZOOM:
                b.bit(t == WM_TRANSLATION)
            ref = pri.gmv[i][1] if pri is not None else DEFAULT_GMV[1]
            mat = [0, 0, 1 << 16, 0, 0, 1 << 16]

            def near(refv, spread, n):
                lo, hi = -(1 << n), (1 << n)
                return int(min(hi, max(lo, refv + self.ri(-spread, spread))))
            if t >= WM_ROT_ZOOM:
                r2 = (ref[2] - (1 << 16)) >> 1
                v = near(r2 if self.p(0.5) else 0, 600, 12)
                b.subexp(v, r2, 12)
                mat[2] = (1 << 16) + 2 * v
                r3 = ref[3] >> 1
                v = near(r3 if self.p(0.5) else 0, 600, 12)
                b.subexp(v, r3, 12)
                mat[3] = 2 * v
                bits, shift = 12, 10
            else:
                bits, shift = 9 - (not fr.hp), 13 + (not fr.hp)
            if t == WM_AFFINE:
                r4 = ref[4] >> 1
                v = near(r4 if self.p(0.5) else 0, 600, 12)
                b.subexp(v, r4, 12)
                mat[4] = 2 * v
                r5 = (ref[5] - (1 << 16)) >> 1
                v = near(r5 if self.p(0.5) else 0, 600, 12)
                b.subexp(v, r5, 12)
                mat[5] = (1 << 16) + 2 * v
            else:
                mat[4], mat[5] = -mat[3], mat[2]
            for j in range(2):
                rj = ref[j] >> shift
                rj = max(-(1 << bits), min(1 << bits, rj))
                v = near(rj if self.p(0.5) else 0, 1 << (bits - 3), bits)
                b.subexp(v, rj, bits)
                mat[j] = v * (1 << shift)
            out.append((t, mat))
        fr.gmv = out

    def _film_grain(self, b, fr):
        seq, k = self.seq, self.k
        fr.grain = None
        if not (seq.film_grain and (fr.show_frame or fr.showable_frame)):
            return
        on = self.p(k.film_grain)
        b.bit(on)
        if not on:
            return
        seed = self.ri(0, 65535)
        b.f(16, seed)
        update = 1
        if fr.frame_type == INTER:
            cands = [i for i in sorted(set(fr.refidx)) if self.slots[i].grain is not None]
            update = int(not cands or self.p(0.6))
            b.bit(update)
            if not update:
                idx = cands[self.ri(0, len(cands) - 1)]
                b.f(3, idx)
                fr.grain = dict(self.slots[idx].grain, seed=seed)
                return
        g = dict(seed=seed)
        ny = self.ri(0, 14) if self.p(0.85) else 0
        b.f(4, ny)
        xs = sorted(self.rng.choice(256, size=ny, replace=False).tolist()) if ny else []
        for x in xs:
            b.f(8, x)
            b.f(8, self.ri(0, 255))
        cfl = 0
        if not seq.monochrome:
            cfl = int(self.p(0.3))
            b.bit(cfl)
        nuv = [0, 0]
        if seq.monochrome or cfl or (seq.ss_ver == 1 and seq.ss_hor == 1 and not ny):
            pass
        else:
            both = self.p(0.8)
            for pl in range(2):
                n = self.ri(1, 10) if both else 0
                if not (seq.ss_hor == 1 and seq.ss_ver == 1) and not both and self.p(0.5):
                    n = self.ri(0, 10)          # 4:4:4 / 4:2:2 may have grain on one chroma plane only
                nuv[pl] = n
                b.f(4, n)
                xs = sorted(self.rng.choice(256, size=n, replace=False).tolist()) if n else []
                for x in xs:
                    b.f(8, x)
                    b.f(8, self.ri(0, 255))
        b.f(2, self.ri(0, 3))                  # scaling_shift - 8
        lag = self.ri(0, 3)
        b.f(2, lag)
        n_pos = 2 * lag * (lag + 1)
        if ny:
            for _ in range(n_pos):
                b.f(8, self.ri(128 - 40, 128 + 40))
        for pl in range(2):
            if nuv[pl] or cfl:
                for _ in range(n_pos + (1 if ny else 0)):
                    b.f(8, self.ri(128 - 40, 128 + 40))
        b.f(2, self.ri(0, 3))                  # ar_coeff_shift - 6
        b.f(2, self.ri(0, 3))                  # grain_scale_shift
        for pl in range(2):
            if nuv[pl]:
                b.f(8, self.ri(0, 255))
                b.f(8, self.ri(0, 255))
                b.f(9, self.ri(0, 511))
        b.bit(self.p(0.7))                     # overlap_flag
        b.bit(self.p(0.3))                     # clip_to_restricted_range
        g.update(num_y=ny, num_uv=nuv, cfl=cfl)
        fr.grain = g

    def _skip_mode_allowed(self, fr):
        """src/obu.c:944-1005"""
        bits = self.seq.order_hint_bits
        if not (fr.switchable_comp_refs and fr.frame_type in (INTER, SWITCH)):
            return False
        poc = fr.frame_offset
        off_before = off_after = -1
        for i in range(7):
            refpoc = self.slots[fr.refidx[i]].frame_offset
            diff = poc_diff(bits, refpoc, poc)
            if diff > 0:
                if off_after < 0 or poc_diff(bits, off_after, refpoc) > 0:
                    off_after = refpoc
            elif diff < 0 and (off_before < 0 or poc_diff(bits, refpoc, off_before) > 0):
                off_before = refpoc
        if off_before >= 0 and off_after >= 0:
            return True
        if off_before >= 0:
            off_before2 = -1
            for i in range(7):
                refpoc = self.slots[fr.refidx[i]].frame_offset
                if poc_diff(bits, refpoc, off_before) < 0:
                    if off_before2 < 0 or poc_diff(bits, refpoc, off_before2) > 0:
                        off_before2 = refpoc
            return off_before2 >= 0
        return False

    # ---- one frame
    def next_frame(self, frame_type=None, tiles_log2=None, show=None, force=None):
        """Appends one temporal unit holding one frame (or, when a hidden frame is due, its show_existing_frame header).
        force: dict of Frame attributes decided by the caller instead of drawn (e.g. dict(seg_enabled=0))."""
        seq, k = self.seq, self.k
        valid = [i for i in range(8) if self.slots[i].valid]
        if self.hidden and (self.p(0.5) or len(self.hidden) > 2) and frame_type is None:
            return self._show_existing(self.hidden.pop(0))
        fr = Frame()
        fr.index = self.n
        if frame_type is None:
            frame_type = KEY if not valid else (INTRA_ONLY if self.p(k.intra_only) else INTER)
        fr.frame_type = frame_type
        is_intra = frame_type in (KEY, INTRA_ONLY)
        fr.show_frame = 1 if frame_type == KEY else int(not self.p(k.hidden)) if show is None else int(show)
        fr.showable_frame = int(frame_type != KEY) if fr.show_frame else 1
        fr.error_resilient = 1 if (frame_type == KEY and fr.show_frame) or frame_type == SWITCH else int(self.p(k.error_resilient))
        b = BitWriter()
        b.bit(0)                                   # show_existing_frame
        b.f(2, frame_type)
        b.bit(fr.show_frame)
        if not fr.show_frame:
            b.bit(fr.showable_frame)
        if not ((frame_type == KEY and fr.show_frame) or frame_type == SWITCH):
            b.bit(fr.error_resilient)
        fr.disable_cdf_update = int(self.p(k.disable_cdf_update))
        b.bit(fr.disable_cdf_update)
        if seq.screen_content_tools == 2:
            fr.allow_sct = int(self.p(k.screen_content))
            b.bit(fr.allow_sct)
        else:
            fr.allow_sct = seq.screen_content_tools
        fr.force_integer_mv = 0
        if fr.allow_sct:
            fr.force_integer_mv = int(self.p(k.force_integer_mv))
            b.bit(fr.force_integer_mv)             # (sequence: adaptive)
        if is_intra:
            fr.force_integer_mv = 1
        fr.allow_intrabc_wanted = bool(is_intra and fr.allow_sct and self.p(k.intrabc))
        # the references of an inter frame (needed before the size: use_ref / scaling limits)
        fr.refidx = [0] * 7
        if not is_intra:
            fr.refidx = [valid[self.ri(0, len(valid) - 1)] for _ in range(7)]
        self._pick_size(fr, None if is_intra else [(self.slots[i].w1, self.slots[i].h) for i in fr.refidx])
        if frame_type == SWITCH:
            fr.frame_size_override = 1
        else:
            b.bit(fr.frame_size_override)
        self.order += 1
        fr.frame_offset = (self.order + (self.ri(1, 3) if not fr.show_frame else 0)) & ((1 << seq.order_hint_bits) - 1)
        b.f(seq.order_hint_bits, fr.frame_offset)
        fr.primary_ref_frame = PRIMARY_REF_NONE
        if not fr.error_resilient and not is_intra:
            if self.p(k.primary_ref):
                fr.primary_ref_frame = self.ri(0, 6)
            b.f(3, fr.primary_ref_frame)
        pri = self.slots[fr.refidx[fr.primary_ref_frame]] if fr.primary_ref_frame != PRIMARY_REF_NONE else None
        # refresh flags
        if frame_type == KEY and fr.show_frame:
            fr.refresh = 0xff
        elif frame_type == SWITCH:
            fr.refresh = 0xff
        else:
            fr.refresh = self.ri(0, 254) if frame_type == INTRA_ONLY else self.ri(0, 255)
            if not fr.show_frame:
                fr.refresh |= 1 << self.ri(0, 7)       # a hidden frame is kept somewhere, or it could never be shown
            if len(valid) < 8 and self.p(0.7):
                free = [i for i in range(8) if i not in valid]
                fr.refresh |= 1 << free[self.ri(0, len(free) - 1)]
            if frame_type == INTRA_ONLY and fr.refresh == 0xff:
                fr.refresh = 0x7f
            b.f(8, fr.refresh)
        if is_intra:
            if fr.refresh != 0xff and fr.error_resilient:
                for i in range(8):
                    b.f(seq.order_hint_bits, self.slots[i].frame_offset)
            self._frame_size(b, fr)
            fr.allow_intrabc = 0
            if fr.allow_sct and not fr.super_res:
                fr.allow_intrabc = int(fr.allow_intrabc_wanted)
                b.bit(fr.allow_intrabc)
        else:
            fr.allow_intrabc = 0
            if fr.error_resilient:
                for i in range(8):
                    b.f(seq.order_hint_bits, self.slots[i].frame_offset)
            b.bit(0)                               # frame_refs_short_signaling
            for i in range(7):
                b.f(3, fr.refidx[i])
            if fr.frame_size_override and not fr.error_resilient:
                # found_ref: take the size of a reference when one happens to have the size we want, else spell it out
                match = [i for i in range(7) if (self.slots[fr.refidx[i]].w1, self.slots[fr.refidx[i]].h) == (fr.w1, fr.h)]
                if match and self.p(0.7):
                    for i in range(match[0]):
                        b.bit(0)
                    b.bit(1)
                    self._superres(b, fr)
                else:
                    for i in range(7):
                        b.bit(0)
                    self._frame_size(b, fr)
            else:
                self._frame_size(b, fr)
            fr.hp = 0
            if not fr.force_integer_mv:
                fr.hp = int(not self.p(k.no_hp))
                b.bit(fr.hp)
            fixed = self.p(k.fixed_filter)
            b.bit(not fixed)
            if fixed:
                b.f(2, self.ri(0, 3))
            b.bit(not self.p(k.no_switchable_motion))
            fr.use_ref_frame_mvs = 0
            if not fr.error_resilient and seq.ref_frame_mvs:
                fr.use_ref_frame_mvs = int(self.p(k.use_ref_frame_mvs))
                b.bit(fr.use_ref_frame_mvs)
        if is_intra:
            fr.hp = 0
        fr.refresh_context = 0
        if not fr.disable_cdf_update:
            fr.refresh_context = int(self.p(0.8))
            b.bit(not fr.refresh_context)
        if tiles_log2 is not None:
            fr.want_tiles_log2 = tiles_log2
        self._tiles(b, fr)
        self._quant(b, fr)
        self._segmentation(b, fr, pri)
        # delta q / lf
        fr.delta_q = fr.delta_lf = fr.delta_lf_multi = 0
        if fr.base_q_idx:
            fr.delta_q = int(self.p(k.delta_q))
            b.bit(fr.delta_q)
            if fr.delta_q:
                b.f(2, self.ri(0, 3))
                if not fr.allow_intrabc:
                    fr.delta_lf = int(self.p(k.delta_lf))
                    b.bit(fr.delta_lf)
                    if fr.delta_lf:
                        b.f(2, self.ri(0, 3))
                        fr.delta_lf_multi = int(self.p(0.5))
                        b.bit(fr.delta_lf_multi)
        # lossless derivation (src/obu.c:824-836)
        delta_lossless = not (fr.ydc or fr.udc or fr.uac or fr.vdc or fr.vac)
        qidx = [min(255, max(0, fr.base_q_idx + (fr.seg[i]["delta_q"] if fr.seg_enabled and fr.seg else 0))) for i in range(8)]
        fr.lossless = [int(not qidx[i] and delta_lossless) for i in range(8)]
        fr.all_lossless = int(all(fr.lossless))
        # loop filter
        fr.ref_delta = list(pri.ref_delta) if pri is not None else [1, 0, 0, 0, -1, 0, -1, -1]
        fr.mode_delta = list(pri.mode_delta) if pri is not None else [0, 0]
        fr.lf_level = [0, 0, 0, 0]
        if fr.all_lossless or fr.allow_intrabc:
            fr.ref_delta, fr.mode_delta = [1, 0, 0, 0, -1, 0, -1, -1], [0, 0]
        else:
            off = self.p(k.lf_off)
            l0, l1 = (0, 0) if off else (self.ri(0, 63), self.ri(0, 63))
            b.f(6, l0)
            b.f(6, l1)
            fr.lf_level[:2] = [l0, l1]
            if not seq.monochrome and (l0 or l1):
                fr.lf_level[2], fr.lf_level[3] = self.ri(0, 63), self.ri(0, 63)
                b.f(6, fr.lf_level[2])
                b.f(6, fr.lf_level[3])
            b.f(3, self.ri(0, 7))                  # sharpness
            en = int(self.p(0.7))
            b.bit(en)
            if en:
                upd = int(self.p(k.lf_delta_update))
                b.bit(upd)
                if upd:
                    for i in range(8):
                        if self.p(0.4):
                            fr.ref_delta[i] = self.ri(-20, 20)
                            b.bit(1)
                            b.su(7, fr.ref_delta[i])
                        else:
                            b.bit(0)
                    for i in range(2):
                        if self.p(0.4):
                            fr.mode_delta[i] = self.ri(-20, 20)
                            b.bit(1)
                            b.su(7, fr.mode_delta[i])
                        else:
                            b.bit(0)
        # cdef
        fr.cdef_bits = 0
        fr.cdef_on = 0
        if not fr.all_lossless and seq.cdef and not fr.allow_intrabc:
            fr.cdef_on = 1
            b.f(2, self.ri(0, 3))                  # damping - 3
            strong = self.p(k.cdef)
            fr.cdef_bits = self.ri(0, 3) if strong else 0
            b.f(2, fr.cdef_bits)
            for _ in range(1 << fr.cdef_bits):
                b.f(6, self.ri(0, 63) if strong else 0)
                if not seq.monochrome:
                    b.f(6, self.ri(0, 63) if strong else 0)
        # loop restoration
        fr.lr_type = [0, 0, 0]
        if (not fr.all_lossless or fr.super_res) and seq.restoration and not fr.allow_intrabc:
            on = self.p(k.restoration)
            n_pl = 1 if seq.monochrome else 3
            for pl in range(n_pl):
                fr.lr_type[pl] = self.ri(0, 3) if on and self.p(0.8) else 0
                b.f(2, fr.lr_type[pl])
            if any(fr.lr_type):
                if self.p(0.5):
                    b.bit(1)
                    if not seq.sb128:
                        b.bit(self.p(0.5))
                else:
                    b.bit(0)
                if (fr.lr_type[1] or fr.lr_type[2]) and seq.ss_hor == 1 and seq.ss_ver == 1:
                    b.bit(self.p(0.5))
        if not fr.all_lossless:
            fr.tx_switchable = int(not self.p(k.tx_largest))
            b.bit(fr.tx_switchable)
        fr.switchable_comp_refs = 0
        if not is_intra:
            fr.switchable_comp_refs = int(not self.p(k.single_ref))
            b.bit(fr.switchable_comp_refs)
        fr.skip_mode_allowed = int(self._skip_mode_allowed(fr))
        if fr.skip_mode_allowed:
            fr.skip_mode_enabled = int(self.p(0.7))
            b.bit(fr.skip_mode_enabled)
        if not fr.error_resilient and not is_intra and seq.warped_motion:
            b.bit(not self.p(k.no_warp))
        b.bit(self.p(k.reduced_txtp_set))
        fr.gmv = [DEFAULT_GMV] * 7
        if not is_intra:
            self._gmv(b, fr, pri)
        self._film_grain(b, fr)
        b.align()
        n_tiles = fr.tile_cols * fr.tile_rows
        if n_tiles > 1:
            b.bit(0)                               # tile_start_and_end_present_flag
            b.align()
        hdr = b.tobytes()
        # tile payloads
        sb = 64 << seq.sb128
        body = bytearray()
        tiles = []
        for tr in range(fr.tile_rows):
            for tc in range(fr.tile_cols):
                tw = min(fr.col_start_sb[tc + 1] * sb, fr.w0) - fr.col_start_sb[tc] * sb
                th = min(fr.row_start_sb[tr + 1] * sb, fr.h) - fr.row_start_sb[tr] * sb
                size = int(tw * th * k.bytes_per_pixel * (1 if seq.monochrome else 1.5 if seq.layout == LAYOUT_I420 else 2 if seq.layout == LAYOUT_I422 else 3) * (2 if seq.bpc > 8 else 1)) + 256
                last = tr == fr.tile_rows - 1 and tc == fr.tile_cols - 1
                if not last:
                    body += (size - 1).to_bytes(fr.n_bytes, "little")
                tiles.append((len(body), size))
                body += self.rng.bytes(size)
        payload = hdr + bytes(body)
        head = b""
        if not self.units:
            head = obu(OBU_SEQ_HDR, seq.write())
        pre = obu(OBU_TD, b"") + head
        frame_obu_hdr = bytes([OBU_FRAME << 3 | 2]) + leb128(len(payload))
        base = len(pre) + len(frame_obu_hdr) + len(hdr)
        data = bytearray(pre + frame_obu_hdr + payload)
        self.units.append(dict(data=data, tiles=[(base + o, s) for o, s in tiles], frame=fr, pre_len=len(pre), obu_hdr_len=len(frame_obu_hdr),
                               size_bytes=fr.n_bytes))
        # the slots the frame goes to
        for i in range(8):
            if fr.refresh >> i & 1:
                s = self.slots[i]
                s.valid = True
                s.frame_offset = fr.frame_offset
                s.w1, s.h = fr.w1, fr.h
                s.frame_type = fr.frame_type
                s.gmv = list(fr.gmv)
                s.ref_delta, s.mode_delta = list(fr.ref_delta), list(fr.mode_delta)
                s.seg = fr.seg if fr.seg_enabled else None
                s.grain = fr.grain
                s.showable = bool(fr.showable_frame)
        if frame_type == KEY and fr.show_frame:
            self.hidden = []
        if not fr.show_frame:
            keep = [i for i in range(8) if fr.refresh >> i & 1]
            self.hidden.append((keep[0], fr.index))
        # a hidden frame whose slot is overwritten can no longer be shown
        self.hidden = [(s, n) for s, n in self.hidden if n == fr.index or not (fr.refresh >> s & 1)]
        self.n += 1
        return fr

    def _show_existing(self, slot_frame):
        slot, index = slot_frame
        b = BitWriter()
        b.bit(1)
        b.f(3, slot)
        b.trailing()
        data = bytearray(obu(OBU_TD, b"") + obu(OBU_FRAME_HDR, b.tobytes()))
        fr = Frame()
        fr.show_existing, fr.index = index, index
        self.units.append(dict(data=data, tiles=[], frame=None, shows=index))
        return None

    def flush_hidden(self):
        while self.hidden:
            self._show_existing(self.hidden.pop(0))

    def reroll(self, unit, tile, from_byte):
        """fresh random bytes for a tile's payload from `from_byte` (offset inside the tile) on"""
        u = self.units[unit]
        off, size = u["tiles"][tile]
        from_byte = max(0, min(from_byte, size - 1))
        u["data"][off + from_byte:off + size] = self.rng.bytes(size - from_byte)


    def truncate(self, unit, tile, new_size):
        """Damage on purpose (the error-path tests): the tile's payload is cut to `new_size` bytes and the framing follows — its
        tile_size_minus_1 field (every tile but the frame's last carries one) and the frame OBU's size — so that dav1d's parser accepts the
        unit and the symbol decoder of that tile runs out of data (msac.cnt <= -15, reference src/decode.c:2743: the tile task fails)."""
        u = self.units[unit]
        off, size = u["tiles"][tile]
        new_size = max(1, min(new_size, size))
        data, nb = u["data"], u["size_bytes"]
        last = tile == len(u["tiles"]) - 1
        if not last:
            data[off - nb:off] = (new_size - 1).to_bytes(nb, "little")
        cut = size - new_size
        body_start = u["pre_len"] + u["obu_hdr_len"]
        payload = bytes(data[body_start:off + new_size]) + bytes(data[off + size:])
        hdr = bytes([OBU_FRAME << 3 | 2]) + leb128(len(payload))
        shift = len(hdr) - u["obu_hdr_len"]
        u["data"] = bytearray(bytes(data[:u["pre_len"]]) + hdr + payload)
        u["obu_hdr_len"] = len(hdr)
        u["tiles"] = [(o + shift - (cut if i > tile else 0), new_size if i == tile else s) for i, (o, s) in enumerate(u["tiles"])]


def make_stream(w, h, layout, bpc, n_frames, seed, sb128=True, knobs=None, **seq_kw):
    seq = Seq(w, h, layout, bpc, sb128, **seq_kw)
    sw = StreamWriter(seq, seed, knobs)
    while sw.n < n_frames:
        sw.next_frame()
    sw.flush_hidden()
    return sw
