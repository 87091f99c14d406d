"""Deterministic oracle-only cases whose outputs are pinned as SHA-256 digests in
tests/golden/dsp_golden.json.  The digests were produced by tests/golden/make_golden.py from the
reference's own C functions (oracle/_ref); test_oracle.py replays the same cases through
oracle/port (and through oracle/_ref again when it is present) and compares digests, so the C
restatement stays pinned even where /root/reference and oracle/_ref are absent."""
import hashlib

import numpy as np

import util
import test_mc


def _digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def itx_case(oracle, bpc, tx):
    rng = np.random.default_rng(4242 + 31 * tx + bpc)
    w, h = util.TX_W[tx], util.TX_H[tx]
    n = min(w, 32) * min(h, 32)
    pd = util.pix_dtype(bpc)
    outs = []
    for txtp in util.legal_txtps(tx):
        for subsh in range(1 if txtp else 0, util.subsh_max(tx)):
            if txtp == util.WHT_WHT and subsh > 1:
                continue
            cf, eob = util.gen_itx_coefs(rng, tx, txtp, bpc, subsh)
            dst = rng.integers(0, 1 << bpc, size=(h, w + 3)).astype(pd)
            oracle.call(bpc, "itxfm_add", tx, txtp, dst.ctypes.data, dst.strides[0], cf, eob)
            outs += [dst, cf[:n]]
    return _digest(*outs)


def mc_case(oracle, bpc, kind):
    rng = np.random.default_rng(900 + bpc * 2 + kind)
    vis_w, vis_h = 200, 150
    pd = util.pix_dtype(bpc)
    refplane = rng.integers(0, 1 << bpc, size=(256, 256)).astype(pd)
    tasks, pos, prep_sz = test_mc._gen_tasks(rng, 120, vis_w, vis_h, 512, 512, kind, test_mc.SIZES_PUT)
    plane = rng.integers(0, 1 << bpc, size=(512, 512)).astype(pd)
    prep = np.zeros(max(prep_sz, 1), np.int16)
    for i, t in enumerate(tasks):
        if kind == 0:
            x, y = pos[i]
            test_mc._oracle_mc(oracle, bpc, refplane, vis_w, vis_h, t, dst_block=plane[y:, x:])
        else:
            test_mc._oracle_mc(oracle, bpc, refplane, vis_w, vis_h, t, tmp=prep[pos[i]:])
    return _digest(plane, prep)


def comp_case(oracle, bpc):
    rng = np.random.default_rng(77 + bpc)
    pd = util.pix_dtype(bpc)
    bias = 8192 if bpc > 8 else 0
    ib = 4 if bpc == 8 else 14 - bpc
    outs = []
    for kind, ss in [(0, 0), (1, 0), (2, 0), (3, 0), (3, 1), (3, 2)] * 4:
        w = int(rng.choice([4, 8, 16, 32, 64, 128]))
        h = int(rng.choice([v for v in [4, 8, 16, 32, 64, 128] if w // 4 <= v <= w * 4]))
        t1 = ((rng.integers(0, 1 << bpc, size=w * h) << ib) - bias).astype(np.int16)
        t2 = ((rng.integers(0, 1 << bpc, size=w * h) << ib) - bias).astype(np.int16)
        dst = rng.integers(0, 1 << bpc, size=(h, w)).astype(pd)
        m = rng.integers(0, 65, size=w * h).astype(np.uint8)
        mo = np.zeros(w * h, np.uint8)
        if kind == 0:
            oracle.call(bpc, "avg", 0, 0, dst, dst.strides[0], t1, t2, w, h)
        elif kind == 1:
            oracle.call(bpc, "w_avg", 0, 0, dst, dst.strides[0], t1, t2, w, h, int(rng.integers(1, 16)))
        elif kind == 2:
            oracle.call(bpc, "mask", 0, 0, dst, dst.strides[0], t1, t2, w, h, m)
        else:
            oracle.call(bpc, "w_mask", ss, 0, dst, dst.strides[0], t1, t2, w, h, mo, int(rng.integers(0, 2)))
            n = (w >> (1 if ss else 0)) * (h >> (1 if ss == 2 else 0))
            outs.append(mo[:n])
        outs.append(dst)
    return _digest(*outs)


def blend_case(oracle, bpc):
    rng = np.random.default_rng(1234 + bpc)
    pd = util.pix_dtype(bpc)
    outs = []
    for fam in ("blend", "blend_v", "blend_h"):
        for w in (4, 8, 16, 32):       # blend_v: w up to 32, blend_h: h up to 32 (obmc mask table)
            for h in (4, 8, 16, 32):
                dst = rng.integers(0, 1 << bpc, size=(h, w + 5)).astype(pd)
                tmp = rng.integers(0, 1 << bpc, size=(h, w)).astype(pd)
                if fam == "blend":
                    m = rng.integers(0, 65, size=w * h).astype(np.uint8)
                    oracle.call(bpc, fam, 0, 0, dst, dst.strides[0], tmp, w, h, m)
                else:
                    oracle.call(bpc, fam, 0, 0, dst, dst.strides[0], tmp, w, h)
                outs.append(dst)
    return _digest(*outs)


def all_cases():
    """name -> callable(oracle)"""
    cases = {}
    for bpc in (8, 10, 12):
        for tx in range(19):
            cases["itx/%dbpc/%s" % (bpc, util.TX_NAMES[tx])] = (lambda o, b=bpc, t=tx: itx_case(o, b, t))
        for kind, nm in ((0, "put"), (1, "prep")):
            cases["mc/%dbpc/%s" % (bpc, nm)] = (lambda o, b=bpc, k=kind: mc_case(o, b, k))
        cases["comp/%dbpc" % bpc] = (lambda o, b=bpc: comp_case(o, b))
        cases["blend/%dbpc" % bpc] = (lambda o, b=bpc: blend_case(o, b))
    return cases
