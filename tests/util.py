"""Shared helpers of the parity tests: oracle access (ctypes), checkasm-style input
generators, and the backend selector (HIP library on a GPU, SIMT-emulated build on CPU)."""
import ctypes as C
import functools
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# The reference compiled here.  The PARITY checker is oracle/_ref (-O2 -g, asserts live).  $DAV1D_REF_BUILD=release (bench.py sets it for its
# CPU-peer timings) selects oracle/_ref_release: the same sources with the flags dav1d's own release build uses (oracle/Makefile REL=1);
# tests/test_oracle.py asserts that the two builds produce identical pictures.
REF_RELEASE = os.environ.get("DAV1D_REF_BUILD") == "release"
REF_SO = os.path.join(ROOT, "oracle", "_ref_release" if REF_RELEASE else "_ref", "libdav1d_ref.so")
REF_FLAGS = "-O3 -DNDEBUG -fomit-frame-pointer -ffast-math (dav1d's release build, meson.build:26-29, 309-312)" if REF_RELEASE else "-O2 -g, asserts live (the parity checker's build)"

PORT_SO = os.path.join(ROOT, "oracle", "libdav1d_port.so")
EMU_SO = os.path.join(ROOT, "tests", "emu", "libdav1d_hip_emu.so")
GOLDEN = os.path.join(ROOT, "tests", "golden")

TX_W = [4, 8, 16, 32, 64, 4, 8, 8, 16, 16, 32, 32, 64, 4, 16, 8, 32, 16, 64]
TX_H = [4, 8, 16, 32, 64, 8, 4, 16, 8, 32, 16, 64, 32, 16, 4, 32, 8, 64, 16]
TX_NAMES = ["4x4", "8x8", "16x16", "32x32", "64x64", "4x8", "8x4", "8x16", "16x8", "16x32", "32x16",
            "32x64", "64x32", "4x16", "16x4", "8x32", "32x8", "16x64", "64x16"]
# itxfm_add table index names (reference src/levels.h:80-100)
TXTP_NAMES = ["DCT_DCT", "ADST_DCT", "DCT_ADST", "ADST_ADST", "FLIPADST_DCT", "DCT_FLIPADST", "FLIPADST_FLIPADST",
              "ADST_FLIPADST", "FLIPADST_ADST", "IDTX", "V_DCT", "H_DCT", "V_ADST", "H_ADST", "V_FLIPADST",
              "H_FLIPADST", "WHT_WHT"]
WHT_WHT = 16


def legal_txtps(tx):
    """legal types per size, reference src/itx_tmpl.c:160-178."""
    w, h = TX_W[tx], TX_H[tx]
    m = max(w, h)
    if m == 64:
        return [0]
    if m == 32:
        return [0, 9]
    if w == 16 and h == 16:
        return list(range(12))
    return list(range(16)) + ([WHT_WHT] if tx == 0 else [])


# ------------------------------------------------------------------ oracle

@functools.lru_cache(None)
def ref_lib():
    """The reference's own C path (oracle/_ref, built from /root/reference by oracle/Makefile)."""
    if not os.path.exists(REF_SO):
        if os.path.isdir("/root/reference/src"):
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref", "-j8"] + (["REL=1"] if REF_RELEASE else []), check=True,
                           stdout=subprocess.DEVNULL)
        else:
            return None
    lib = C.CDLL(REF_SO)
    lib.dav1d_ref_dsp_entry.restype = C.c_void_p
    lib.dav1d_ref_dsp_entry.argtypes = [C.c_int, C.c_char_p, C.c_int, C.c_int]
    lib.dav1d_ref_table.restype = C.c_void_p
    lib.dav1d_ref_table.argtypes = [C.c_char_p, C.POINTER(C.c_size_t)]
    lib.dav1d_ref_scan.restype = C.POINTER(C.c_uint16)
    lib.dav1d_ref_scan.argtypes = [C.c_int]
    lib.dav1d_ref_tx1d_fn.restype = C.c_void_p
    lib.dav1d_ref_tx1d_fn.argtypes = [C.c_int, C.c_int]
    lib.dav1d_ref_wht4_1d.restype = C.c_void_p
    return lib


@functools.lru_cache(None)
def ref_release_lib():
    """oracle/_ref_release: the reference with dav1d's release flags (oracle/Makefile REL=1), or None."""
    so = os.path.join(ROOT, "oracle", "_ref_release", "libdav1d_ref.so")
    if not os.path.exists(so):
        if not os.path.isdir("/root/reference/src"):
            return None
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "ref", "-j8", "REL=1"], check=True, stdout=subprocess.DEVNULL)
    lib = C.CDLL(so)
    lib.dav1d_ref_dsp_entry.restype = C.c_void_p
    lib.dav1d_ref_dsp_entry.argtypes = [C.c_int, C.c_char_p, C.c_int, C.c_int]
    return lib


@functools.lru_cache(None)
def port_lib():
    """This repo's C restatement (oracle/port/*.c)."""
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "port"], check=True, stdout=subprocess.DEVNULL)
    lib = C.CDLL(PORT_SO)
    lib.dav1d_port_dsp_entry.restype = C.c_void_p
    lib.dav1d_port_dsp_entry.argtypes = [C.c_int, C.c_char_p, C.c_int, C.c_int]
    return lib


_vp, _i, _pd = C.c_void_p, C.c_int, C.c_ssize_t
# C prototypes of the reference DSP entries; the 16 bpc flavours carry a trailing bitdepth_max
PROTO = {
    "itxfm_add": [_vp, _pd, _vp, _i],
    "mc": [_vp, _pd, _vp, _pd, _i, _i, _i, _i],
    "mct": [_vp, _vp, _pd, _i, _i, _i, _i],
    "avg": [_vp, _pd, _vp, _vp, _i, _i],
    "w_avg": [_vp, _pd, _vp, _vp, _i, _i, _i],
    "mask": [_vp, _pd, _vp, _vp, _i, _i, _vp],
    "w_mask": [_vp, _pd, _vp, _vp, _i, _i, _vp, _i],
    "emu_edge": [_pd, _pd, _pd, _pd, _pd, _pd, _vp, _pd, _vp, _pd],
    "mc_scaled": [_vp, _pd, _vp, _pd, _i, _i, _i, _i, _i, _i],
    "mct_scaled": [_vp, _vp, _pd, _i, _i, _i, _i, _i, _i],
    "warp8x8": [_vp, _pd, _vp, _pd, _vp, _i, _i],
    "warp8x8t": [_vp, _pd, _vp, _pd, _vp, _i, _i],
    "resize": [_vp, _pd, _vp, _pd, _i, _i, _i, _i, _i],
    "blend": [_vp, _pd, _vp, _i, _i, _vp],
    "blend_v": [_vp, _pd, _vp, _i, _i],
    "blend_h": [_vp, _pd, _vp, _i, _i],
    "loop_filter_sb": [_vp, _pd, _vp, _vp, _pd, _vp, _i],
    "intra_pred": [_vp, _pd, _vp, _i, _i, _i, _i, _i],
    "cfl_ac": [_vp, _vp, _pd, _i, _i, _i, _i],
    "cfl_pred": [_vp, _pd, _vp, _i, _i, _vp, _i],
    "pal_pred": [_vp, _pd, _vp, _vp, _i, _i],
    "wiener": [_vp, _pd, _vp, _vp, _i, _i, _vp, _i],
    "sgr": [_vp, _pd, _vp, _vp, _i, _i, _vp, _i],
    "cdef_dir": [_vp, _pd, _vp],
    "cdef_fb": [_vp, _pd, _vp, _vp, _vp, _i, _i, _i, _i, _i],
    "generate_grain_y": [_vp, _vp],
    "generate_grain_uv": [_vp, _vp, _vp, _pd],
    "fgy_32x32xn": [_vp, _vp, _pd, _vp, C.c_size_t, _vp, _vp, _i, _i],
    "fguv_32x32xn": [_vp, _vp, _pd, _vp, C.c_size_t, _vp, _vp, _i, _i, _vp, _pd, _i, _i],
}
RET_INT = {"cdef_dir"}
NO_HBD_SUFFIX = {"blend", "blend_v", "blend_h", "emu_edge", "cfl_ac", "pal_pred"}


class Oracle:
    """Calls DSP entries of an oracle library (`ref` = the reference's C path, `port` = oracle/port)."""

    def __init__(self, which="ref"):
        # "ref_release": the release-flag build of the reference whatever $DAV1D_REF_BUILD says (test_oracle.py compares the builds)
        self.which = "ref" if which == "ref_release" else which
        self.lib = ref_release_lib() if which == "ref_release" else ref_lib() if which == "ref" else port_lib()
        if self.lib is None:
            raise RuntimeError("oracle library unavailable: " + which)
        self._entry = self.lib.dav1d_ref_dsp_entry if self.which == "ref" else self.lib.dav1d_port_dsp_entry
        self._cache = {}

    def fn(self, bpc, family, i=0, j=0):
        key = (bpc, family, i, j)
        if key not in self._cache:
            p = self._entry(bpc, family.encode(), i, j)
            if not p:
                raise KeyError("no %s[%d][%d] in %s oracle" % (family, i, j, self.which))
            args = list(PROTO[family])
            if bpc > 8 and family not in NO_HBD_SUFFIX:
                args.append(_i)
            self._cache[key] = C.CFUNCTYPE(C.c_int if family in RET_INT else None, *args)(p)
        return self._cache[key]

    def call(self, bpc, family, i, j, *args):
        f = self.fn(bpc, family, i, j)
        a = [x.ctypes.data if isinstance(x, np.ndarray) else x for x in args]
        if bpc > 8 and family not in NO_HBD_SUFFIX:
            a.append((1 << bpc) - 1)
        return f(*a)


def ref_table(name, dtype):
    """A constant table of the reference build (oracle/ref_shim.c) as a numpy array."""
    sz = C.c_size_t(0)
    p = ref_lib().dav1d_ref_table(name.encode(), C.byref(sz))
    assert p, name
    return np.frombuffer((C.c_uint8 * sz.value).from_address(p), dtype=dtype).copy()


def available_oracles():
    out = []
    if ref_lib() is not None:
        out.append("ref")
    if os.path.isdir(os.path.join(ROOT, "oracle", "port")) and os.listdir(os.path.join(ROOT, "oracle", "port")):
        out.append("port")
    return out


@functools.lru_cache(None)
def default_oracle():
    """The checker used by the HIP parity tests: the real reference when its prebuilt
    library travelled with the repo, else the C restatement (DAV1D_TEST_ORACLE=port forces the latter)."""
    if os.environ.get("DAV1D_TEST_ORACLE") == "port":
        return Oracle("port")
    return Oracle("ref") if ref_lib() is not None else Oracle("port")


# --------------------------------------------------------------- backends

def emu_lib_path():
    from dav1d_amd import build
    return build.build_emu()


def make_context(backend):
    from dav1d_amd import api
    if backend == "hip":
        return api.Context(0)
    return api.Context(0, lib_path=emu_lib_path())


def pix_dtype(bpc):
    return np.uint8 if bpc == 8 else np.uint16


def coef_dtype(bpc):
    return np.int16 if bpc == 8 else np.int32


# --------------------------------------------------- scan tables (inputs only)

@functools.lru_cache(None)
def scans():
    path = os.path.join(GOLDEN, "scans.npz")
    if os.path.exists(path):
        z = np.load(path)
        return [z["tx%d" % t] for t in range(19)]
    lib = ref_lib()
    out = []
    for t in range(19):
        n = min(TX_W[t], 32) * min(TX_H[t], 32)
        out.append(np.ctypeslib.as_array(lib.dav1d_ref_scan(t), shape=(n,)).copy())
    return out


# --------------------------------------------- itx inputs (tests/checkasm/itx.c:74-242)

_SCALE = [4.0, 4.0 * np.sqrt(0.5), 2.0, 2.0 * np.sqrt(0.5), 1.0, 0.5 * np.sqrt(0.5), 0.25, 0.125 * np.sqrt(0.5), 0.0625]
# 1-D kinds of the forward generator per itxfm_add index: 0 dct 1 adst(/flipadst) 2 identity 3 wht
_FWD = [(0, 0), (0, 1), (1, 0), (1, 1), (0, 1), (1, 0), (1, 1), (1, 1), (1, 1), (2, 2), (2, 0), (0, 2), (2, 1), (1, 2),
        (2, 1), (1, 2), (3, 3)]


def _fwd_mat(kind, n):
    i = np.arange(n)[:, None]
    j = np.arange(n)[None, :]
    if kind == 0:
        m = np.cos(np.pi * (2 * j + 1) * i / (2.0 * n))
        m[0] *= np.sqrt(0.5)
        return m
    if kind == 1:
        if n == 4:
            return np.sin(np.pi * (j + 1) * (2 * i + 1) / 9.0)
        return np.sin(np.pi * (2 * j + 1) * (2 * i + 1) / (4.0 * n))
    if kind == 3:
        return np.array([[.5, .5, .5, .5], [.5, .5, -.5, -.5], [.5, -.5, -.5, .5], [.5, -.5, .5, -.5]])
    return np.eye(n)


def tx_class(txtp):
    """0 = 2D, 1 = H, 2 = V (reference dav1d_tx_type_class, src/tables.c)."""
    if txtp in (11, 13, 15):
        return 1
    if txtp in (10, 12, 14):
        return 2
    return 0


def gen_itx_coefs(rng, tx, txtp, bpc, subsh):
    """Coefficients in the valid dynamic range + a consistent eob, in the reference slab
    layout (column-major, min(w,32) x min(h,32)).  Mirrors ftx()/copy_subcoefs()."""
    w, h = TX_W[tx], TX_H[tx]
    sw, sh = min(w, 32), min(h, 32)
    bdmax = (1 << bpc) - 1
    res = rng.integers(-bdmax, bdmax + 1, size=(h, w)).astype(np.float64)
    k1, k2 = _FWD[txtp]
    scale = _SCALE[int(np.log2(w * h)) - 4]
    t = res @ _fwd_mat(k1, w).T * scale          # rows
    out = _fwd_mat(k2, h) @ t                    # columns; out[y, x]
    # slab[y + x*sh] (reference src/itx_tmpl.c:98-105)
    buf = np.zeros(sw * sh, np.int64)
    for x in range(sw):
        buf[x * sh:(x + 1) * sh] = np.floor(out[:sh, x] + 0.5)
    # ---- copy_subcoefs
    scan = scans()[tx]
    cls = tx_class(txtp)
    sub_high = subsh * 8 - 1 if subsh > 0 else 0
    sub_low = sub_high - 8 if subsh > 1 else 0
    eob = 0
    n = 0
    while n < sw * sh:
        if cls == 0:
            rc = int(scan[n]); rcx, rcy = rc % sh, rc // sh
        elif cls == 1:
            rcx, rcy = n % sh, n // sh
        else:
            rcx, rcy = n // sw, n % sw
        if rcx > sub_high or rcy > sub_high:
            break
        if not eob and (rcx > sub_low or rcy > sub_low):
            eob = n
        n += 1
    if eob:
        eob += int(rng.integers(0, 1 << 30)) % (n - eob - 1) if n - eob - 1 > 0 else 0
    if cls == 0:
        buf[scan[eob + 1:].astype(np.int64)] = 0
    elif cls == 1:
        buf[eob + 1:] = 0
    else:
        rcx, rcy = eob // sw, eob % sw
        while rcx < sh:
            rcy += 1
            while rcy < sw:
                buf[rcy * sh + rcx] = 0
                rcy += 1
            rcx += 1
            rcy = -1
    return buf.astype(coef_dtype(bpc)), eob


def pack_coefs(tx, txtp, dense, eob):
    """Dense slab -> the eob + 1 values in decode order (the DAV1D_HIP_ITX_PACKED wire format): scan position i sits at
    dav1d_scans[tx][i] for the 2-D classes, at i for the H classes, at (i & (sw-1)) * sh + (i >> log2(sw)) for the V classes
    (reference src/recon_tmpl.c:458-496, 548-575)."""
    w, h = TX_W[tx], TX_H[tx]
    sw, sh = min(w, 32), min(h, 32)
    i = np.arange(eob + 1)
    cls = tx_class(txtp)
    if cls == 0:
        rc = scans()[tx][i].astype(np.int64)
    elif cls == 1:
        rc = i
    else:
        rc = (i & (sw - 1)) * sh + (i >> int(np.log2(sw)))
    return dense[rc].copy()


SUBSH_ITERS = [2, 2, 3, 5, 5]


def subsh_max(tx):
    lw, lh = int(np.log2(TX_W[tx])) - 2, int(np.log2(TX_H[tx])) - 2
    return SUBSH_ITERS[max(lw, lh)]


# ------------------------------------------------------------- task-list replay

@functools.lru_cache(None)
def replay_lib():
    """oracle/replay.c: walks the flat task lists through an oracle's DSP entries on the CPU."""
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "replay"], check=True, stdout=subprocess.DEVNULL)
    lib = C.CDLL(os.path.join(ROOT, "oracle", "libdav1d_replay.so"))
    for n in ("dav1d_replay_itx", "dav1d_replay_mc", "dav1d_replay_comp"):
        getattr(lib, n).restype = C.c_int
    lib.dav1d_replay_itx.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.dav1d_replay_mc.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    lib.dav1d_replay_comp.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    lib.dav1d_replay_recon_mt.restype = C.c_int
    lib.dav1d_replay_recon_mt.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                          C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
    for n in ("dav1d_replay_lf", "dav1d_replay_cdef", "dav1d_replay_lr"):
        getattr(lib, n).restype = C.c_int
    lib.dav1d_replay_lf.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_ssize_t, C.c_void_p]
    lib.dav1d_replay_cdef.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    lib.dav1d_replay_lr.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    return lib
