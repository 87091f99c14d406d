"""Builds the HIP library (and, for CPU-side development, the SIMT-emulated twin).

    python -m dav1d_amd.build            # hipcc --offload-arch=gfx950 -> dav1d_amd/libdav1d_hip.so
    python -m dav1d_amd.build --emu      # g++ against tests/emu -> tests/emu/libdav1d_hip_emu.so

The emulated build compiles the *same* kernel sources with g++ against the fiber-based
SIMT shim under tests/emu; it is test infrastructure only (see tests/emu/hip/hip_runtime.h)
and is never loaded by the package, bench.py or smoke().
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["capi.hip", "chunk.hip", "frame.hip", "dsp_table.hip", "dsp_table_post.hip", "itx.hip", "mc.hip", "recon.hip", "intra_pair.hip", "intra_flow.hip", "intra_sb.hip", "lfmask.hip", "refmvs.hip", "peer.hip", "mcx.hip", "comp.hip", "cdef.hip", "loopfilter.hip", "ipred.hip", "lr.hip", "fg.hip"]
# host side of the pass-2 hand-off: plain C99 (the lister, its AV1 geometry), compiled with gcc
HOST_SOURCES = ["av1_host.c", "lister.c", "filter_lister.c", "lf_rects.c"]
HOST = os.path.join(HERE, "host")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CC = os.environ.get("CC", "gcc")


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(not os.path.exists(d) or os.path.getmtime(d) > t for d in deps)


def _obj_deps(obj, src, hdrs):
    """What `obj` was really compiled from: the compiler's own dependency file (-MMD, written next to the object) when there is one —
    a change to one kernel body then recompiles the translation units that include it, not all twenty — else every header."""
    d = obj[:-2] + ".d"
    if not os.path.exists(d) or not os.path.exists(obj):
        return [src] + hdrs
    try:
        txt = open(d).read().replace("\\\n", " ")
        deps = txt.split(":", 1)[1].split()
        deps = [x if os.path.isabs(x) else os.path.join(ROOT, x) for x in deps]
        return [x for x in deps if not x.startswith(("/opt/", "/usr/"))] or [src] + hdrs
    except (OSError, IndexError):
        return [src] + hdrs


def _deps():
    out = [os.path.join(ROOT, "include", "dav1d_hip.h")]
    for d in (CSRC, HOST):
        for f in os.listdir(d):
            if f.endswith(".h"):
                out.append(os.path.join(d, f))
    return out


def _host_jobs(objdir, hdrs, force):
    jobs, objs = [], []
    for s in HOST_SOURCES:
        src = os.path.join(HOST, s)
        obj = os.path.join(objdir, "host_" + s.replace(".c", ".o"))
        objs.append(obj)
        if force or _newer(obj, _obj_deps(obj, src, hdrs)):
            jobs.append([CC, "-std=gnu99", "-O2", "-fPIC", "-fvisibility=hidden", "-Wall", "-MMD"] + os.environ.get("DAV1D_HIP_HOST_CFLAGS", "").split() + ["-I" + os.path.join(ROOT, "include"), "-c", src, "-o", obj])
    return jobs, objs


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), r.stdout))
    return r.stdout


def build_hip(force=False, verbose=False):
    """hipcc cross-compiles for gfx950 (works without a GPU)."""
    out = os.path.join(HERE, "libdav1d_hip.so")
    objdir = os.path.join(HERE, "build", "hip")
    os.makedirs(objdir, exist_ok=True)
    hdrs = _deps()
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        if force or _newer(obj, _obj_deps(obj, src, hdrs)):
            jobs.append([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-MMD",
                         "-I" + os.path.join(ROOT, "include"), "-c", src, "-o", obj])
    hjobs, hobjs = _host_jobs(objdir, hdrs, force)
    jobs += hjobs
    with ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as ex:
        for o in ex.map(_run, jobs):
            if verbose and o:
                print(o)
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES] + hobjs
    if force or jobs or _newer(out, objs):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-lpthread"])
    return out


def build_emu(force=False):
    emu = os.path.join(ROOT, "tests", "emu")
    out = os.path.join(emu, "libdav1d_hip_emu.so")
    objdir = os.path.join(HERE, "build", "emu")
    os.makedirs(objdir, exist_ok=True)
    hdrs = _deps() + [os.path.join(emu, "hip", "hip_runtime.h"), os.path.join(emu, "emu_rt.cpp")]
    jobs = []
    srcs = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(emu, "emu_rt.cpp")]
    objs = []
    for src in srcs:
        obj = os.path.join(objdir, os.path.basename(src).rsplit(".", 1)[0] + ".o")
        objs.append(obj)
        if force or _newer(obj, _obj_deps(obj, src, hdrs)):
            jobs.append(["g++", "-O2", "-std=c++17", "-fPIC", "-w", "-MMD"] + os.environ.get("DAV1D_HIP_EMU_CXXFLAGS", "").split() + ["-I" + emu, "-I" + os.path.join(ROOT, "include"),
                         "-x", "c++", "-c", src, "-o", obj])
    hjobs, hobjs = _host_jobs(objdir, hdrs, force)
    jobs += hjobs
    objs += hobjs
    with ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as ex:
        list(ex.map(_run, jobs))
    if force or jobs or _newer(out, objs):
        _run(["g++", "-shared", "-fPIC", "-o", out] + objs + ["-lpthread"])
    return out


if __name__ == "__main__":
    if "--emu" in sys.argv:
        print(build_emu(force="--force" in sys.argv))
    else:
        print(build_hip(force="--force" in sys.argv, verbose=True))
