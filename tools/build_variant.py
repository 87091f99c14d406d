#!/usr/bin/env python3
"""A variant of the HIP library for A/B runs on the GPU box: the named translation units compiled with extra flags, everything else the
tree's own objects.  tools/build_variant.py <name> "<flags>" file.hip [file.hip ...]  ->  dav1d_amd/build/variants/<name>.so
(tools/layout_sweep.py --lib / tools/ab_step.sh load them)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dav1d_amd import build as b   # noqa: E402

name, flags, files = sys.argv[1], sys.argv[2].split(), sys.argv[3:]
b.build_hip()
vdir = os.path.join(b.HERE, "build", "variants")
odir = os.path.join(vdir, name + "_obj")
os.makedirs(odir, exist_ok=True)
objdir = os.path.join(b.HERE, "build", "hip")
objs = []
jobs = []
for s in b.SOURCES:
    if s in files:
        o = os.path.join(odir, s.replace(".hip", ".o"))
        jobs.append([b.HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-I" + os.path.join(ROOT, "include")] + flags +
                    ["-c", os.path.join(b.CSRC, s), "-o", o])
    else:
        o = os.path.join(objdir, s.replace(".hip", ".o"))
    objs.append(o)
objs += [os.path.join(objdir, "host_" + s.replace(".c", ".o")) for s in b.HOST_SOURCES]
with ThreadPoolExecutor(8) as ex:
    list(ex.map(b._run, jobs))
out = os.path.join(vdir, name + ".so")
b._run([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-lpthread"])
print(out)
