"""INTEGRATION.md shows the code a dav1d maintainer adds.  The self-contained pieces of it — the frame descriptor filled from a
Dav1dFrameContext and the two Dav1dPicAllocator callbacks — are compiled here against the reference's OWN headers (syntax and
types only: gcc -fsyntax-only), so the document cannot drift away from either side of the boundary."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
INC = os.path.join(ROOT, "oracle", "_ref", "inc")


def _blocks():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    return [re.sub(r"(?m)^  ", "", b) for b in re.findall(r"```c\n(.*?)```", text, re.S)]


def _function(src, name):
    """the text of `static ... name(...) { ... }` inside a snippet"""
    m = re.search(r"(?m)^static [^\n;]*\b%s\(" % name, src)
    assert m, name
    i = src.index("{", m.end())
    depth, j = 0, i
    while True:
        depth += {"{": 1, "}": -1}.get(src[j], 0)
        j += 1
        if depth == 0:
            break
    return src[m.start():j]


@pytest.mark.skipif(not os.path.isdir(REF) or not os.path.isdir(INC), reason="needs the reference tree and oracle/_ref (built by __graft_entry__.build())")
def test_glue_of_integration_md_compiles_against_the_reference_headers(tmp_path):
    blocks = _blocks()
    desc = next(b for b in blocks if "hip_frame_desc" in b)
    alloc = next(b for b in blocks if "hip_alloc_picture" in b)
    src = "\n".join([
        '#include "config.h"', "#include <errno.h>", "#include <stdlib.h>", "#include <string.h>",
        '#include "src/internal.h"', '#include "common/frame.h"', '#include "dav1d/picture.h"', '#include "dav1d_hip.h"',
        _function(desc, "hip_frame_desc"), _function(alloc, "hip_alloc_picture"), _function(alloc, "hip_release_picture"),
        "/* the allocator as Dav1dSettings takes it */",
        "Dav1dPicAllocator hip_allocator(Dav1dHipContext *ctx) { Dav1dPicAllocator a = { ctx, hip_alloc_picture, hip_release_picture }; return a; }",
        "void use(Dav1dHipFrameDesc *d, const Dav1dFrameContext *f) { hip_frame_desc(d, f); }", ""])
    f = tmp_path / "glue.c"
    f.write_text(src)
    cmd = ["gcc", "-std=c11", "-D_GNU_SOURCE", "-fsyntax-only", "-Wall", "-Werror=implicit-function-declaration", "-Werror=incompatible-pointer-types",
           "-Werror=int-conversion", "-I" + INC, "-I" + REF, "-I" + os.path.join(REF, "include"), "-I" + os.path.join(REF, "include", "dav1d"),
           "-I" + os.path.join(REF, "src"), "-I" + os.path.join(ROOT, "include"), str(f)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
