import ctypes as C, os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import torch
torch.cuda.init()
import util
ctx = util.make_context("hip")
ident = (C.c_uint8 * 128)()
print("id", ctx.lib.dav1d_hip_peer_unique_id(ident))
h = C.c_void_p()
print("open", ctx.lib.dav1d_hip_peer_open(ctx.h, C.byref(h), ident, 0, 1))
os.system("grep rccl /proc/%d/maps | awk '{print $6}' | sort -u" % os.getpid())
