"""dav1d's task loop with the binding's frames ending on N devices of THIS process in turn (Dav1dHipGlueOptions.n_devices; reference
src/internal.h:354-388: dav1d is one process with n_fc frame contexts): bench.py's dav1d_task_loop_n_gpus leg, run as a child so that
whatever happens in here stays out of the bench line.  Prints one JSON object.
    python tools/task_loop_n_devices.py N [--width W --height H --bpc B --frames F --emu]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("n", type=int)
    ap.add_argument("--width", type=int, default=7680)
    ap.add_argument("--height", type=int, default=4320)
    ap.add_argument("--bpc", type=int, default=10)
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--threads", type=int, default=min(64, os.cpu_count() or 8))
    ap.add_argument("--intra-pct", type=int, default=10)
    ap.add_argument("--emu", action="store_true", help="the SIMT-emulated library ($DAV1D_EMU_DEVICES emulated devices): small sizes only")
    a = ap.parse_args()
    if a.emu:
        os.environ.setdefault("DAV1D_EMU_DEVICES", str(max(2, a.n)))
    import ctypes as C
    import hooked_util as hk
    import util
    from dav1d_amd import _lib
    path = util.emu_lib_path() if a.emu else _lib.DEFAULT_PATH
    if hk.lib() is None:
        print(json.dumps({"status": "skipped: oracle/_ref_hooked is not built"}))
        return
    have = C.CDLL(path).dav1d_hip_device_count()
    if have < a.n:
        print(json.dumps({"status": "skipped: %d device(s) visible to this process, %d asked for" % (have, a.n)}))
        return
    out = {}
    for n in (1, a.n):            # one device first: the same chain, the same process, the same box
        r = hk.task_loop_rate(path, a.width, a.height, a.bpc, tiles=(4, 1), threads=a.threads, frame_delay=8, frames=a.frames, n_devices=n, intra_pct=a.intra_pct)
        out["devices_%d" % n] = {k: r.get(k) for k in ("fps", "ms_per_frame", "steady_state", "devices", "parity", "n_fc", "worker_threads")}
    out["steady_state"] = out["devices_%d" % a.n]["steady_state"]
    out["parity"] = r["parity"]
    out["one_device_steady_state_fps"] = (out["devices_1"]["steady_state"] or {}).get("fps")
    out["workload"] = r["workload"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
