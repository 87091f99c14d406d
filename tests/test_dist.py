"""The N > 1 path of bench.py on CPU: world_size-2 gloo run of the frame sharding + timing reduce."""
import os
import subprocess
import sys
import textwrap

import util


def test_frame_parallel_world2_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent("""
        import sys, json
        sys.path.insert(0, %r)
        import torch.distributed as dist
        from dav1d_amd import dist as dd
        rank, local, world = dd.env()
        dist.init_process_group("gloo")
        mine = dd.frames_of_rank(9, rank, world)
        dd.barrier(world)
        t = dd.max_over_ranks(1.0 + rank, world)          # slowest rank wins
        v = dd.job_throughput(100.0, 4, t, world)
        out = [None] * world
        dist.all_gather_object(out, mine)
        if rank == 0:
            print(json.dumps({"t": t, "v": v, "frames": out}))
        dist.destroy_process_group()
    """ % util.ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29517", str(script)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["t"] == 2.0 and d["v"] == 2 * 100.0 * 4 / 2.0
    assert d["frames"] == [[0, 2, 4, 6, 8], [1, 3, 5, 7]]
