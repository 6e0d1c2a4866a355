"""The N>1 control flow of bench.py (one process per GPU, barrier on both sides of
the timed region, MAX over ranks, rank 0 reports the whole-job aggregate)
rehearsed on CPU with gloo, world_size 2.  The data path has no collective, so
this is all the distributed logic there is."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    import bench
    rank, world, local_rank = bench.dist_env()
    dist = bench.dist_init(world, "gloo")
    calls = []
    def step():
        calls.append(1)
        time.sleep(0.002 * (rank + 1))      # rank 1 is the slow one
    dt_local = bench.timed_steps(step, steps=5, warmup=2, sync=lambda: None, world=world, dist=dist)
    dt = bench.max_over_ranks(dt_local, world, dist, "cpu")
    with open(os.path.join(os.environ["HVC_TEST_OUT"], "rank%%d.json" %% rank), "w") as f:
        json.dump({"rank": rank, "world": world, "calls": len(calls), "dt_local": dt_local, "dt": dt,
                   "value": bench.whole_job_mpixels(world, 4, 5, dt)}, f)
    dist.destroy_process_group()
""") % ROOT


def test_two_rank_gloo_timing_closure(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HVC_TEST_OUT=str(tmp_path))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)],
                         capture_output=True, text=True, env=env, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    recs = [json.loads((tmp_path / ("rank%d.json" % r)).read_text()) for r in (0, 1)]
    assert [r["rank"] for r in recs] == [0, 1] and all(r["world"] == 2 for r in recs)
    assert all(r["calls"] == 7 for r in recs)                      # 2 warm-up + exactly 5 timed
    assert recs[0]["dt"] == recs[1]["dt"]                          # MAX over ranks, same on both
    assert recs[0]["dt"] >= max(r["dt_local"] for r in recs) - 1e-9
    assert recs[1]["dt_local"] >= 5 * 0.004                        # the slow rank bounds the job
    # the barrier makes the fast rank wait: its local bracket also covers the slow rank's steps
    assert recs[0]["dt_local"] >= 5 * 0.004 * 0.9
    want = 2 * 4 * 5 * 1920 * 1080 / recs[0]["dt"] / 1e6            # whole job: both ranks' frames / max time
    assert abs(recs[0]["value"] - want) < 1e-6 * want


def test_single_rank_needs_no_process_group():
    sys.path.insert(0, ROOT)
    import bench
    n = []
    dt = bench.timed_steps(lambda: n.append(1), steps=3, warmup=1, sync=lambda: None, world=1)
    assert len(n) == 4 and dt >= 0
    assert bench.max_over_ranks(dt, 1) == dt
