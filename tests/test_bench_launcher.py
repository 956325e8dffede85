"""bench.py --gpus N without a launcher starts the N ranks itself, and never reports a rank count it did not see.

CPU only: the children here print their environment instead of rendering (passes are independent and additive,
/root/reference/src/render.rs:214 and src/main.rs:256-288, which is what lets one process per GPU take every N-th pass).
"""
import importlib.util
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_spawn_ranks_sets_the_rendezvous_and_relays_rank_zero(capfd):
    bench = _bench()
    child = ("import os, json; print(json.dumps({k: os.environ.get(k) for k in "
             "('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}))")
    rc = bench.spawn_ranks(3, [sys.executable, "-c", child])
    out = capfd.readouterr().out.strip().splitlines()
    assert rc == 0
    assert len(out) == 1, "only rank 0's stdout is relayed"
    env = json.loads(out[0])
    assert env["RANK"] == "0" and env["LOCAL_RANK"] == "0" and env["WORLD_SIZE"] == "3"
    assert env["MASTER_ADDR"] == "127.0.0.1" and 0 < int(env["MASTER_PORT"]) < 65536


def test_spawn_ranks_every_rank_gets_its_own_rank(tmp_path):
    bench = _bench()
    child = "import os; open(os.path.join(%r, 'rank' + os.environ['RANK']), 'w').write(os.environ['LOCAL_RANK'])" % str(tmp_path)
    assert bench.spawn_ranks(4, [sys.executable, "-c", child]) == 0
    assert sorted(os.listdir(tmp_path)) == ["rank0", "rank1", "rank2", "rank3"]
    assert [open(tmp_path / ("rank%d" % r)).read() for r in range(4)] == ["0", "1", "2", "3"]


def test_spawn_ranks_fails_when_one_rank_fails_and_ends_the_others():
    bench = _bench()
    child = "import os, sys, time\nif os.environ['RANK'] == '1': sys.exit(3)\ntime.sleep(60)"
    t0 = time.time()
    rc = bench.spawn_ranks(3, [sys.executable, "-c", child])
    assert rc == 3
    assert time.time() - t0 < 30, "the surviving ranks are ended, not waited for"


def test_spawn_ranks_time_limit():
    bench = _bench()
    rc = bench.spawn_ranks(2, [sys.executable, "-c", "import time; time.sleep(60)"], timeout=1.0)
    assert rc == 124


def _run_bench(extra_env, *argv):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True, text=True,
                          timeout=300)


def test_bench_refuses_a_world_size_that_is_not_gpus():
    r = _run_bench({"WORLD_SIZE": "3", "RANK": "0", "LOCAL_RANK": "0"}, "--gpus", "2")
    assert r.returncode != 0
    assert "launcher started 3 rank(s)" in r.stderr
    assert "{" not in r.stdout, "no result line"


def test_bench_gpus_without_enough_devices_says_so():
    import torch

    if torch.cuda.device_count() >= 2:
        import pytest

        pytest.skip("this machine has the GPUs")
    r = _run_bench({}, "--gpus", "2")
    assert r.returncode != 0
    assert "--gpus 2 asked for" in r.stderr and "visible" in r.stderr
