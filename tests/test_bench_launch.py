"""bench.py's launch contract on CPU: `--gpus N` with no WORLD_SIZE starts its N ranks itself (fresh children, gloo here) and prints ONE
line with n_gpus = N whose totals are the sum over the ranks; under an external launcher (RANK/WORLD_SIZE set) it does not spawn again.
The solver is tests/bench_stub.py: this tests the launch path, nothing is measured."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(LPBOX_BENCH_STUB="1", LPBOX_BENCH_BACKEND="gloo", **kw)
    return env


def _stub_total(world, batch):
    sys.path.insert(0, ROOT)
    import bench
    from bench_stub import StubBatch
    tot = 0
    for r in range(world):
        shard, _ = bench.lp_shard(type("A", (), {"config": 2, "batch": batch})(), r)
        sb = StubBatch(shard)
        tot += sum(sb.counters(i)[0] for i in range(batch))
    return tot


def test_gpus_2_starts_two_ranks_itself():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "8",
                          "--config", "2"], env=_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["stub"] is True
    assert d["scaling"] == "weak" and d["config"]["parallelism"] == "instance-sharded x2"
    assert d["detail"]["instance_iters_per_step"] > 0
    # whole-job aggregate: value * time = iterations of BOTH ranks over all steps
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - _stub_total(2, 8)) < 1e-6 * _stub_total(2, 8)


def test_default_invocation_with_gpus_2_also_self_launches():
    """What the driver types: no --config."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0", "--batch", "4"],
                         env=_env(), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 2 and "configs" not in d["detail"]


def test_under_an_external_launcher_it_does_not_spawn():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(2):
        env = _env(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                                       "--batch", "4", "--config", "2"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert [p.returncode for p in procs] == [0, 0], outs[0][1][-2000:] + outs[1][1][-2000:]
    assert json.loads(outs[0][0].strip().splitlines()[-1])["n_gpus"] == 2
    assert not [l for l in outs[1][0].splitlines() if l.startswith("{")]          # only rank 0 prints the line


def test_world_size_mismatch_is_refused():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--config", "2"],
                         env=_env(RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="1"),
                         capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE=2" in out.stderr


def test_a_failing_rank_stops_the_others_instead_of_hanging():
    """Rank 1 dies before the rendezvous (LPBOX_BENCH_FAIL_RANK, test hook): the parent must stop rank 0 and return non-zero quickly."""
    import time
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "4", "--config", "2"],
                         env=_env(LPBOX_BENCH_FAIL_RANK="1"), capture_output=True, text=True, timeout=240)
    assert out.returncode != 0 and "rank 1 exited" in out.stderr
    assert time.time() - t0 < 200
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
