"""CPU: `python bench.py --gpus N` started WITHOUT a launcher must start its N ranks itself (one child per GPU, before the parent
touches a GPU), rendezvous them, and print rank 0's one JSON line with n_gpus = N -- the command the driver's scaling run uses when
it does not go through torchrun.  `--rehearse` keeps everything but the engine (gloo, a sleep as the workload), so this runs here."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rehearse", *flags], capture_output=True, text=True, timeout=240, env=e)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_spawn_two_ranks_weak_scaling():
    d = _run("--gpus", "2", "--steps", "3", "--warmup", "1")
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 16
    # rank 1 sleeps 40 ms per step: the job time is the MAX over ranks, the images are the SUM over ranks
    assert 40.0 <= d["ms_per_step"] < 120.0
    assert abs(d["value"] - 16 / (d["ms_per_step"] * 1e-3)) < 0.02 * d["value"]


def test_spawn_strong_scaling_prompt_shards():
    d = _run("--gpus", "3", "--steps", "2", "--warmup", "0", "--prompts", "10", "--batch", "4")
    assert d["n_gpus"] == 3 and d["scaling"] == "strong" and d["config"]["global_batch"] == 10
    assert abs(d["value"] - 10 / (d["ms_per_step"] * 1e-3)) < 0.02 * d["value"]     # every prompt counted once, whatever the split


def test_torchrun_style_environment_is_honoured():
    """Under a launcher (WORLD_SIZE set) the process must NOT spawn: a lone rank 0 of world 1 prints its own line."""
    d = _run("--gpus", "1", "--steps", "1", "--warmup", "0", env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    assert d["n_gpus"] == 1


def test_spawn_cfg_pair_split():
    """`--cfg-pair-split` through the spawn path: four ranks = two CFG pairs, every pair counts its prompts once."""
    d = _run("--gpus", "4", "--steps", "2", "--warmup", "0", "--cfg-pair-split")
    assert d["n_gpus"] == 4 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 2 * 16 and "cfg-pair2" in d["config"]["parallelism"]      # two pairs x (2 x batch 8) prompts
    assert abs(d["value"] - 32 / (d["ms_per_step"] * 1e-3)) < 0.02 * d["value"]


def test_a_failing_rank_ends_the_job_with_its_output():
    """One rank dies at start-up: the parent must not hang in the survivors' rendezvous -- it ends them, exits non-zero and shows the
    failing rank's output."""
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e["SDE_BENCH_TEST_FAIL_RANK"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rehearse", "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=120, env=e)
    assert p.returncode == 7, (p.returncode, p.stderr[-1500:])
    assert "rank 1 exited with code 7" in p.stderr and "injected start-up failure" in p.stderr
