"""bench.py launcher logic on the CPU box (no GPU): --gpus N without a launcher environment must start N child ranks
(never fall back to a one-GPU run) and propagate their failure; a world size that differs from --gpus is an error."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True,
                          timeout=300)


def test_gpus_n_spawns_n_ranks_and_fails_loudly_without_gpus():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--cpu-sample", "0"])
    assert r.returncode != 0
    assert "starting 2 ranks" in r.stderr and "--nproc-per-node=2" in r.stderr
    assert r.stdout.strip() == ""                       # no result line: nothing that looks like a 1-GPU measurement


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "4"], env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in r.stderr
    r = _run(["--gpus", "1"], env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_no_gpu_is_an_error_not_a_fallback():
    r = _run(["--gpus", "1"])
    assert r.returncode != 0 and "no CPU fallback" in r.stderr and r.stdout.strip() == ""
