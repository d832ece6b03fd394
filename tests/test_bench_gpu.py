"""bench.py's one-line JSON contract on a real GPU (short run): the keys the driver and the judge read are present and
consistent -- metric / value / unit / n_gpus / steps / warmup / ms_per_step / scaling / dtype / config.workload, the `roofline`
object of the dominant hot-path kernel (HBM bound, achieved = algorithmic bytes / live event-timed launch duration, frac =
achieved / peak <= 1, traffic from the committed PMC pass) and the `cpu_baseline` object (oracle port on the host cores)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_json_contract():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "2", "--cpu-sample", "2", "--no-micro",
                        "--no-alt"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]                   # stdout carries exactly one line
    j = json.loads(lines[0])
    assert j["metric"] == "clip-query pairs/sec fwd+bwd at B=64,T=128,d=1024" and j["unit"] == "pairs/s"
    assert j["n_gpus"] == 1 and j["rccl_ranks"] == 1 and j["steps"] == 3 and j["warmup"] == 2
    assert j["higher_is_better"] is True and j["scaling"] == "weak" and j["vs_baseline"] is None and j["data"] == "synthetic"
    assert j["dtype"] == "f32s" and "gmd_train_step" in j["config"]["workload"] and j["config"]["global_batch"] == 64
    assert abs(j["value"] - 64 * 1000.0 / j["ms_per_step"]) / j["value"] < 1e-3
    roof = j["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == 8000.0
    assert 0.0 < roof["frac"] <= 1.0 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    assert roof["pairs_per_launch"] == 128 and roof["launches_timed"] == 6          # two encoder blocks x 3 timed steps
    assert abs(roof["achieved"] - roof["alg_bytes_per_launch"] / roof["mean_launch_us"] / 1e3) / roof["achieved"] < 1e-2
    assert roof["traffic"] is not None and 0.95 < roof["traffic"] / roof["alg_bytes_per_launch"] < 1.1
    assert roof["traffic_source"].startswith("profiles/") and "gate-fused" in roof["alg_bytes_formula"]
    assert j["value_mode"].startswith("eager") and j["graph_replay_in_process"] is None
    assert j["skipped_updates"] == 0 and "value_f32" in j and "value_bf16" in j
    assert roof["timing"].startswith("HIP event pair of the launch itself") and roof["around_call_mean_us"] >= roof["mean_launch_us"]
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "pairs/s" and cb["value"] > 0 and cb["cores"] >= 1 and "oracle" in cb["sample"]


def _bench(args, env_extra=None, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_multi_gpu_code_path_at_world_1():
    """The WHOLE N > 1 code path of bench.py on one GPU (TSG_FORCE_DIST=1: a world-1 RCCL process group): the flat-gradient
    exchange in the eager step, then the same step replayed from two HIP graphs in the same process with `dp.exchange_static`
    between them (what --gpus N > 1 measures by default), strong-scaling batch sharding, per-rank host enqueue times."""
    j = _bench(["--steps", "3", "--warmup", "2", "--cpu-sample", "0", "--no-micro", "--no-alt", "--graph", "on", "--scaling", "strong",
                "--B", "16"], env_extra={"TSG_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29531"})
    assert j["n_gpus"] == 1 and j["rccl_ranks"] == 1 and j["scaling"] == "strong"
    assert j["config"]["global_batch"] == 16 and j["config"]["pairs_per_gpu"] == 16 and "strong scaling" in j["config"]["workload"]
    g = j["graph_replay_in_process"]
    assert g is not None and "error" not in g, g
    assert g["finite"] and g["value"] > 0 and len(g["host_enqueue_ms_per_step_by_rank"]) == 1
    # (two graph launches vs ~400 kernel launches -- but with the runtime's graph packet capture off (_runtime_env.py) a launch dispatches its nodes
    # one by one, so the replay's host time is of the order of the eager step's, not a fiftieth of it: only its presence is checked)
    assert g["host_enqueue_ms_per_step_by_rank"][0] > 0 and j["host_enqueue_ms_per_step_by_rank"][0] > 0
    assert j["skipped_updates"] == 0 and g["skipped_updates"] == 0
    assert j["value_mode"].startswith(("eager", "graph_replay")) and j["value"] == max(j["eager"]["value"], g["value"])
    assert abs(j["value"] - 16 * 1000.0 / j["ms_per_step"]) / j["value"] < 1e-3
    assert j["roofline"]["pairs_per_launch"] == 32                                                  # 16 original + 16 shuffled videos


def test_bench_bf16_storage_line():
    """`bench.py --dtype bf16`: the bf16 STORAGE step as its own labelled line (BASELINE configs 2 / 4); the roofline object is
    then the TSG_BF16 K1g launch with 2-byte activations in its algorithmic byte count."""
    j = _bench(["--steps", "3", "--warmup", "2", "--cpu-sample", "0", "--no-micro", "--no-alt", "--dtype", "bf16"])
    assert j["dtype"] == "bf16" and "bf16 STORAGE" in j["config"]["workload"]
    roof = j["roofline"]
    assert roof["pairs_per_launch"] == 128 and "e=2" in roof["alg_bytes_formula"]
    assert roof["alg_bytes_per_launch"] == 128 * ((3 * 128 + 2 * 20) * 1024 * 2 + 128 * 20 * 4)
    assert 0.0 < roof["frac"] <= 1.0
