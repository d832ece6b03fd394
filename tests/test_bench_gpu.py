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
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "pairs/s" and cb["value"] > 0 and cb["cores"] >= 1 and "oracle" in cb["sample"]
