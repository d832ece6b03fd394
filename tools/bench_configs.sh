#!/bin/bash
# The BASELINE.json configs other than the headline one, each in its own dtype, on ONE GPU, on the tree as it is -- the refresh the round-5 review
# asked for (item 5): config 1 (forward only, fp32), config 3 (B = 64 on one GPU and the 16-pair shard a 4-GPU run gives each rank), config 4
# (B = 128 bf16 on one GPU and the 16-pair shard of an 8-GPU run).  The full-batch and the shard line of a config together bound its STRONG
# scaling before any communication.  One JSON line per config under OUT (default gpurun_out/configs), a table on stdout.
#     usage:  tools/bench_configs.sh [OUT] [steps]
OUT=${1:-gpurun_out/configs}; STEPS=${2:-10}
mkdir -p "$OUT"
run() {   # name, bench flags...
  name=$1; shift
  python bench.py --steps "$STEPS" --warmup 3 --cpu-sample 0 --no-micro --no-alt --graph on "$@" > "$OUT/$name.json" 2> "$OUT/$name.log" || echo "$name: bench.py failed (see $OUT/$name.log)"
}
run config1_fwd_only_B32_T64_N20_d512_f32       --model qave --fwd-only --B 32 --T 64 --N 20 --d 512 --dtype f32 --graph off
run config3_B64_T256_N25_d1024_f32s             --B 64 --T 256 --N 25 --d 1024 --dtype f32s
run config3_shard_B16_T256_N25_d1024_f32s       --B 16 --T 256 --N 25 --d 1024 --dtype f32s
run config4_B128_T512_N25_d1024_bf16            --B 128 --T 512 --N 25 --d 1024 --dtype bf16
run config4_shard_B16_T512_N25_d1024_bf16       --B 16 --T 512 --N 25 --d 1024 --dtype bf16
python - "$OUT" <<'PY'
import json, os, sys
out = sys.argv[1]
rows = []
for f in sorted(os.listdir(out)):
    if not f.endswith(".json"):
        continue
    try:
        d = json.loads(open(os.path.join(out, f)).read().strip().splitlines()[-1])
    except Exception as e:                      # noqa: BLE001
        rows.append((f[:-5], "no result", "", "", "")); continue
    r = d.get("roofline") or {}
    rows.append((f[:-5], f"{d['value']:.0f} pairs/s", f"{d['ms_per_step']:.2f} ms/step ({d.get('value_mode', '').split(':')[0]})",
                 f"eager {d['eager']['ms_per_step']:.2f} ms", f"K1g fwd {r.get('mean_launch_us')} us = {r.get('frac')} of 8 TB/s"))
w = [max(len(r[i]) for r in rows) for i in range(5)]
for r in rows:
    print("  ".join(c.ljust(w[i]) for i, c in enumerate(r)))
PY
