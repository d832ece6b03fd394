#!/bin/bash
# GPU job of round 4 (bb): bench lines at the other BASELINE configurations' shapes on this round's tree + a kernel trace of config 4's per-GPU shard in bf16
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r4bb; rm -rf $O; mkdir -p $O
C="--cpu-sample 0 --no-alt --no-micro"
python3 bench.py --B 32 --T 64 --N 20 --d 512 --fwd-only --dtype f32 $C > $O/bench_config1_fwd_only_f32.json 2> $O/c1.err
python3 bench.py --B 64 --T 256 --N 25 $C > $O/bench_config3_B64_T256_N25.json 2> $O/c3.err
python3 bench.py --B 64 --T 256 --N 25 --dtype bf16 $C > $O/bench_config3_B64_T256_N25_bf16.json 2> $O/c3b.err
python3 bench.py --B 16 --T 256 --N 25 $C > $O/bench_config3_perGPU_B16_T256_N25.json 2> $O/c3p.err
python3 bench.py --B 16 --T 512 --N 25 $C > $O/bench_config4_perGPU_B16_T512_N25.json 2> $O/c4p.err
python3 bench.py --B 16 --T 512 --N 25 --dtype bf16 $C > $O/bench_config4_perGPU_B16_T512_N25_bf16.json 2> $O/c4pb.err
python3 bench.py --B 128 --T 512 --N 25 --dtype bf16 --steps 10 --warmup 3 $C > $O/bench_config4_B128_T512_N25_bf16.json 2> $O/c4b.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c4 -o bench -- python3 bench.py --B 16 --T 512 --N 25 --dtype bf16 --steps 30 --warmup 5 $C > $O/bench_trace_c4.json 2> $O/bench_trace_c4.err
T=$(find $O/trace_c4 -name "*kernel_trace.csv" | head -1)
python3 tools/trace_medians.py $T 60 > $O/bench_config4_perGPU_bf16_kernel_medians.txt
python3 tools/step_breakdown.py $T > $O/bench_config4_perGPU_bf16_step_breakdown.txt 2>&1
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
for f in $O/bench_config*.json; do echo $(basename $f) $(python3 -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'], d['dtype'], d['roofline']['frac'])" 2>&1 | tail -1); done
cat $O/bench_config4_perGPU_bf16_step_breakdown.txt | head -12
