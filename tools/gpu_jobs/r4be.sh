#!/bin/bash
# GPU job of round 4 (be): whole GPU suite + smoke + default / bf16 lines + the per-GPU shard lines of configs 3 / 4 after the small-batch LSTM changes
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=$PWD/gpurun_out/r4be; rm -rf $O; mkdir -p $O
(timeout 3000 python -m pytest tests -x -q -m gpu 2>&1 | tail -5) > $O/pytest_gpu_full.txt
(python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2) > $O/smoke.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --dtype bf16 --no-alt --cpu-sample 0 > $O/bench_bf16.json 2> $O/bench_bf16.err
C="--cpu-sample 0 --no-alt --no-micro"
python3 bench.py --B 16 --T 256 --N 25 $C > $O/bench_config3_perGPU_B16_T256_N25.json 2> $O/c3p.err
python3 bench.py --B 16 --T 256 --N 25 --dtype bf16 $C > $O/bench_config3_perGPU_B16_T256_N25_bf16.json 2> $O/c3pb.err
python3 bench.py --B 16 --T 512 --N 25 $C > $O/bench_config4_perGPU_B16_T512_N25.json 2> $O/c4p.err
python3 bench.py --B 16 --T 512 --N 25 --dtype bf16 $C > $O/bench_config4_perGPU_B16_T512_N25_bf16.json 2> $O/c4pb.err
TSG_LSTM_PAD=0 TSG_LSTM_NW=8 python3 bench.py --B 16 --T 512 --N 25 $C > $O/bench_config4_perGPU_B16_T512_N25_nopad.json 2> $O/c4pn.err
TSG_LSTM_PAD=0 TSG_LSTM_NW=8 python3 bench.py --B 16 --T 512 --N 25 --dtype bf16 $C > $O/bench_config4_perGPU_B16_T512_N25_bf16_nopad.json 2> $O/c4pbn.err
cat $O/pytest_gpu_full.txt $O/smoke.txt
for f in $O/bench_*.json; do echo $(basename $f) $(python3 -c "import json,sys; d=json.load(open('$f')); print(d['value'], d['ms_per_step'], d['dtype'], d['roofline']['frac'])" 2>&1 | tail -1); done
