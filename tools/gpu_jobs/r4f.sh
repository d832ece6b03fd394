#!/bin/bash
# GPU job of round 4 (f): the LSTM layers' GEMMs on the own split-on-load kernels (TSG_LSTM_GEMM=own) vs planes + library (lib): parity of
# the LSTM / model tests with the switch on, then the step time A/B (alternating processes on one box); the new ANet T=256 tests.
mkdir -p gpurun_out/r4f
(TSG_LSTM_GEMM=own timeout 900 python -m pytest tests/test_lstm_gpu.py tests/test_models_gpu.py tests/test_fullsize_gpu.py tests/test_config4_gpu.py -x -q -m gpu 2>&1 | tail -8) > gpurun_out/r4f/pytest_lstm_own.txt
(timeout 600 python -m pytest tests/test_config_anet256_gpu.py -x -q -m gpu 2>&1 | tail -8) > gpurun_out/r4f/pytest_anet256.txt
for i in 1 2 3; do
  (TSG_LSTM_GEMM=lib python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-200 | sed "s/^/LIB  /")
  (TSG_LSTM_GEMM=own python bench.py --no-alt --cpu-sample 0 --no-micro --graph on 2>/dev/null | tail -1 | cut -c1-200 | sed "s/^/OWN  /")
done > gpurun_out/r4f/bench_lstm_own_gemm_ab.txt
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4f/prof; mkdir -p $O
TSG_LSTM_GEMM=own rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python3 bench.py --steps 30 --warmup 5 --cpu-sample 0 --no-alt --no-micro > $O/bench.json 2> $O/bench.err
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 tools/trace_medians.py $T 70 > gpurun_out/r4f/bench_gmd_kernel_medians_own.txt
python3 tools/step_breakdown.py $T > gpurun_out/r4f/bench_gmd_step_breakdown_own.txt 2>&1
rm -rf $O/trace
cat gpurun_out/r4f/pytest_lstm_own.txt gpurun_out/r4f/pytest_anet256.txt gpurun_out/r4f/bench_lstm_own_gemm_ab.txt; head -12 gpurun_out/r4f/bench_gmd_step_breakdown_own.txt
