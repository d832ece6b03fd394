#!/bin/bash
# GPU job of round 4 (i): fused heads after the batched last-arriver loads (parity + kernel time), the default bench line, and a kernel trace
# of the step with the glue and library-GEMM launches listed.
mkdir -p gpurun_out/r4i
(timeout 600 python -m pytest tests/test_head_gemm_gpu.py tests/test_bench_gpu.py -x -q -m gpu 2>&1 | tail -4) > gpurun_out/r4i/pytest.txt
(python bench.py 2>gpurun_out/r4i/bench_err.txt | tail -1) > gpurun_out/r4i/bench_default.json
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4i/prof; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python3 bench.py --steps 30 --warmup 5 --cpu-sample 0 --no-alt --no-micro > $O/bench.json 2> $O/bench.err
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 tools/trace_medians.py $T 70 > gpurun_out/r4i/bench_gmd_kernel_medians.txt
python3 tools/step_breakdown.py $T --glue --gemms > gpurun_out/r4i/bench_gmd_step_breakdown.txt 2>&1
rm -rf $O/trace
cat gpurun_out/r4i/pytest.txt; cut -c1-300 gpurun_out/r4i/bench_default.json; echo; cat gpurun_out/r4i/bench_gmd_step_breakdown.txt | cut -c1-170
