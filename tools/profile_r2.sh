#!/bin/bash
# Round-2 profiling session on the GPU box (run through gpurun): kernel-trace statistics of the bench step with medians,
# HBM traffic (PMC, separate passes) of the K1 kernels, SQ counters of the fused K1 backward, FETCH_SIZE calibration.
# Writes under gpurun_out/prof_r2/; the summaries are copied to profiles/r2/ afterwards.
set -x
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=$PWD/gpurun_out/prof_r2; mkdir -p $O
ARGS="${BENCH_ARGS:---steps 30 --warmup 5 --cpu-sample 0 --no-alt --no-micro}"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python3 bench.py $ARGS > $O/bench_trace.json 2> $O/bench_trace.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_bwd -o p -- python3 tools/k1_bwd_only.py 128 8 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_bwd -o p -- python3 tools/k1_bwd_only.py 128 8 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_fwd -o p -- python3 tools/k1_fwd_only.py 128 8 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_fwd -o p -- python3 tools/k1_fwd_only.py 128 8 > /dev/null 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_64 -o p -- python3 tools/k1_only.py 8 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_64 -o p -- python3 tools/k1_only.py 8 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/pmc_sq_bwd -o p -- python3 tools/k1_bwd_only.py 128 6 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/pmc_sq_fwd -o p -- python3 tools/k1_fwd_only.py 128 6 > /dev/null 2>&1
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/ubench/read_width.hip -o /tmp/read_width && \
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/cal_fetch -o p -- /tmp/read_width && \
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/cal_write -o p -- /tmp/read_width
find $O -name "*.csv" | head -50
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 tools/trace_medians.py $T 60 > $O/bench_kernel_medians.txt
python3 tools/step_breakdown.py $T > $O/bench_step_breakdown.txt 2>&1
for d in pmc_fetch_bwd pmc_write_bwd pmc_fetch_fwd pmc_write_fwd pmc_fetch_64 pmc_write_64 pmc_sq_bwd pmc_sq_fwd cal_fetch cal_write; do
  C=$(find $O/$d -name "*counter_collection.csv" | head -1)
  echo "== $d" >> $O/pmc_summary.txt
  for k in scdm_fwd scdm_bwd_fused scdm_bwd_rows scdm_bwd_cols copy_k; do python3 tools/pmc_summary.py $C $k 2>/dev/null | sed "s/^/$k  /" >> $O/pmc_summary.txt; done
done
cat $O/pmc_summary.txt
find $O -name "*kernel_trace.csv" -path "*trace/*" -exec cp {} $O/bench_kernel_trace.csv \;
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
# keep the merge-back small: only CSVs
du -sh $O
