#!/bin/bash
# Round-3 profiling session on the GPU box (run through gpurun): kernel-trace statistics of the bench step in the f32s and the bf16
# storage mode (medians + per-step breakdown), HBM traffic (PMC, FETCH_SIZE / WRITE_SIZE in separate passes) and SQ counters of the
# K1g forward kernels (f32: VALU kernel, f32s: matrix-pipe kernel, bf16: storage variant).  Writes under gpurun_out/prof_r3/; the
# summaries are copied to profiles/r3/ afterwards.
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=$PWD/gpurun_out/prof_r3; rm -rf $O; mkdir -p $O
for mode in f32s bf16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$mode -o bench -- python3 bench.py --dtype $mode --steps 30 --warmup 5 --cpu-sample 0 --no-alt --no-micro > $O/bench_trace_$mode.json 2> $O/bench_trace_$mode.err
  T=$(find $O/trace_$mode -name "*kernel_trace.csv" | head -1)
  python3 tools/trace_medians.py $T 70 > $O/bench_gmd_kernel_medians_$mode.txt
  python3 tools/step_breakdown.py $T > $O/bench_gmd_step_breakdown_$mode.txt 2>&1
  S=$(find $O/trace_$mode -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && head -40 $S > $O/bench_gmd_kernel_stats_$mode.csv
done
for dt in 0 2 1; do
  for Bp in 128 64; do
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_${dt}_$Bp -o p -- python3 tools/k1_fwd_only.py $Bp 8 $dt > /dev/null 2>&1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_${dt}_$Bp -o p -- python3 tools/k1_fwd_only.py $Bp 8 $dt > /dev/null 2>&1
  done
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/pmc_sq_${dt} -o p -- python3 tools/k1_fwd_only.py 128 6 $dt > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma_${dt} -o p -- python3 tools/k1_fwd_only.py 128 6 $dt > /dev/null 2>&1
done
for d in $(cd $O; ls -d pmc_*); do
  C=$(find $O/$d -name "*counter_collection.csv" | head -1)
  echo "== $d" >> $O/k1_pmc_summary.txt
  for k in scdm_fwd_kernel scdm_fwd_mm_kernel; do python3 tools/pmc_summary.py $C $k 2>/dev/null | sed "s/^/$k  /" >> $O/k1_pmc_summary.txt; done
done
cat $O/k1_pmc_summary.txt
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
du -sh $O
