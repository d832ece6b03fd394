#!/bin/bash
# Round-4 profiling session on the GPU box (run through gpurun).  Writes summaries under gpurun_out/prof_r4/ (copied to profiles/r4/).
#  1. kernel-trace statistics of the bench step (f32s headline mode, bf16 storage mode): per-kernel medians + per-step breakdown
#  2. the K1g BACKWARD's HBM traffic (FETCH_SIZE x2 + WRITE_SIZE, separate passes) and SQ counters (round-3 review item 3)
#  3. the bf16-storage K1g forward with the packed-f16 score loop: traffic + SQ counters
#  4. the north star's literal kernel IN the step: bench.py --predictor self_attn under kernel-trace, and PMC traffic of the K2
#     kernels from that same step (inputs fresh from the projection GEMMs, not cache-resident micro-benchmark buffers)
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
O=$PWD/gpurun_out/prof_r4; rm -rf $O; mkdir -p $O
for mode in f32s bf16; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$mode -o bench -- python3 bench.py --dtype $mode --steps 30 --warmup 5 --cpu-sample 0 --no-alt --no-micro > $O/bench_trace_$mode.json 2> $O/bench_trace_$mode.err
  T=$(find $O/trace_$mode -name "*kernel_trace.csv" | head -1)
  python3 tools/trace_medians.py $T 70 > $O/bench_gmd_kernel_medians_$mode.txt
  python3 tools/step_breakdown.py $T > $O/bench_gmd_step_breakdown_$mode.txt 2>&1
  S=$(find $O/trace_$mode -name "*kernel_stats.csv" | head -1); [ -n "$S" ] && head -40 $S > $O/bench_gmd_kernel_stats_${mode}_summary.csv
done
# 2. K1g backward
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_bwd_fetch -o p -- python3 tools/k1_bwd_only.py 128 8 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_bwd_write -o p -- python3 tools/k1_bwd_only.py 128 8 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/pmc_bwd_sq -o p -- python3 tools/k1_bwd_only.py 128 6 > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_bwd_mfma -o p -- python3 tools/k1_bwd_only.py 128 6 > /dev/null 2>&1
for d in pmc_bwd_fetch pmc_bwd_write pmc_bwd_sq pmc_bwd_mfma; do
  C=$(find $O/$d -name "*counter_collection.csv" | head -1)
  echo "== $d" >> $O/k1g_bwd_pmc_summary.txt
  python3 tools/pmc_summary.py $C scdm_bwd_fused_kernel 2>/dev/null | sed "s/^/scdm_bwd_fused_kernel  /" >> $O/k1g_bwd_pmc_summary.txt
done
# 3. bf16 K1g forward (dtype 1)
for Bp in 128; do
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fwd16_fetch -o p -- python3 tools/k1_fwd_only.py $Bp 8 1 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_fwd16_write -o p -- python3 tools/k1_fwd_only.py $Bp 8 1 > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --kernel-trace --output-format csv -d $O/pmc_fwd16_sq -o p -- python3 tools/k1_fwd_only.py $Bp 6 1 > /dev/null 2>&1
done
for d in pmc_fwd16_fetch pmc_fwd16_write pmc_fwd16_sq; do
  C=$(find $O/$d -name "*counter_collection.csv" | head -1)
  echo "== $d" >> $O/k1g_fwd_bf16_pmc_summary.txt
  python3 tools/pmc_summary.py $C scdm_fwd_ws_kernel 2>/dev/null | sed "s/^/scdm_fwd_ws_kernel  /" >> $O/k1g_fwd_bf16_pmc_summary.txt
done
# 4. K2 in the step (--predictor self_attn): kernel trace, then traffic of the mha kernels from the SAME command
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_sa -o bench -- python3 bench.py --predictor self_attn --steps 20 --warmup 5 --cpu-sample 0 --no-alt --no-micro > $O/bench_self_attn.json 2> $O/bench_self_attn.err
T=$(find $O/trace_sa -name "*kernel_trace.csv" | head -1)
python3 tools/trace_medians.py $T 40 > $O/bench_self_attn_kernel_medians.txt
python3 tools/step_breakdown.py $T > $O/bench_self_attn_step_breakdown.txt 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_sa_fetch -o p -- python3 bench.py --predictor self_attn --steps 6 --warmup 3 --cpu-sample 0 --no-alt --no-micro --graph off > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_sa_write -o p -- python3 bench.py --predictor self_attn --steps 6 --warmup 3 --cpu-sample 0 --no-alt --no-micro --graph off > /dev/null 2>&1
for d in pmc_sa_fetch pmc_sa_write; do
  C=$(find $O/$d -name "*counter_collection.csv" | head -1)
  echo "== $d" >> $O/k2_in_step_pmc_summary.txt
  for k in mha_fwd_split_kernel mha_bwd_dkv_split_kernel mha_bwd_dq_split_kernel mha_bwd_split_kernel mha_fwd mha_bwd; do
    python3 tools/pmc_summary.py $C $k 2>/dev/null | sed "s/^/$k  /" >> $O/k2_in_step_pmc_summary.txt
  done
done
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
cat $O/bench_gmd_step_breakdown_f32s.txt | head -12; cat $O/k1g_bwd_pmc_summary.txt; cat $O/k2_in_step_pmc_summary.txt | head -40
du -sh $O
