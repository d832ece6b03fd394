cd /root/repo; export TMPDIR=/tmp
O=$PWD/gpurun_out/prof_sa; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python3 bench.py --predictor self_attn --steps 20 --warmup 5 --cpu-sample 0 --no-alt --no-micro > $O/bench.json 2> $O/bench.err
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python3 tools/trace_medians.py $T 40 > $O/medians.txt
python3 tools/step_breakdown.py $T > $O/breakdown.txt 2>&1
find $O -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
cat $O/breakdown.txt | head -12; head -34 $O/medians.txt | cut -c1-175
