#!/bin/bash
# GPU job of round 5: the whole GPU suite, smoke(), the default bench line (what the driver runs), the bf16-storage line
O=gpurun_out/r5full; mkdir -p $O
(timeout 3000 python -m pytest tests -q -m gpu -s 2>&1 | grep -v "^$" | grep "passed\|failed\|FAILED\|Error\|relative L2 errors" | cut -c1-3000 | tail -40) > $O/pytest_gpu_full.txt
(python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2) > $O/smoke.txt
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --dtype bf16 --no-alt --cpu-sample 0 > $O/bench_bf16.json 2> $O/bench_bf16.err
cat $O/pytest_gpu_full.txt $O/smoke.txt; cut -c1-400 $O/bench_default.json; cut -c1-300 $O/bench_bf16.json; tail -5 $O/bench_default.err
