#!/bin/bash
# The replay defect, reproduced by tests/test_models_gpu.py::test_graph_replay_gradients_match_the_eager_step, under the probe switches of
# engine.GraphedTrainStep (TSG_GRAPH_PROBE: ab / ba = host synchronisation between the graph launches, skipb = graph A only)
O=gpurun_out/$1; mkdir -p $O
for m in "" ab ba abba skipb; do
  echo "== TSG_GRAPH_PROBE='$m'" >> $O/probe.txt
  TSG_GRAPH_PROBE=$m python -m pytest tests/test_models_gpu.py -q -m gpu -k 'replay_gradients' 2>&1 | grep "deviates\|passed\|failed" | cut -c1-260 >> $O/probe.txt
done
cat $O/probe.txt
