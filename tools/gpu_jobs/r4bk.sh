#!/bin/bash
O=gpurun_out/r4bk; rm -rf $O; mkdir -p $O
(timeout 900 python -m pytest tests/test_lstm_gpu.py -x -q -m gpu -k "padded" 2>&1 | tail -30) > $O/pytest.txt; cat $O/pytest.txt
