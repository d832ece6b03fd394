#!/bin/bash
O=gpurun_out/r4bj; rm -rf $O; mkdir -p $O
python tools/dropout_time.py > $O/dropout_time.txt 2>&1; cat $O/dropout_time.txt
