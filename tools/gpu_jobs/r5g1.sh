#!/bin/bash
O=gpurun_out/r5g1; mkdir -p $O
python tools/glue_sites.py > $O/glue_sites.txt 2>&1; head -70 $O/glue_sites.txt | cut -c1-230
