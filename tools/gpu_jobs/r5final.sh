#!/bin/bash
# GPU job of round 5, final tree: whole GPU suite, smoke(), default bench line, bf16-storage line, then the profiling session
bash tools/gpu_jobs/r5full.sh
bash tools/profile_r5.sh
