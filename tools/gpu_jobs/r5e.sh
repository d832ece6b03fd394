#!/bin/bash
# round 5, session 5: own bf16 GEMM tuning (chunk depth, scheduling recipe, ablations), poisoned LSTM exchange test, config-5 norms
O=gpurun_out/r5e; mkdir -p $O
(timeout 1500 python -m pytest tests/test_gemm_bf16_gpu.py tests/test_lstm_soak_gpu.py tests/test_config5_bf16_gpu.py -q -m gpu -s 2>&1 | grep -v "^$" | tail -30) > $O/pytest.txt
cat $O/pytest.txt
run() { echo "== $1" >> $O/gemm_bf16.txt; shift; env "$@" python tools/gemm_bf16_time.py 2>&1 | grep -v amdgpu | head -5 >> $O/gemm_bf16.txt; }
run "KC=64 (default), sched_group_barrier recipe" X=1
run "KC=32" TSG_BGEMM_KC=32
run "KC=64, no recipe" TSG_HIP_LIB=tools/_ablate/bg_nosgb.so
run "KC=32, no recipe" TSG_HIP_LIB=tools/_ablate/bg_nosgb.so TSG_BGEMM_KC=32
run "ablation: no DMA" TSG_HIP_LIB=tools/_ablate/bg_abl1.so
run "ablation: no MFMA (fragment reads kept)" TSG_HIP_LIB=tools/_ablate/bg_abl2.so
run "ablation: no fragment reads, no MFMA (DMA + barriers + epilogue)" TSG_HIP_LIB=tools/_ablate/bg_abl4.so
run "ablation: nothing but barriers + epilogue" TSG_HIP_LIB=tools/_ablate/bg_abl5.so
cat $O/gemm_bf16.txt
