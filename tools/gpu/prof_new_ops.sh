cd /root/repo; mkdir -p gpurun_out; export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_msda -o msda --output-format csv -- python -m pytest /root/repo/tests/test_zz_gpu_msda.py -q -m gpu -k "rate" -p no:cacheprovider > /root/repo/gpurun_out/rocprof_msda.log 2>&1); echo "msda rc=$?"
KS=$(find gpurun_out/prof_msda -name "*kernel_stats.csv" | head -1); head -n 8 "$KS" | cut -c1-220; cp "$KS" gpurun_out/r06_call45_kernel_stats_msda.csv
(cd /tmp && timeout 800 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_train_x3 -o train --output-format csv -- python /root/repo/bench.py --mode train-dp --variant res --precision f16x3 --batch 4 --steps 3 --warmup 1 > /root/repo/gpurun_out/rocprof_train_x3.log 2>&1); echo "train rc=$?"
KS=$(find gpurun_out/prof_train_x3 -name "*kernel_stats.csv" | head -1); grep -v "miopen\|MIOpen\|Cijk\|naive\|igemm_\|batched_transpose\|SubTensor\|OpTensor" "$KS" | head -n 24 | cut -c1-200; cp "$KS" gpurun_out/r06_call45_kernel_stats_train_res_f16x3.csv
find gpurun_out/prof_msda gpurun_out/prof_train_x3 -name "*kernel_trace.csv" -delete
