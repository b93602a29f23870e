cd /root/repo
for i in 1 2; do SOAK_VARIANT=swin timeout 600 python tools/gpu/dbg_soak.py 1 500 2>&1 | grep "lane_probe=" | cut -c1-330; done
for i in 1 2; do SOAK_BATCH=1 timeout 600 python tools/gpu/dbg_soak.py 1 2500 2>&1 | grep "lane_probe=" | cut -c1-330; done
for i in 1 2; do timeout 600 python tools/gpu/dbg_train_soak.py res bf16 250 2>&1 | grep "^train"; done
timeout 600 python tools/gpu/dbg_train_soak.py swin bf16 120 2>&1 | grep "^train"
timeout 600 python tools/gpu/dbg_train_soak.py res f16x3 150 2>&1 | grep "^train"
