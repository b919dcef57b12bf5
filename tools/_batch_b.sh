#!/bin/bash
python -m pytest tests/test_train_main.py tests/test_rccl_world1.py tests/test_hip_exchange.py -x -q 2>&1 | tail -3
for mb in 0 4 2; do echo "== SMK_LINEAR_MB=$mb"; SMK_LINEAR_MB=$mb python3 tools/linear_probe.py 4096 2>&1 | grep "^M=" | head -3; SMK_LINEAR_MB=$mb SMK_PROBE_BATCHES=4 python3 tools/inference_probe.py 2>&1 | grep eager; done
