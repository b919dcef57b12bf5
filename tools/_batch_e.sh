#!/bin/bash
python -m pytest tests/test_hip_linear.py tests/test_hip_transformer.py tests/test_hip_pipeline.py -x -q 2>&1 | tail -2
for ks in 1 0; do echo "KS2=$ks: $(SMK_LINEAR_KS2=$ks python3 tools/linear_probe.py 4096 2>&1 | grep '^M=' | head -3)"; for b in 1 2 4 8; do echo "KS2=$ks batch=$b: $(SMK_LINEAR_KS2=$ks SMK_PROBE_BATCHES=$b python3 tools/inference_probe.py 2>&1 | grep eager)"; done; done
