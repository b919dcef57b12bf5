#!/bin/bash
python -m pytest tests/test_hip_linear.py tests/test_hip_transformer.py tests/test_hip_pipeline.py tests/test_hip_exchange.py -x -q 2>&1 | tail -4
python3 tools/inference_probe.py 2>&1 | grep -E "eager|max" 
SMK_BODY_FUSE_LN=0 python3 tools/inference_probe.py 2>&1 | grep -E "eager"
