#!/bin/bash
SMK_PROBE_BATCH=4 python3 tools/find_memcpys.py 2>&1 | grep -v "^\-\-\-\|^$" | head -40
for w in 2 1; do for b in 2 4 8 16; do echo "want=$w batch=$b: $(SMK_LINEAR_WANT=$w SMK_PROBE_BATCHES=$b python3 tools/inference_probe.py 2>&1 | grep eager)"; done; done
