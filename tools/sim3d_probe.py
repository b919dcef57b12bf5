"""configs[4] stepper: ms per step at 512 x 512 x 64 x 8 (under rocprofv3 --kernel-trace --stats for the per-kernel split)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
print(json.dumps(bench.config4_leg(torch.device("cuda", 0), steps=int(sys.argv[1]) if len(sys.argv) > 1 else 6)))
