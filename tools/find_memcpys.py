import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from torch.profiler import profile, ProfilerActivity
from smokephysai_amd.models import SmokePhysNet
torch.manual_seed(0)
m = SmokePhysNet().cuda().eval()
x = torch.rand(int(os.environ.get("SMK_PROBE_BATCH", "1")), 1, 256, 256, device="cuda")
with torch.no_grad():
    for _ in range(3): m(x)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
        m(x)
        torch.cuda.synchronize()
evs = prof.events()
# find cpu ops whose children include a memcpy
names = {}
for e in evs:
    if "Memcpy" in e.name or "copyBuffer" in e.name or "memcpy" in e.name.lower():
        p = e.cpu_parent
        chain = []
        while p is not None and len(chain) < 4:
            chain.append(p.name); p = p.cpu_parent
        names[" <- ".join(chain)] = names.get(" <- ".join(chain), 0) + 1
for k, v in sorted(names.items(), key=lambda kv: -kv[1]): print(v, k)
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=12))
