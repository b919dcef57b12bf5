#!/bin/bash
# Shader clock while the headline step runs: rocm-smi is polled in the background during a long bench leg (the encoder is 84 % of the step).
# gpurun -- bash tools/clock_probe.sh <tag>
set -u
TAG=${1:-clk}; R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
python3 $R/bench.py --steps 12000 --warmup 50 --cpu-frames 0 --no-inference --no-alt --no-config1 --no-dataset --no-config4 --no-train-step > $OUT/bench.json 2> $OUT/bench.err &
BP=$!
for i in $(seq 1 400); do      # poll from the start; the loaded samples are the ones with sclk far above idle
  /opt/rocm/bin/rocm-smi --showclocks --showpower --json 2>/dev/null | tr -d '\n' >> $OUT/smi.jsonl; echo >> $OUT/smi.jsonl
  sleep 0.15
  kill -0 $BP 2>/dev/null || break
done
wait $BP
python3 - <<PY
import json, re
rows = [json.loads(l) for l in open("$OUT/smi.jsonl") if l.strip().startswith("{")]
sclk, pw = [], []
for r in rows:
    c = r.get("card0", {})
    for k, v in c.items():
        if "sclk" in k.lower():
            m = re.search(r"(\d+)\s*Mhz", str(v), re.I)
            if m: sclk.append(int(m.group(1)))
        if "power" in k.lower() and "W" in k:
            try: pw.append(float(v))
            except Exception: pass
b = json.loads(open("$OUT/bench.json").read().strip().splitlines()[-1])
busy = [(c, p) for c, p in zip(sclk, pw) if p > 600]
print(json.dumps({"samples": len(sclk), "loaded_samples": len(busy), "sclk_MHz_loaded": [c for c, _ in busy], "power_W_loaded": [p for _, p in busy], "sclk_MHz": sclk, "power_W": pw, "ms_encode_per_step": b["ms_encode_per_step"], "ms_sim_per_step": b["ms_sim_per_step"], "value": b["value"]}))
PY
