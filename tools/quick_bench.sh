#!/bin/bash
# GPU box: a parity spot check and the kernel breakdown of C1 (and optionally C2) without the CPU legs
#   tools/quick_bench.sh [c2]
set -o pipefail
python tools/exact_debug.py 300 2300 3200 || exit 1
python bench.py --no-cpu-baseline --no-distribute --no-e2e --no-resident-batch > gpurun_out/qb_c1.json 2> gpurun_out/qb_c1.err || { tail -5 gpurun_out/qb_c1.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/qb_c1.json").read().strip().splitlines()[-1])
print("C1 ms_per_step", round(d["ms_per_step"], 3), "residual", d["residual_inf"], "roof", round(d["roofline"]["frac"], 3))
for k, v in d["kernel_breakdown"].items():
    print("  %-18s %8.3f ms  %5.0f launches  %8.2f us" % (k, v["ms_per_step"], v["launches_per_step"], v["avg_us"]))
PY
if [ "$1" = "c2" ]; then
python bench.py --n 2048 --batch 64 --no-cpu-baseline --no-distribute --no-e2e > gpurun_out/qb_c2.json 2> gpurun_out/qb_c2.err || { tail -5 gpurun_out/qb_c2.err; exit 1; }
python - <<'PY'
import json
d = json.loads(open("gpurun_out/qb_c2.json").read().strip().splitlines()[-1])
print("C2 ms_per_step", round(d["ms_per_step"], 3), "residual", d["residual_inf"], "roof", round(d["roofline"]["frac"], 3))
for k, v in d["kernel_breakdown"].items():
    print("  %-18s %8.3f ms  %5.0f launches  %8.2f us" % (k, v["ms_per_step"], v["launches_per_step"], v["avg_us"]))
PY
fi
