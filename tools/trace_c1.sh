#!/bin/bash
# GPU box: rocprofv3 kernel trace of the default bench configuration (look-ahead on), per-kernel statistics
#   tools/trace_c1.sh [name] [bench args...]
export TMPDIR=/tmp
NAME=${1:-la}; shift
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/prof
mkdir -p "$OUT"
cd "${GRAFT_REPO_ROOT:-$PWD}" || exit 1
rocprofv3 --kernel-trace --stats -d "$OUT/$NAME" -o "$NAME" --output-format csv -- python3 bench.py --steps 5 --warmup 2 \
    --no-cpu-baseline --no-profile-pass --no-e2e --no-resident-batch "$@" > "$OUT/${NAME}_bench.json" 2> "$OUT/$NAME.log" || { tail -5 "$OUT/$NAME.log"; exit 1; }
python3 - "$OUT/$NAME" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    print("%-72s calls %6s avg %9.2f us total %9.3f ms %6s%%" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                 float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
python3 -c "
import json,sys
d=json.loads(open('$OUT/${NAME}_bench.json').read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'])"
