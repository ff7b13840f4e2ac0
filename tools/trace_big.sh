#!/bin/bash
# GPU box: rocprofv3 kernel trace of one large single-matrix inversion (look-ahead on) and the timeline of a few blocks
#   tools/trace_big.sh [n] [first_launch] [count]
export TMPDIR=/tmp
N=${1:-16384}
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/prof; mkdir -p "$OUT"
cd "${GRAFT_REPO_ROOT:-$PWD}" || exit 1
rocprofv3 --kernel-trace -d "$OUT/big$N" -o "big$N" --output-format csv -- python3 bench.py --n $N --steps 1 --warmup 1 \
    --no-cpu-baseline --no-profile-pass --no-e2e --no-resident-batch > "$OUT/big$N.json" 2> "$OUT/big$N.log" || { tail -5 "$OUT/big$N.log"; exit 1; }
python3 tools/trace_timeline.py "$OUT/big$N" ${2:-600} ${3:-90}
