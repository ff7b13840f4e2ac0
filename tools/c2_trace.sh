#!/bin/bash
# GPU box: rocprofv3 kernel trace of C2 (64 x 2048^2, split in two halves on two streams) and its timeline by queue
#   tools/c2_trace.sh [name]
export TMPDIR=/tmp
NAME=${1:-c2}
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/prof; mkdir -p "$OUT"
cd "${GRAFT_REPO_ROOT:-$PWD}" || exit 1
rocprofv3 --kernel-trace -d "$OUT/$NAME" -o "$NAME" --output-format csv -- python3 bench.py --n 2048 --batch 64 --steps 2 --warmup 1 \
    --no-cpu-baseline --no-profile-pass --no-e2e --no-distribute > "$OUT/$NAME.json" 2> "$OUT/$NAME.log" || { tail -5 "$OUT/$NAME.log"; exit 1; }
python3 tools/trace_timeline.py "$OUT/$NAME" 0 140
