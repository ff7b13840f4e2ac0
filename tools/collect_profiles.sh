#!/bin/bash
# Diagnostic: collect the rocprofv3 evidence bench.py's numbers are checked against (run on the GPU box).
#   kernel-trace --stats summaries of the default bench command (C1), of C2 and of the sweep path;
#   FETCH_SIZE / WRITE_SIZE in passes of their own per configuration (MI355X_MICROARCH.md, HBM section);
#   one SQ pass (MFMA busy, wave cycles, waits) for C1 and C2.
# Output: gpurun_out/prof/ (scratch; tools/pmc_traffic.py + a copy of the stats CSVs go into profiles/roundN/).
set -u
export TMPDIR=/tmp
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/prof
mkdir -p "$OUT"
cd "${GRAFT_REPO_ROOT:-$PWD}"
run_stats() {  # name, bench args...
    local name=$1; shift
    rocprofv3 --kernel-trace --stats -d "$OUT/$name" -o "$name" --output-format csv -- python3 bench.py "$@" \
        > "$OUT/${name}_bench_under_rocprof.json" 2> "$OUT/${name}.log" || echo "$name failed"
}
run_stats c1_n4096 --steps 5 --warmup 2 --no-cpu-baseline
run_stats c2_64x2048 --n 2048 --batch 64 --steps 3 --warmup 1 --no-cpu-baseline
run_stats c1_n4096_sweep --algo sweep --steps 2 --warmup 1 --no-cpu-baseline
echo "stats done"
# PMC passes: look-ahead off so that every rank-bw update is ONE launch of the kernel the roofline is quoted for
export MI32_LOOKAHEAD=0
pmc() {  # counter-list name bench-args...
    local ctrs=$1 name=$2; shift 2
    local tag=${ctrs%% *}
    rocprofv3 --pmc $ctrs --kernel-trace -d "$OUT/pmc_${tag}_${name}" -o "pmc_${tag}_${name}" --output-format csv -- \
        python3 bench.py "$@" --no-cpu-baseline --no-profile-pass > "$OUT/pmc_${tag}_${name}.json" 2> "$OUT/pmc_${tag}_${name}.log" \
        || echo "pmc $tag $name failed"
    echo "pmc $tag $name done"
}
for ctr in FETCH_SIZE WRITE_SIZE; do
    pmc $ctr c1 --steps 2 --warmup 1
    pmc $ctr c2 --n 2048 --batch 64 --steps 2 --warmup 1
    pmc $ctr sweep --algo sweep --steps 1 --warmup 1
    pmc $ctr c4 --n 16384 --steps 1 --warmup 1
done
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
pmc "$SQ" c1 --steps 2 --warmup 1
pmc "$SQ" c2 --n 2048 --batch 64 --steps 2 --warmup 1
unset MI32_LOOKAHEAD
# plain bench lines (what the driver runs)
python3 bench.py > "$OUT/bench_c1_default.json" 2> "$OUT/bench_c1_default.log"; echo "bench c1 done"
python3 bench.py --n 2048 --batch 64 --steps 5 --warmup 2 > "$OUT/bench_c2.json" 2> "$OUT/bench_c2.log"; echo "bench c2 done"
python3 bench.py --n 16384 --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/bench_c4.json" 2> "$OUT/bench_c4.log"; echo "bench c4 done"
python3 bench.py --algo sweep --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/bench_c1_sweep.json" 2> "$OUT/bench_c1_sweep.log"; echo "bench sweep done"
python3 tools/pmc_traffic.py "$OUT" "$OUT/pmc_traffic.json" > "$OUT/pmc_traffic.log" 2>&1
python3 tools/pmc_sq_summary.py "$OUT" "$OUT/pmc_sq_summary.json" > "$OUT/pmc_sq_summary.log" 2>&1
ls "$OUT" | head -80
