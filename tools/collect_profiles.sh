#!/bin/bash
# Diagnostic: collect the rocprofv3 evidence bench.py's numbers are checked against (run on the GPU box).
#   kernel-trace --stats summaries of the default bench command (C1), of C2 and of the sweep path;
#   FETCH_SIZE / WRITE_SIZE in passes of their own for the rank-bw kernel (MI355X_MICROARCH.md, HBM section).
# Output: gpurun_out/prof/ (scratch; copy what is to be kept into profiles/).
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
# PMC passes: look-ahead off so that every rank-bw update is ONE launch of the kernel the roofline is quoted for
export MI32_LOOKAHEAD=0
for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --kernel-trace -d "$OUT/pmc_$ctr" -o pmc_$ctr --output-format csv -- python3 bench.py \
        --steps 2 --warmup 1 --no-cpu-baseline --no-profile-pass > "$OUT/pmc_$ctr.json" 2> "$OUT/pmc_$ctr.log" || echo "pmc $ctr failed"
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace \
    -d "$OUT/pmc_sq" -o pmc_sq --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile-pass \
    > "$OUT/pmc_sq.json" 2> "$OUT/pmc_sq.log" || echo "pmc sq failed"
unset MI32_LOOKAHEAD
# plain bench lines (what the driver runs)
python3 bench.py > "$OUT/bench_c1_default.json" 2> "$OUT/bench_c1_default.log"
python3 bench.py --n 2048 --batch 64 --steps 5 --warmup 2 > "$OUT/bench_c2.json" 2> "$OUT/bench_c2.log"
python3 bench.py --n 16384 --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/bench_c4.json" 2> "$OUT/bench_c4.log"
python3 bench.py --algo sweep --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/bench_c1_sweep.json" 2> "$OUT/bench_c1_sweep.log"
ls -R "$OUT" | head -60
