#!/bin/bash
# Diagnostic: collect the rocprofv3 evidence bench.py's numbers are checked against (run on the GPU box).
#   kernel-trace --stats summaries of the default bench command (C1), of C2 and of the sweep path;
#   FETCH_SIZE / WRITE_SIZE in passes of their own per configuration (MI355X_MICROARCH.md, HBM section);
#   one SQ pass (MFMA busy, wave cycles, waits) for C1 and C2.
# Output: gpurun_out/prof/ (scratch; tools/pmc_traffic.py + a copy of the stats CSVs go into profiles/roundN/).
# Every profiled command runs bench.py's TIMED LOOP ONLY (--no-e2e --no-profile-pass --no-cpu-baseline): round 2's
# PMC passes of the sweep path died with SIGSEGV inside the profiler's tool library during bench.py's e2e leg
# (six host-pointer calls = 24 k more dispatches behind the 8 k timed ones); the counters are only wanted for the
# dominant kernels anyway (--kernel-include-regex).  A failing step is reported with its log and makes the script
# exit non-zero; the remaining steps still run (one call of gpurun per collection).
#   tools/collect_profiles.sh [stats|pmc|sq|bench|all]...
set -u
export TMPDIR=/tmp
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/prof
mkdir -p "$OUT"
cd "${GRAFT_REPO_ROOT:-$PWD}" || exit 1
WHAT=${*:-all}
FAILED=()
want() { [[ " $WHAT " == *" all "* || " $WHAT " == *" $1 "* ]]; }
fail() {  # name, log
    FAILED+=("$1")
    echo "FAILED: $1 -- last lines of $2:"
    tail -15 "$2"
}
run_stats() {  # name, bench args...
    local name=$1; shift
    rocprofv3 --kernel-trace --stats -d "$OUT/$name" -o "$name" --output-format csv -- python3 bench.py "$@" --no-resident-batch \
        --no-cpu-baseline --no-e2e > "$OUT/${name}_bench_under_rocprof.json" 2> "$OUT/${name}.log" || fail "stats $name" "$OUT/${name}.log"
    echo "stats $name done"
}
pmc() {  # counter-list name kernel-regex bench-args...
    local ctrs=$1 name=$2 regex=$3; shift 3
    local tag=${ctrs%% *}
    rocprofv3 --pmc $ctrs --kernel-trace --kernel-include-regex "$regex" -d "$OUT/pmc_${tag}_${name}" -o "pmc_${tag}_${name}" \
        --output-format csv -- python3 bench.py "$@" --no-cpu-baseline --no-profile-pass --no-e2e --no-resident-batch \
        > "$OUT/pmc_${tag}_${name}.json" 2> "$OUT/pmc_${tag}_${name}.log" || fail "pmc $tag $name" "$OUT/pmc_${tag}_${name}.log"
    echo "pmc $tag $name done"
}
if want stats; then
    run_stats c1_n4096 --steps 5 --warmup 2
    run_stats c2_64x2048 --n 2048 --batch 64 --steps 3 --warmup 1
    run_stats c1_n4096_sweep --algo sweep --steps 2 --warmup 1
fi
# PMC passes: look-ahead off so that every rank-bw update is ONE launch of the kernel the roofline is quoted for
export MI32_LOOKAHEAD=0
if want pmc; then
    for ctr in FETCH_SIZE WRITE_SIZE; do
        pmc $ctr c1 "gj_rank_bw2" --steps 2 --warmup 1
        pmc $ctr c2 "gj_rank_bw2" --n 2048 --batch 64 --steps 2 --warmup 1
        pmc $ctr sweep "gj_sweep_step" --algo sweep --steps 1 --warmup 1
        pmc $ctr c4 "gj_rank_bw2" --n 16384 --steps 1 --warmup 1
    done
fi
if want sq; then
    SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"
    pmc "$SQ" c1 "gj_rank_bw2|gj_subpanel|gj_inblock|gj_block_strip" --steps 2 --warmup 1
    pmc "$SQ" c2 "gj_rank_bw2|gj_subpanel|gj_inblock|gj_block_strip" --n 2048 --batch 64 --steps 2 --warmup 1
fi
unset MI32_LOOKAHEAD
if want bench; then  # plain bench lines (what the driver runs)
    python3 bench.py > "$OUT/bench_c1_default.json" 2> "$OUT/bench_c1_default.log" || fail "bench c1" "$OUT/bench_c1_default.log"; echo "bench c1 done"
    python3 bench.py --n 2048 --batch 64 --steps 5 --warmup 2 > "$OUT/bench_c2.json" 2> "$OUT/bench_c2.log" || fail "bench c2" "$OUT/bench_c2.log"; echo "bench c2 done"
    python3 bench.py --n 16384 --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/bench_c4.json" 2> "$OUT/bench_c4.log" || fail "bench c4" "$OUT/bench_c4.log"; echo "bench c4 done"
    python3 bench.py --algo sweep --steps 2 --warmup 1 --no-cpu-baseline > "$OUT/bench_c1_sweep.json" 2> "$OUT/bench_c1_sweep.log" || fail "bench sweep" "$OUT/bench_c1_sweep.log"; echo "bench sweep done"
fi
python3 tools/pmc_traffic.py "$OUT" "$OUT/pmc_traffic.json" > "$OUT/pmc_traffic.log" 2>&1 || fail "pmc_traffic.py" "$OUT/pmc_traffic.log"
python3 tools/pmc_sq_summary.py "$OUT" "$OUT/pmc_sq_summary.json" > "$OUT/pmc_sq_summary.log" 2>&1 || fail "pmc_sq_summary.py" "$OUT/pmc_sq_summary.log"
ls "$OUT" | head -80
if [ ${#FAILED[@]} -ne 0 ]; then
    echo "collect_profiles: ${#FAILED[@]} step(s) FAILED: ${FAILED[*]}"
    exit 1
fi
echo "collect_profiles: all steps done"
