export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
python3 bench.py --n 16384 --steps 2 --warmup 1 --no-cpu-baseline --no-e2e --no-resident-batch > $OUT/bench_c4.json 2> $OUT/bench_c4.log || { tail -5 $OUT/bench_c4.log; exit 1; }
python3 tools/show_bench.py $OUT/bench_c4.json 2>/dev/null | head -12
export MI32_LOOKAHEAD=0
for ctr in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --pmc $ctr --kernel-trace --kernel-include-regex "gj_rank_bw2" -d $OUT/pmc_${ctr}_c4 -o pmc_${ctr}_c4 --output-format csv -- python3 bench.py --n 16384 --steps 1 --warmup 1 --no-cpu-baseline --no-profile-pass --no-e2e --no-resident-batch > $OUT/pmc_${ctr}_c4.json 2> $OUT/pmc_${ctr}_c4.log || { tail -5 $OUT/pmc_${ctr}_c4.log; exit 1; }
done
python3 tools/pmc_traffic.py $OUT $OUT/pmc_traffic.json > /dev/null; python3 -c "
import json; d=json.load(open('$OUT/pmc_traffic.json')); print({k: round(v['traffic_over_algorithmic'],3) for k,v in d.items() if isinstance(v,dict) and 'traffic_over_algorithmic' in v})"
