#!/bin/bash
# GPU box: C2 (64 x 2048^2) under a list of environment settings, one bench run each:  tools/c2_env_exp.sh "A=1 B=2" "C=3" ...
for setting in "$@"; do
  env $setting python bench.py --n 2048 --batch 64 --steps 5 --warmup 2 --no-cpu-baseline --no-distribute --no-e2e --no-profile-pass > gpurun_out/c2x.json 2> gpurun_out/c2x.err || { tail -3 gpurun_out/c2x.err; exit 1; }
  python -c "
import json; d=json.loads(open('gpurun_out/c2x.json').read().strip().splitlines()[-1]); print('$setting: ms', round(d['ms_per_step'],3), 'resid', d['residual_inf'])"
done
