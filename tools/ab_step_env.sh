#!/bin/bash
# A/B of one tuning-build environment switch inside the training step, alternating on one box:
#   VAR=GPV_WG8H A=0 B=1 bash tools/ab_step_env.sh      -> ms_per_step and the conv body's roofline.frac of each run
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
VAR=${VAR:-GPV_WG8H}; A=${A:-0}; B=${B:-1}
{
for r in 1 2 3; do
  for v in $A $B; do
    echo -n "$VAR=$v: "
    env GPV_TUNING_LIB=1 $VAR=$v timeout 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-decode --no-ragged --no-extra 2>/dev/null | \
      python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.3f ms  frac %.4f' % (d['ms_per_step'], d['roofline']['frac']))"
  done
done
} > gpurun_out/ab_step_env.txt 2>&1
cat gpurun_out/ab_step_env.txt
