#!/bin/bash
# A/B of the half-width eight-phase weight-gradient tiles (layer2's grouped call), one box: parity test, then the three grouped
# calls of the bench workload alone with WG8H=0 | 1 and a sweep of the work-unit length (tuning build; 0 = chosen per call).
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "wgrad" 2>&1 | tail -3
for r in 1 2; do
  WG8H=0 timeout 300 python tools/bench_wgrad_ablate.py 2>&1 | grep -v amdgpu.ids | tail -2
  WG8H=1 timeout 300 python tools/bench_wgrad_ablate.py 2>&1 | grep -v amdgpu.ids | tail -2
done
for kt in ${KTS:-180 200 220 372 400 480}; do
  GPV_TUNING_LIB=1 GPV_WG8H_KT=$kt WG8H=1 timeout 300 python tools/bench_wgrad_ablate.py 2>&1 | grep -v amdgpu.ids | tail -2
done
} > gpurun_out/ab_wg8h.txt 2>&1
cat gpurun_out/ab_wg8h.txt
