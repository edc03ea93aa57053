# same-box A/B of two builds of the library: gpv-1_amd/csrc/libgpv_hip_old.so (reference point) vs the in-tree build
for i in 1 2; do
for lib in old new; do
  if [ $lib = old ]; then export GPV_HIP_LIB=$PWD/gpv-1_amd/csrc/libgpv_hip_old.so; else unset GPV_HIP_LIB; fi
  echo "== $lib"
  python tools/bench_conv.py 2>&1 | grep "^sum us"
  python tools/bench_dgrad.py 2>&1 | tail -1
  python bench.py --no-cpu-baseline --no-decode 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['fwd_ms'], d['roofline']['bwd_ms'])"
done; done
