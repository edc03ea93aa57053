( echo "# tools/fuzz_kernels.py 5 30 | fuzz_model.py | fuzz_decode.py | fuzz_full.py at the end of round 5 (small-M tile shapes, fused in-projection attention, projection + LayerNorm launch, DMA-staged streaming kernel, half-width weight-gradient tiles, inline-asm stores of the streaming 1x1 kernels, halo-image 3x3 tile kernel in)"
python tools/fuzz_kernels.py 5 30 2>&1 | tail -12
python tools/fuzz_model.py 2>&1 | tail -6
python tools/fuzz_decode.py 2>&1 | tail -5
python tools/fuzz_full.py 2>&1 | tail -5
python bench.py --no-cpu-baseline --no-decode --no-ragged --no-extra --soak 400 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('soak', d.get('soak'))"
) > gpurun_out/r05_fuzz_summary.txt 2>&1
tail -30 gpurun_out/r05_fuzz_summary.txt
