# usage: tools/fuzz_round.sh <tag> "<what is new in the kernels / host code this round>"   (on the GPU box; writes gpurun_out/<tag>_fuzz_summary.txt)
tag=${1:-r06}
what=${2:-"end of round"}
out=gpurun_out/${tag}_fuzz_summary.txt
mkdir -p gpurun_out
( echo "# tools/fuzz_kernels.py 5 30 | fuzz_model.py | fuzz_decode.py | fuzz_full.py | bench.py --soak 400 | soak_decode.py -- ${tag}: ${what}"
timeout 600 python tools/fuzz_kernels.py 5 30 2>&1 | tail -12
timeout 600 python tools/fuzz_model.py 2>&1 | tail -6
timeout 600 python tools/fuzz_decode.py 2>&1 | tail -5
timeout 600 python tools/fuzz_full.py 2>&1 | tail -5
timeout 600 python bench.py --no-cpu-baseline --no-decode --no-ragged --no-extra --no-traffic --soak 400 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('soak', d.get('soak'))"
timeout 600 python tools/soak_decode.py 2>&1 | tail -3
) > $out 2>&1
tail -40 $out
