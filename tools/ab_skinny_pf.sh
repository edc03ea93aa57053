# small-M kernel: tile choice A/B (tuning build): GPV_SKINNY_TILE=1 = 64 x 64 everywhere (round 4), 0 = by cost
export GPV_TUNING_LIB=1
for t in 1 0 2 3 1 0; do echo TILE=$t; GPV_SKINNY_TILE=$t python tools/bench_skinny_pf.py; done
for t in 1 0 1 0; do echo TILE=$t; GPV_SKINNY_TILE=$t python bench.py --no-cpu-baseline --no-ragged --no-extra 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('ms_per_step %.3f' % d['ms_per_step'], 'greedy bs1 %.3f bs64 %.3f' % (d['greedy_decode']['bs1']['ms_per_image'], d['greedy_decode']['bs64']['ms_per_batch']), d['roofline']['timed_region_brackets_ms'])"; done
