# A/B of one environment switch on ONE box: bash tools/ab_env.sh VAR "bench.py flags" key [key2]   (alternates VAR=1 / VAR=0 twice; prints d[key] of the bench line)
VAR=$1; FLAGS=$2; KEY=$3
for e in 1 0 1 0; do
  env $VAR=$e python bench.py $FLAGS 2>/dev/null | tail -1 > /tmp/ab_line.json
  python - "$VAR=$e" "$KEY" <<'PY'
import json, sys
d = json.loads(open('/tmp/ab_line.json').read())
v = d
for k in sys.argv[2].split('.'):
    v = v[k]
print(sys.argv[1], 'ms_per_step %.3f' % d['ms_per_step'], sys.argv[2], json.dumps(v)[:400])
PY
done
