# usage (GPU box): bash tools/ab_env.sh "VAR=val [VAR2=val]" ...  -- headline-only bench with the default environment and with each setting, alternating, 3 rounds
F="--steps 20 --warmup 5 --no-cpu-baseline --no-allpairs --no-detect256 --no-dropin --no-latency --no-f32-loop"
P='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], d["value"])'
for i in 1 2 3; do
  timeout -k 10 200 python bench.py $F 2>/dev/null | python -c "$P" default || exit 1
  for v in "$@"; do
    env $v timeout -k 10 200 python bench.py $F 2>/dev/null | python -c "$P" "$v" || exit 1
  done
done
