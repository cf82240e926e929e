# usage (GPU box): bash tools/gpu_bench_env_ab.sh "VAR=val" ...  -- headline + detect_256 of bench.py with the default environment and with each setting, alternating, 3 rounds
cd $GRAFT_REPO_ROOT
F="--steps 20 --warmup 5 --no-cpu-baseline --no-allpairs --no-dropin --no-latency --no-f32-loop"
P='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], d["value"], "pairs/s; detect_256", (d.get("summary") or {}).get("detect_256_frames_per_s"))'
for i in 1 2 3; do
  timeout -k 10 300 python bench.py $F 2>/dev/null | python -c "$P" default || exit 1
  for v in "$@"; do
    env $v timeout -k 10 300 python bench.py $F 2>/dev/null | python -c "$P" "$v" || exit 1
  done
done
