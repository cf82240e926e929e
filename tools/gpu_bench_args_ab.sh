# usage (GPU box): bash tools/gpu_bench_args_ab.sh "<extra bench.py args A>" "<extra args B>" ...  -- headline-only bench under each argument set, alternating, 2 rounds
cd $GRAFT_REPO_ROOT
F="--no-cpu-baseline --no-allpairs --no-detect256 --no-dropin --no-latency --no-f32-loop"
P='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], "->", d["value"], "pairs/s", d["ms_per_step"], "ms/step", d["config"]["pairs_per_gpu_per_step"], "pairs/step")'
for i in 1 2; do
  for a in "$@"; do
    timeout -k 10 300 python bench.py $F $a 2>/dev/null | python -c "$P" "$a" || exit 1
  done
done
