# usage (GPU box): bash tools/ab_bench.sh <variant> [<variant> ...]  -- headline-only bench of the product and of tools/_variants/libnm_hip_<variant>.so,
# alternating, three rounds; value (frame-pairs/s) and the coarse launch in situ (ms)
F="--steps 20 --warmup 5 --no-cpu-baseline --no-allpairs --no-detect256 --no-dropin --no-latency --no-f32-loop"
P='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], d["value"], d["roofline"]["avg_ms"], d["summary"].get("verified_pair0_vs_oracle"))'
for i in 1 2 3; do
  timeout -k 10 200 python bench.py $F 2>/dev/null | python -c "$P" product || exit 1
  for v in "$@"; do
    NM_DIAGNOSTIC=1 NM_HIP_LIB=$PWD/tools/_variants/libnm_hip_$v.so timeout -k 10 200 python bench.py $F 2>/dev/null | python -c "$P" $v || exit 1
  done
done
