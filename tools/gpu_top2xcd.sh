# usage (GPU box): bash tools/gpu_top2xcd.sh  -- the single-pass screens with one XCD per pair (default) against one launch per pair on the whole chip (NM_TOP2_PAIR_XCD=0)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_match.py tests/test_gpu_bench_config.py -m gpu -x -q 2>&1 | tail -3 || exit 1
F="--steps 10 --warmup 3 --no-cpu-baseline --no-allpairs --no-detect256 --no-dropin --no-latency"
P='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); r=d["roofline_f32_screen"]; print(sys.argv[1], "value", d["value"], "value_f32_screen", d["value_f32_screen"], "f32 frac", r["frac"], "avg_ms", r["avg_ms"], r.get("same_matches_as_default_screen"))'
for i in 1 2 3; do
timeout -k 10 300 python bench.py $F 2>/dev/null | python -c "$P" "pair-per-xcd" || exit 1
NM_TOP2_PAIR_XCD=0 timeout -k 10 300 python bench.py $F 2>/dev/null | python -c "$P" "chip-per-pair" || exit 1
done
