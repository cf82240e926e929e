# usage (GPU box): bash tools/gpu_flags.sh  -- headline-only bench under match-stream configurations, alternating
cd $GRAFT_REPO_ROOT
F="--steps 20 --warmup 5 --no-cpu-baseline --no-allpairs --no-detect256 --no-dropin --no-latency --no-f32-loop"
P='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], d["value"], d["roofline"]["avg_ms"], d["roofline"]["frac"], d["summary"].get("verified_pair0_vs_oracle"))'
for i in 1 2 3; do
timeout -k 10 200 python bench.py $F 2>/dev/null | python -c "$P" default || exit 1
timeout -k 10 200 python bench.py $F --match-streams 2 2>/dev/null | python -c "$P" match-streams-2 || exit 1
timeout -k 10 200 python bench.py $F --match-pipeline 2>/dev/null | python -c "$P" match-pipeline || exit 1
done
