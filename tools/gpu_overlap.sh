# usage (GPU box): bash tools/gpu_overlap.sh  -- headline-only bench under stream / batch / grid-cap combinations (r06_d)
cd $GRAFT_REPO_ROOT
F="--steps 20 --warmup 5 --no-cpu-baseline --no-allpairs --no-detect256 --no-dropin --no-latency --no-f32-loop"
P='import sys,json; d=json.loads([l for l in sys.stdin if l.startswith("{")][-1]); print(sys.argv[1], d["value"], d["summary"].get("verified_pair0_vs_oracle"))'
run() { name="$1"; shift; env "$@" timeout -k 10 200 python bench.py $F $EXTRA 2>/dev/null | python -c "$P" "$name" || exit 1; }
for i in 1 2; do
EXTRA="" run "default" A=1
EXTRA="--streams 2 --batch 32" run "s2b32" A=1
EXTRA="--streams 2 --batch 32" run "s2b32_desc2048" NM_DESC_BLOCKS=2048
EXTRA="--streams 2 --batch 32" run "s2b32_desc3072_or256" NM_DESC_BLOCKS=3072 NM_ORIENT_BLOCKS=256
EXTRA="--streams 4 --batch 16" run "s4b16_desc2048" NM_DESC_BLOCKS=2048
EXTRA="" run "default_desc4096" NM_DESC_BLOCKS=4096
EXTRA="" run "default_desc16384" NM_DESC_BLOCKS=16384
EXTRA="--overlap" run "overlap" A=1
done
