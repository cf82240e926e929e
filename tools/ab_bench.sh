F="--steps 20 --warmup 5 --no-cpu-baseline --no-allpairs --no-detect256 --no-dropin --no-latency --no-f32-loop"
for i in 1 2 3; do
  timeout -k 10 200 python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('product ', d['value'], d['roofline']['avg_ms'])" || exit 1
  NM_DIAGNOSTIC=1 NM_HIP_LIB=$PWD/tools/_variants/libnm_hip_nocross.so timeout -k 10 200 python bench.py $F 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('nocross ', d['value'], d['roofline']['avg_ms'])" || exit 1
done
