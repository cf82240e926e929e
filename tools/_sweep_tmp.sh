for a in "" "--match-pipeline" "--match-streams 2" "--match-pipeline" ""; do echo "== $a"; python bench.py --no-cpu-baseline --no-allpairs --no-detect256 --no-dropin --no-latency --no-f32-loop $a 2>/dev/null | python -c "
import sys, json
j = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(j['value'], j['ms_per_step'], 'roofline frac', j['roofline']['frac'], 'avg_ms', j['roofline'].get('avg_ms'), j.get('summary',{}).get('verified_pair0_vs_oracle'))"; done
