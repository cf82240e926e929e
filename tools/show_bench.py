"""Pretty-print the JSON line of bench.py: tools/show_bench.py <file>."""
import json
import sys

d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print("value %.1f %s  ms/step %.3f  n_gpus %d  verified %s" % (d["value"], d["unit"], d["ms_per_step"], d["n_gpus"], d.get("verified_pair0_vs_oracle")))
r = d["roofline"]
print("%s: %.1f TF / %.0f peak = frac %.4f avg %.1f us (%d launches) traffic %s %s" % (r["kernel"], r["achieved"], r["peak"], r["frac"], 1e3 * r["avg_ms"], r["launches_timed"], r.get("traffic"),
      {k: r[k] for k in ("executed_TFLOPs", "frac_executed", "vs_f32_mfma_peak") if k in r}))
if "roofline_f32_screen" in d:
    print("f32 screen:", json.dumps(d["roofline_f32_screen"]))
p = d["roofline_pyramid"]
print("pyramid all: %s GB/s alg frac %s (with grad %s) %s us/frame physical %s GB/s (%s)" % (p.get("achieved"), p.get("frac"), p.get("frac_with_gradients"), p.get("us_per_frame"), p.get("physical_GBps"), p.get("physical_frac")))
if "octave0" in p:
    print("  octave0:", p["octave0"])
for k in ("dropin_api", "detect_256", "allpairs_100k", "cpu_baseline"):
    if k in d:
        print(k, json.dumps(d[k]))
