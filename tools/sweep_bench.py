"""Headline-only bench.py runs under different environment settings (tuning hooks of the library), one child process per
setting: prints frame-pairs/s and ms/step of each.   python tools/sweep_bench.py "A=1 B=2" "A=3" ... [-- extra bench args]"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
extra = []
if "--" in args:
    k = args.index("--")
    args, extra = args[:k], args[k + 1:]
base = [sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-allpairs", "--no-detect256", "--no-dropin",
        "--no-latency", "--no-f32-loop", "--steps", "10", "--warmup", "3"] + extra
for setting in args or [""]:
    env = dict(os.environ)
    for kv in setting.split():
        k, v = kv.split("=", 1)
        env[k] = v
    r = subprocess.run(base, env=env, capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        print("%-50s FAILED rc=%d %s" % (setting, r.returncode, r.stderr[-300:]), flush=True)
        continue
    j = json.loads(line[-1])
    print("%-50s %8.1f frame-pairs/s  %7.3f ms/step  verified=%s" % (setting or "(default)", j["value"], j["ms_per_step"],
          j.get("summary", {}).get("verified_pair0_vs_oracle")), flush=True)
