#!/usr/bin/env python3
"""Every stored counter constant bench.py reads, re-measured on the launch shapes bench.py runs (64-frame detect call, 16-pair
match call), in one go on the GPU box:

    python3 tools/pmc_collect.py <tag> [only ...]        # e.g. r06_a ; only = subset of the job names below

Per job: rocprofv3 --pmc <counter> (one counter set per pass, --pmc alone: no trace domains) with the program itself behind
`--` (python3 tools/<probe>.py; this script is only the parent and never touches the GPU), the per-dispatch CSVs reduced to
  gpurun_out/<tag>_pmc_<job>.txt      the human-readable summary (copy into profiles/)
  gpurun_out/<tag>_pmc_traffic.json   the entries of profiles/pmc_traffic.json, each with launch_shape, round tag and the
                                      summary file it came from
HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: the gfx950 correction of MI355X_MICROARCH.md's HBM section (FETCH_SIZE counts
64 B per 128-B request on wide streaming reads; WRITE_SIZE is exact for 16-B-per-lane stores)."""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
B, MB = 64, 16                                  # bench.py's defaults: frames per detect call, pairs per match call
SEQ = 6                                         # scale_space_batch calls tools/kpyr_all.py makes with reps = 4 (2 warm + 4)

# job -> (probe argv, env, counter sets, kernel-name substring, reducer)
JOBS = collections.OrderedDict([
    ("pyramid_all", (["tools/kpyr_all.py", str(B), "4", "all"], {}, ["FETCH_SIZE", "WRITE_SIZE"], "conv_pk", "chain")),
    ("pyramid_frame_driver", (["tools/kpyr_all.py", str(B), "4", "nodog"], {}, ["FETCH_SIZE", "WRITE_SIZE"], "conv_pk", "chain")),
    ("pyramid_levels_dog_only", (["tools/kpyr_all.py", str(B), "4", "dogonly"], {}, ["FETCH_SIZE", "WRITE_SIZE"], "conv_pk", "chain")),
    ("match_coarse_kernel", (["tools/kcoarse16.py"], {"PAIRS": str(MB)}, ["FETCH_SIZE", "WRITE_SIZE"], "match_coarse_kernel", "launch")),
    ("match_top2_kernel_f32", (["tools/kmatch_sustained.py", "16"], {"NM_MATCH_SCREEN": "f32"}, ["FETCH_SIZE", "WRITE_SIZE"], "match_top2_kernel", "launch")),
    ("match_top2_kernel_bf16x3", (["tools/kmatch_sustained.py", "16"], {"NM_MATCH_SCREEN": "bf16x3"}, ["FETCH_SIZE", "WRITE_SIZE"], "match_top2_kernel", "launch")),
    ("match_top2_group_kernel_f32", (["tools/kcoarse16.py"], {"PAIRS": str(MB), "NM_MATCH_SCREEN": "f32"}, ["FETCH_SIZE", "WRITE_SIZE"], "match_top2_group_kernel", "launch")),
    ("distance_mfma_kernel", (["tools/kdist.py"], {}, ["FETCH_SIZE", "WRITE_SIZE"], "distance_mfma_kernel", "launch")),
    ("frame_desc_kernel", (["tools/ksite.py", "describe", str(B)], {}, ["SQ_INSTS_VALU SQ_INSTS_LDS", "SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"], "frame_desc_kernel", "counters")),
    ("frame_orient_kernel", (["tools/ksite.py", "orient", str(B)], {}, ["SQ_INSTS_VALU SQ_INSTS_LDS"], "frame_orient_kernel", "counters")),
    ("detect_stage_kernel", (["tools/ksite.py", "detect", str(B)], {}, ["FETCH_SIZE", "WRITE_SIZE"], "detect_stage_kernel", "launch_max")),
])
SHAPES = {"pyramid_all": [1920, 1080, B], "pyramid_frame_driver": [1920, 1080, B], "pyramid_levels_dog_only": [1920, 1080, B],
          "match_coarse_kernel": [12223, 12080, 128, MB], "match_top2_kernel_f32": [12223, 12080, 128, 1],
          "match_top2_kernel_bf16x3": [12223, 12080, 128, 1], "match_top2_group_kernel_f32": [12223, 12080, 128, 8],
          "distance_mfma_kernel": [12223, 12080, 128, 1],
          "frame_desc_kernel": [1920, 1080, B], "frame_orient_kernel": [1920, 1080, B], "detect_stage_kernel": [1920, 1080, B]}


def rows(path):
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            yield r


def run_pass(tag, job, i, counters, argv, env):
    d = os.path.join(OUT, "%s_pmcraw_%s_%d" % (tag, job, i))
    shutil.rmtree(d, ignore_errors=True)
    e = dict(os.environ)
    e.update(env)
    e["TMPDIR"] = "/tmp"
    cmd = ["timeout", "-k", "10", "400", "rocprofv3", "--pmc"] + counters.split() + ["-d", d, "--output-format", "csv", "--", "python3"] + argv
    t0 = time.time()
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True)
    print("[pmc_collect] %s pass %d (%s): rc %d, %.0f s; %s" % (job, i, counters, r.returncode, time.time() - t0,
                                                                 (r.stdout.strip().splitlines() or [""])[-1][:200]), flush=True)
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-2000:])
    return d, r.returncode, (r.stdout.strip().splitlines() or [""])[-1]


def main():
    tag = sys.argv[1]
    only = sys.argv[2:]
    os.makedirs(OUT, exist_ok=True)
    result = {}
    for job, (argv, env, sets, sub, how) in JOBS.items():
        if only and job not in only:
            continue
        per = collections.defaultdict(lambda: collections.defaultdict(list))     # (kernel, grid) -> counter -> values
        last, failed = "", False
        dirs = []
        for i, cs in enumerate(sets):
            d, rc, last = run_pass(tag, job, i, cs, argv, env)
            dirs.append(d)
            if rc != 0:                                   # a probe that failed or was killed: stop here, start no further GPU step
                failed = True
                break
            for r in rows(d):
                if sub in r["Kernel_Name"]:
                    k = (r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-60:], r["Grid_Size"])
                    per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for d in dirs:
            shutil.rmtree(d, ignore_errors=True)
        if failed:
            print("[pmc_collect] %s FAILED; stopping" % job)
            break
        txt = os.path.join(OUT, "%s_pmc_%s.txt" % (tag, job))
        lines = ["# %s: rocprofv3 --pmc <set> -- python3 %s   (env %s), %s" % (job, " ".join(argv), env, time.strftime("%Y-%m-%d")),
                 "# probe said: " + last]
        entry = {"launch_shape": SHAPES[job], "round": tag, "profile": "profiles/%s_pmc_%s.txt" % (tag, job)}
        tot = collections.defaultdict(float)
        ndisp = 0
        for (k, g), cs in sorted(per.items()):
            n = max(len(v) for v in cs.values())
            ndisp += n
            lines.append("%-62s grid %-9s dispatches %d" % (k, g, n))
            for c, v in sorted(cs.items()):
                lines.append("    %-28s avg %16.1f  min %16.1f  max %16.1f  sum %18.1f" % (c, sum(v) / len(v), min(v), max(v), sum(v)))
                tot[c] += sum(v)
        if how == "chain":
            hbm = (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024 / SEQ
            entry.update({"hbm_bytes_per_call": round(hbm), "hbm_bytes_per_frame": round(hbm / B),
                          "read_bytes_per_frame": round(2 * tot["FETCH_SIZE"] * 1024 / SEQ / B),
                          "written_bytes_per_frame": round(tot["WRITE_SIZE"] * 1024 / SEQ / B), "dispatches": ndisp, "calls": SEQ})
        elif how in ("launch", "launch_max"):
            # per launch of the kernel: the average over its dispatches ("launch_max": of the LARGEST grid, i.e. octave 0)
            keys = list(per.keys())
            if how == "launch_max" and keys:
                gmax = max(int(g) for _, g in keys)
                keys = [k for k in keys if int(k[1]) == gmax]
            f = sum(sum(per[k]["FETCH_SIZE"]) for k in keys)
            w = sum(sum(per[k]["WRITE_SIZE"]) for k in keys)
            n = sum(len(per[k]["FETCH_SIZE"]) for k in keys) or 1
            entry.update({"hbm_bytes_per_launch": round((2 * f + w) * 1024 / n), "read_bytes_per_launch": round(2 * f * 1024 / n),
                          "written_bytes_per_launch": round(w * 1024 / n), "dispatches": n})
        else:
            keys = list(per.keys())
            n = sum(len(per[k].get("SQ_INSTS_VALU", [])) for k in keys) or 1
            for c in sorted(tot):
                entry[c.lower() + "_per_launch"] = round(tot[c] / n)
            entry["dispatches"] = n
            kp = [w for w in last.replace(",", " ").split() if w.isdigit()]
            if "keypoints" in last:
                try:
                    entry["keypoints_per_launch"] = int(last.split("keypoints")[0].replace(",", " ").split()[-1])
                except (ValueError, IndexError):
                    pass
        lines.append("# reduced: " + json.dumps(entry))
        open(txt, "w").write("\n".join(lines) + "\n")
        result[job] = entry
        print("[pmc_collect] %s -> %s" % (job, json.dumps(entry)), flush=True)
        json.dump(result, open(os.path.join(OUT, "%s_pmc_traffic.json" % tag), "w"), indent=1)


if __name__ == "__main__":
    main()
