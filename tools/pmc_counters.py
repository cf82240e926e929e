"""Per-kernel averages of arbitrary rocprofv3 --pmc counters: tools/pmc_counters.py <dir> [kernel-substring]."""
import collections
import csv
import glob
import sys

path = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-40:]
        if want in k:
            agg[(k, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k, g), cs in sorted(agg.items()):
    n = max(len(v) for v in cs.values())
    print("%-42s grid %-9s dispatches %d" % (k, g, n))
    for c, v in sorted(cs.items()):
        print("    %-34s avg %16.1f   min %16.1f   max %16.1f" % (c, sum(v) / len(v), min(v), max(v)))
