"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB per dispatch).
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 64 B per 128-B request on wide streaming
reads -> doubled here; WRITE_SIZE is exact for 16-B-per-lane stores (4-B stores are uncalibrated)."""
import collections
import csv
import glob
import sys

def load(path, name):
    out = collections.defaultdict(list)
    for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                k = (r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-46:], r["Grid_Size"])
                out[k].append(float(r["Counter_Value"]))
    return out

fe = load(sys.argv[1], "FETCH_SIZE")
wr = load(sys.argv[2], "WRITE_SIZE")
print("%-48s %-10s %6s %14s %14s %16s" % ("kernel", "grid", "calls", "FETCH_KiB(raw)", "WRITE_KiB", "HBM_MB(2F+W)"))
rows = []
for k in fe:
    f = sum(fe[k]) / len(fe[k]); w = sum(wr.get(k, [0])) / max(1, len(wr.get(k, [0])))
    rows.append((2 * f + w, k, len(fe[k]), f, w))
for tot, k, n, f, w in sorted(rows, reverse=True)[: int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
    print("%-48s %-10s %6d %14.1f %14.1f %16.2f" % (k[0], k[1], n, f, w, tot * 1024 / 1e6))
