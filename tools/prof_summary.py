"""Summarise a rocprofv3 --kernel-trace run: per (kernel, grid) count / avg / min microseconds.
Reads the *kernel_trace.csv files under a directory (--output-format csv) or the rocpd *.db (the default format of ROCm 7.2)."""
import collections
import csv
import glob
import sqlite3
import sys

path = sys.argv[1]
agg = collections.defaultdict(list)


def short(name):
    return name.replace("(anonymous namespace)::", "").split("(")[0][-44:]


for f in glob.glob(path + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        key = (short(r["Kernel_Name"]), "%sx%sx%s" % (r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"]),
               r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"])
        agg[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for f in ([path] if path.endswith(".db") else glob.glob(path + "/**/*.db", recursive=True)):
    cur = sqlite3.connect(f).cursor()
    for name, gx, gy, gz, st, en, lds, vg, ag in cur.execute(
            "select name, grid_x, grid_y, grid_z, start, end, lds_size, vgpr_count, accum_vgpr_count from kernels"):
        agg[(short(name), "%dx%dx%d" % (gx, gy, gz), str(vg), str(ag), str(lds))].append(en - st)
tot = sum(sum(v) for v in agg.values())
print("%-46s %-18s %5s %5s %7s %6s %10s %10s %6s" % ("kernel", "grid", "vgpr", "agpr", "lds", "calls", "avg_us", "min_us", "pct"))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print("%-46s %-18s %5s %5s %7s %6d %10.1f %10.1f %6.2f" % (k[0], k[1], k[2], k[3], k[4], len(v), sum(v) / len(v) / 1e3,
                                                              min(v) / 1e3, 100.0 * sum(v) / tot))
