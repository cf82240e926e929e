# usage (GPU box): bash tools/gpu_call_timeline.sh  -- kernel timeline of ONE 64-frame nm_sift_detect_describe_batch call (the last of tools/ksite.py's)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/_ct
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/_ct -- python3 tools/ksite.py describe 64 > /dev/null 2>&1 || exit 1
python3 - <<'PY'
import csv, glob
rows = []
for fn in glob.glob("gpurun_out/_ct/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(fn)))
rows = [r for r in rows if "at::native" not in r["Kernel_Name"] and "Functor" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last call = the kernels from the last base-blur launch (conv_pk_kernel<7, false, false, false> with the largest grid) on
starts = [i for i, r in enumerate(rows) if "conv_pk_kernel<7, false, false, false>" in r["Kernel_Name"] and int(r["Grid_Size_X"]) > 10000000]
rows = rows[starts[-1]:]
t0 = int(rows[0]["Start_Timestamp"]); end_prev = t0
for r in rows:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
    print("%8.1f  dur %7.1f  gap %7.1f  %-44s grid %s" % ((st - t0) / 1e3, (en - st) / 1e3, (st - end_prev) / 1e3, name, r["Grid_Size_X"]))
    end_prev = max(end_prev, en)
print("span %.1f us" % ((end_prev - t0) / 1e3))
PY
rm -rf gpurun_out/_ct
