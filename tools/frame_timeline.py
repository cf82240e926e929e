"""Kernel timeline of ONE 1080p frame at batch 1 (nm_sift_detect_describe on one stream), for the latency work:
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -- python3 tools/frame_timeline.py run
    python3 tools/frame_timeline.py show gpurun_out/tl
`show` prints, for the last frame of the run, every kernel's start offset, duration and the gap to the previous kernel end."""
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if sys.argv[1] == "run":
    import torch
    import bench
    import niftymatch_amd as nm
    dev = torch.device("cuda:0")
    f = bench.make_frames(nm, torch, dev, [0])
    a = nm.SiftArena(bench.W, bench.H, bench.CAP, device=dev)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(5):
            a.detect_describe(f[0])
            s.synchronize()
elif sys.argv[1] == "runpair":            # one PAIR at batch 1: a 2-frame call + the device-sized match (bench.latency_probe's pair)
    import time
    import torch
    import bench
    import niftymatch_amd as nm
    dev = torch.device("cuda:0")
    f = bench.make_frames(nm, torch, dev, [0, 1])
    a = [nm.SiftArena(bench.W, bench.H, bench.CAP, device=dev) for _ in range(2)]
    ws = nm.MatchBatchDevWorkspace(1, bench.CAP, bench.CAP, dev)
    res = torch.full((bench.CAP,), -1, dtype=torch.int32, device=dev)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(5):
            nm.detect_describe_batch(a, f)
            nm.sift_match_batch_dev([a[0].desc], [a[0].num_items], [a[1].desc], [a[1].num_items], [res], 0.8, workspace=ws)
            s.synchronize()
            time.sleep(0.002)
else:
    rows = []
    for fn in glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True):
        rows += list(csv.DictReader(open(fn)))
    rows = [r for r in rows if "at::native" not in r["Kernel_Name"] and "Functor" not in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last frame = the kernels after the last gap > 200 us
    cut = 0
    for i in range(1, len(rows)):
        if int(rows[i]["Start_Timestamp"]) - int(rows[i - 1]["End_Timestamp"]) > 200000:
            cut = i
    rows = rows[cut:]
    t0 = int(rows[0]["Start_Timestamp"])
    end_prev = t0
    busy = 0
    for r in rows:
        st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:44]
        print("%8.1f  dur %7.1f  gap %6.1f  %-44s grid %s" % ((st - t0) / 1e3, (en - st) / 1e3, (st - end_prev) / 1e3, name, r["Grid_Size_X"]))
        end_prev = max(end_prev, en)
        busy += en - st
    print("frame: %d kernels, span %.1f us, sum of kernel times %.1f us" % (len(rows), (end_prev - t0) / 1e3, busy / 1e3))
