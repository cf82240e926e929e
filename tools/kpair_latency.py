"""One 1080p PAIR at batch 1, eager, one stream: a 2-frame nm_sift_detect_describe_batch call + the device-sized match (what
bench.py's latency probe calls pair_us_eager). Prints the mean over 300 back-to-back pairs; environment switches of the library
(NM_DESC_BLOCKS, NM_ORIENT_BLOCKS, ...) are read by the library itself: tools/gpu_env_ab.sh-style alternation from the shell."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import niftymatch_amd as nm
dev = torch.device("cuda:0")
f = bench.make_frames(nm, torch, dev, [0, 1])
a = [nm.SiftArena(bench.W, bench.H, bench.CAP, device=dev) for _ in range(2)]
ws = nm.MatchBatchDevWorkspace(1, bench.CAP, bench.CAP, dev)
res = torch.full((bench.CAP,), -1, dtype=torch.int32, device=dev)
s = torch.cuda.Stream()
single = len(sys.argv) > 1 and sys.argv[1] == "frame"        # one single-frame call instead of the pair
def pair():
    if single:
        a[0].detect_describe(f[0])
        return
    nm.detect_describe_batch(a, f)
    nm.sift_match_batch_dev([a[0].desc], [a[0].num_items], [a[1].desc], [a[1].num_items], [res], 0.8, workspace=ws)
with torch.cuda.stream(s):
    for _ in range(20):
        pair()
    s.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        pair()
    s.synchronize()
    us = (time.perf_counter() - t0) / 300 * 1e6
print(("frame" if single else "pair") + " %s: %.1f us (%d + %d keypoints)" % (" ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("NM_") and k != "NM_BENCH_DETAIL"), us,
                                                  int(a[0].num_items.item()), int(a[1].num_items.item())))
