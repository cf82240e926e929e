"""Single-stream latency of one frame (detect+describe) and one pair (2 frames + match): eager launches vs HIP graph
replay. BASELINE configs[1] (640x480) and configs[2] (1080p). Prints one JSON line per configuration."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import niftymatch_amd as nm  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


for (w, h) in ((640, 480), (1920, 1080)):
    bench.W, bench.H = w, h
    f = bench.make_frames(nm, torch, dev, [0, 1])
    a0, a1 = nm.SiftArena(w, h, bench.CAP, device=dev), nm.SiftArena(w, h, bench.CAP, device=dev)
    ws = nm.MatchWorkspace(bench.CAP, bench.CAP, dev)
    res = torch.full((bench.CAP,), -1, dtype=torch.int32, device=dev)
    a0.detect_describe(f[0]); a1.detect_describe(f[1]); torch.cuda.synchronize()
    n0, n1 = int(a0.num_items.item()), int(a1.num_items.item())
    s = torch.cuda.Stream()

    def frame():
        with torch.cuda.stream(s):
            a0.detect_describe(f[0])

    def pair():
        with torch.cuda.stream(s):
            a0.detect_describe(f[0]); a1.detect_describe(f[1])
            nm.sift_match(a0.desc, a1.desc, 0.8, prior=res, workspace=ws, nA=n0, nB=n1)

    def pair_batched():
        with torch.cuda.stream(s):
            nm.detect_describe_batch([a0, a1], [f[0], f[1]])
            nm.sift_match(a0.desc, a1.desc, 0.8, prior=res, workspace=ws, nA=n0, nB=n1)

    frame(); pair(); pair_batched(); torch.cuda.synchronize()
    gf, gp = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(gf, stream=s):
        a0.detect_describe(f[0])
    with torch.cuda.graph(gp, stream=s):
        a0.detect_describe(f[0]); a1.detect_describe(f[1])
        nm.sift_match(a0.desc, a1.desc, 0.8, prior=res, workspace=ws, nA=n0, nB=n1)
    out = {"frame": "%dx%d" % (w, h), "keypoints": [n0, n1],
           "frame_us_eager": round(timeit(frame), 1), "frame_us_graph": round(timeit(gf.replay), 1),
           "pair_us_eager": round(timeit(pair, 100), 1), "pair_us_graph": round(timeit(gp.replay, 100), 1),
           "pair_us_batched_call": round(timeit(pair_batched, 100), 1)}
    out["frames_per_s_graph"] = round(1e6 / out["frame_us_graph"], 1)
    out["keypoints_per_s_graph"] = round(n0 * 1e6 / out["frame_us_graph"], 1)
    print(json.dumps(out))
