"""Single-stream, one-kernel-at-a-time workload for `rocprofv3 --kernel-trace --stats`: N 1080p frames of
detect+describe and N fused matches, each followed by a device sync, so per-kernel durations are isolated."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import bench  # noqa: E402
import niftymatch_amd as nm  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=4)
    ap.add_argument("--size", default="1920x1080")
    ap.add_argument("--batch", type=int, default=1, help="frames per detect call")
    args = ap.parse_args()
    w, h = map(int, args.size.split("x"))
    bench.W, bench.H = w, h
    dev = torch.device("cuda:0")
    frames = bench.make_frames(nm, torch, dev, list(range(args.frames)))
    arenas = [nm.SiftArena(w, h, bench.CAP, device=dev) for _ in range(max(2, args.batch))]
    ws = nm.MatchWorkspace(bench.CAP, bench.CAP, dev)
    res = torch.full((bench.CAP,), -1, dtype=torch.int32, device=dev)
    if args.batch > 1:
        for rep in range(3):
            nm.detect_describe_batch(arenas[:args.batch], [frames[i % args.frames] for i in range(args.batch)])
            torch.cuda.synchronize()
    for rep in range(2):
        for i in range(0, args.frames, 2):
            arenas[0].detect_describe(frames[i]); torch.cuda.synchronize()
            arenas[1].detect_describe(frames[i + 1]); torch.cuda.synchronize()
            nA, nB = int(arenas[0].num_items.item()), int(arenas[1].num_items.item())
            nm.sift_match(arenas[0].desc, arenas[1].desc, 0.8, prior=res, workspace=ws, nA=nA, nB=nB)
            torch.cuda.synchronize()
    print("keypoints", nA, nB, "matches", int((res[:nA] >= 0).sum()))


if __name__ == "__main__":
    main()
