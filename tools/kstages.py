"""Timings of the stages either side of the hot path at 1080p / 12k points: element-wise front end,
warps, RANSAC. Each op is timed over back-to-back launches on one stream."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import niftymatch_amd as nm  # noqa: E402

dev = torch.device("cuda:0")
W, H = 1920, 1080
rng = np.random.default_rng(0)


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


bgra = torch.from_numpy(rng.integers(0, 256, (H, W, 4), dtype=np.uint8)).to(dev)
plane = torch.rand((H, W), device=dev) * 255
print("grayscale 1080p            %7.1f us" % timeit(lambda: nm.grayscale(bgra)))
print("cast f32->u8 1080p         %7.1f us" % timeit(lambda: nm.cast_f32_u8(plane, 0)))
print("downsample uchar4 1080p    %7.1f us" % timeit(lambda: nm.downsample2_u8x4(bgra, W // 2, H // 2)))
yy, xx = torch.meshgrid(torch.arange(H, device=dev, dtype=torch.float32), torch.arange(W, device=dev, dtype=torch.float32), indexing="ij")
xx, yy = xx.contiguous(), yy.contiguous()
cam = torch.tensor([1700.0, 1750.0, 960.0, 540.0], device=dev)
dist = torch.tensor([-0.2, 0.05, -0.01], device=dev)
print("undistort map 1080p        %7.1f us" % timeit(lambda: nm.undistort_map(xx, yy, cam, dist)))
u, v = nm.undistort_map(xx, yy, cam, dist)
gray8 = (plane).to(torch.uint8)
print("resample_undistort 1080p   %7.1f us" % timeit(lambda: nm.resample_undistort(gray8, u, v)))
Hm = torch.tensor([[1.01, 0.02, 5.0], [-0.01, 0.99, -3.0], [1e-5, -1e-5, 1.0]], device=dev)
print("perspective warp 1080p     %7.1f us" % timeit(lambda: nm.resample_perspective(bgra, W, H, Hm, True)))
canvas = torch.zeros((1200, 2100, 4), dtype=torch.uint8, device=dev)
cw = torch.zeros((1200, 2100), device=dev)
mask = torch.ones((H, W), device=dev)
wts = torch.rand((H, W), device=dev) + 0.1
print("transform_blend 1080p      %7.1f us" % timeit(lambda: nm.transform_blend(canvas, cw, bgra, W, H, Hm, 60, 50, mask, wts)))
n = 12000
sx = torch.rand(n, device=dev) * W
sy = torch.rand(n, device=dev) * H
dx, dy = sx * 1.01 + 5, sy * 0.99 - 3
for model, ns, name in ((0, 1, "translation"), (1, 2, "similarity"), (2, 4, "homography")):
    rl = torch.randint(0, n, (4096, ns), dtype=torch.int32, device=dev)
    print("ransac %-11s 4096 hyp x 12k pts %7.1f us" % (name, timeit(lambda: nm.ransac(model, sx, sy, dx, dy, rl, 3.0), 20)))
