"""Octave-0 pyramid sequence of a 1080p frame, alone on the device, timed with the launcher's own HIP events
(NM_PROF_PYRAMID_O0): min / median microseconds over N frames."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import niftymatch_amd as nm  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1          # frames per launch sequence
frames = bench.make_frames(nm, torch, dev, list(range(2 * B)))
arenas = [nm.SiftArena(bench.W, bench.H, bench.CAP, device=dev) for _ in range(B)]
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for e in ev:
    e.record()           # torch creates the HIP event lazily
torch.cuda.synchronize()
ts = []
for i in range(40):
    nm.profile_events(nm.PROF_PYRAMID_O0, ev[0], ev[1])
    nm.detect_describe_batch(arenas, [frames[(i & 1) * B + k] for k in range(B)])
    nm.profile_events(nm.PROF_PYRAMID_O0, None, None)
    torch.cuda.synchronize()
    ts.append(ev[0].elapsed_time(ev[1]) * 1e3)
ts = sorted(ts[5:])
alg = 136.0 * bench.W * bench.H * B
print("batch %d pyramid_o0_us min %.1f median %.1f  -> %.0f GB/s algorithmic (%.1f%% of 8 TB/s)  selftest %d"
      % (B, ts[0], ts[len(ts) // 2], alg / ts[len(ts) // 2] / 1e3, alg / ts[len(ts) // 2] / 1e3 / 80.0, nm.selftest_sqrt()))
