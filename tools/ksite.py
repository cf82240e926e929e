"""One frame-driver kernel of a 16-frame 1080p detect call alone, through the library's profile sites:
    python tools/ksite.py [describe|orient|detect|pyramid] [frames per call]
median / min us over 10 calls (frame_desc_kernel, frame_orient_kernel, detect_stage_kernel of octave 0, octave-0 pyramid).
NM_HIP_LIB selects a scratch build (tools/build_variant.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import niftymatch_amd as nm
dev = torch.device("cuda:0")
if os.environ.get("KSITE_TALL_MIN"):                # unit groups per launch from which detection takes its tall groups
    nm.set_detect_tall_min(int(os.environ["KSITE_TALL_MIN"]))
what = sys.argv[1] if len(sys.argv) > 1 else "describe"
site = {"describe": nm.PROF_DESCRIBE, "orient": nm.PROF_ORIENT, "detect": nm.PROF_DETECT_O0, "pyramid": nm.PROF_PYRAMID_O0}[what]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
frames = bench.make_frames(nm, torch, dev, list(range(B)))
arenas = [nm.SiftArena(bench.W, bench.H, bench.CAP, device=dev) for _ in range(B)]
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(12)]
for a, b in ev:
    a.record(); b.record()
for a, b in ev:
    nm.profile_events(site, a, b)
    nm.detect_describe_batch(arenas, frames)
    nm.profile_events(site, None, None)
torch.cuda.synchronize()
ts = sorted(a.elapsed_time(b) * 1e3 for a, b in ev[2:])
kp = sum(int(a.num_items.item()) for a in arenas)
print("%s, %d frames, %d keypoints: median %.1f us, min %.1f us = %.2f us per frame" % (what, B, kp, ts[len(ts) // 2], ts[0], ts[len(ts) // 2] / B))
