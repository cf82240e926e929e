#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X: SIFT detect+describe + brute-force L2 match on 1080p pairs.

    python bench.py --gpus N --steps K --warmup W

One process per GPU (torch.distributed/RCCL only for the barrier and the max-over-ranks clock; the path shards by
frame pair, no data-path collective). A step = one batch of `--pairs` synthetic 1080p frame pairs per GPU, inputs
resident in HBM: per pair 2 x (Gaussian pyramid + DoG + gradients + extrema + orientations + descriptors) and one
fused MFMA brute-force match of the two descriptor sets (~12k x ~12k). Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

_ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, _ROOT)

METRIC = "SIFT keypoints/sec + 128-D L2 matches/sec on 1080p pairs; 1->8 GPU scaling"
W, H, CAP = 1920, 1080, 16384
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 MFMA
MFMA_BF16_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA (~2.5 PF; not the 2:1-sparsity figure)
# MFMA flops the bf16x3 screen EXECUTES per algorithmic flop: 3 products per (a_k, b_k) + one 16-deep k-slot step for
# the norms = (3 * 128 + 16) / 128
BF16X3_EXECUTED_PER_ALGORITHMIC = (3 * 128 + 16) / 128.0
F16_EXECUTED_PER_ALGORITHMIC = (128 + 16) / 128.0      # coarse pass of the two-stage screen: one product + the norm k-slot step
# what a bare v_mfma_f32_32x32x16_bf16 loop sustains on this device (1.8 GHz under dense bf16 MFMA load):
# profiles/r02_k_mfma_bf16_peak_microbench.txt. Reported beside the nominal peak, never instead of it.
MFMA_BF16_SUSTAINED_MEASURED_TFLOPS = 1850.0


def make_frames(nm, torch, dev, seeds):
    """Uniform[0,255) noise (counter-based PRNG) + zero-padded Gaussian pre-blur sigma=4 done by the HIP convolve."""
    from niftymatch_amd import synth
    taps, r = nm.create_kernel_for_sigma(synth.preblur_sigma(W, H))
    taps_d = torch.from_numpy(taps).to(dev)
    out = []
    for s in seeds:
        raw = synth.noise_frame_torch(s, W, H, dev)      # bit-identical to synth.noise_frame, made on the device
        out.append(nm.convolve(raw, taps_d, r))
    torch.cuda.synchronize()
    return out


def host_cpu_share():
    """Threads the CPU baseline runs on = the cores this process is actually granted: min(affinity mask, cgroup v2 / v1 CPU
    quota rounded up); NM_BENCH_CPU_THREADS overrides. Reported as cpu_baseline.cores."""
    env = os.environ.get("NM_BENCH_CPU_THREADS")
    if env:
        return max(1, int(env))
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, -(-q // per)))
        except (OSError, ValueError):
            pass
    return max(1, min(n, 64))     # the oracle's detect + describe does not scale past a few dozen threads


def cpu_baseline(seed0=0, seed1=1):
    """The CPU oracle (a port of the reference's semantics; the reference has no CPU path, src/utils/macros.h:1-8 is the
    whole directory) on one 1080p pair, BASELINE.md section 2: median of 3 repetitions at all host threads (the whole
    pair, nothing scaled) and at 1 thread (both frames whole + the first 1024 query rows of the match, scaled to all
    rows: the scan is linear in the rows). Also returns the oracle's outputs for pair 0 so that the bench can check
    what it timed."""
    sys.path.insert(0, os.path.join(_ROOT, "tests"))
    import statistics
    import oracle_lib as O
    import helpers as Hh
    f0, f1 = Hh.blurred_frame(seed0, W, H), Hh.blurred_frame(seed1, W, H)

    def timed(fn):
        t0 = time.time()
        r = fn()
        return time.time() - t0, r

    def one(rows):
        td, (r0, r1) = timed(lambda: (O.sift_detect_describe(f0, CAP), O.sift_detect_describe(f1, CAP)))
        n = r0["n"] if rows is None else min(rows, r0["n"])
        tm, m = timed(lambda: O.sift_matches(r0["desc"][:n], r1["desc"], 0.8, want_distance=False))
        return td, tm * (r0["n"] / float(n)), r0, r1, m

    # the host cores this process may really use: the affinity mask and the cgroup CPU quota, whichever is smaller (a GPU box
    # shows 256 processors to OpenMP but grants a 1-GPU job 16 of them: 256 spinning threads took 31 s for what one does in 4)
    want = host_cpu_share()
    threads = O.set_threads(want)
    assert threads == want
    reps_all = [one(None) for _ in range(3)]
    assert O.set_threads(1) == 1
    reps_1 = [one(1024) for _ in range(3)]
    assert O.set_threads(want) == threads, "the oracle's thread count was not restored after the 1-thread leg"
    med = lambda reps: statistics.median(td + tm for td, tm, *_ in reps)
    td, tm, r0, r1, m = sorted(reps_all, key=lambda r: r[0] + r[1])[1]
    out = {"value": round(1.0 / med(reps_all), 4), "unit": "frame-pairs/s", "cores": int(threads), "kind": "port",
           "value_1_thread": round(1.0 / med(reps_1), 5), "reps": 3,
           "sample": "median of 3: one whole 1080p pair (frame seeds %d, %d: pair 0 of the last timed step), nothing scaled, "
                     "%d threads: both frames detect+describe (%.2fs) + %d x %d match (%.2fs); 1 thread: both frames whole "
                     "+ 1024 of the query rows, scaled to all rows" % (seed0, seed1, threads, td, r0["n"], r1["n"], tm)}
    return out, (r0, r1, m[0])


def per_rank(torch, dist, cdev, world, value):
    """Every rank's `value` (a float), rank-ordered, on every rank: the first real N > 1 run checks itself (a straggler or
    a rank that did not take part shows in the list)."""
    if world <= 1:
        return [float(value)]
    mine = torch.tensor([float(value)], dtype=torch.float64, device=cdev)
    out = torch.empty(world, dtype=torch.float64, device=cdev)
    dist.all_gather_into_tensor(out, mine)
    return [float(v) for v in out.tolist()]


def allpairs_100k(nm, torch, dist, dev, cdev, rank, world, steps=3):
    """BASELINE config 5 as a secondary, separately timed measurement: all-pairs match of 100 000 x 100 000 random
    descriptors, candidates row-sharded over the ranks, ONE all-gather of 12 B per row per rank, merge on every rank.
    Not part of `value`. Verified on rank 0 against an fp64 brute force for a sample of the queries."""
    from niftymatch_amd import parallel
    n = 100_000
    g = torch.Generator(device=dev).manual_seed(1234)          # same data on every rank
    A = torch.rand((n, 128), device=dev, generator=g)
    B = torch.rand((n, 128), device=dev, generator=g)
    b, e = parallel.block_range(n, world, rank)
    Bs = B[b:e].contiguous()
    ws = nm.MatchWorkspace(n, e - b, dev)
    res = torch.full((n,), -1, dtype=torch.int32, device=dev)

    def shard_fn(Aq, Bq, off):
        return nm.sift_match_shard(Aq, Bq, off, workspace=ws)

    def run():
        return parallel.match_sharded(A, Bs, b, 0.8, prior=res, shard_fn=shard_fn)

    run()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = run()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    ranks_ms = per_rank(torch, dist, cdev, world, 1e3 * dt / steps)
    dt = max(ranks_ms) * steps / 1e3
    ok = None
    if rank == 0:
        q = torch.randint(0, n, (128,), device=dev, generator=g)
        d = torch.cdist(A[q].double(), B.double()) ** 2
        top = d.topk(2, dim=1, largest=False)
        want = torch.where(top.values[:, 0] / top.values[:, 1] < 0.8, top.indices[:, 0], torch.full_like(top.indices[:, 0], -1))
        ok = bool(torch.equal(out[q].long(), want))
    return {"workload": "configs[4]: all-pairs 100k x 100k 128-D, candidates sharded over %d rank(s)" % world,
            "ms_per_step": round(1e3 * dt / steps, 3), "steps": steps,
            "descriptor_comparisons_per_s": round(n * float(n) * steps / dt, 1),
            "tflops_2NM128_aggregate": round(256.0 * n * n * steps / dt / 1e12, 2),
            "collective": "1 x all_gather_into_tensor of (3, N) int32 per match call" if world > 1 else "none (1 rank)",
            "per_rank_ms_per_step": [round(v, 3) for v in ranks_ms], "verified_sample_vs_fp64": ok}


def detect_256(nm, torch, dist, dev, cdev, rank, world, arenas, streams, B, passes=5, frames=None, after_call=None):
    """BASELINE configs[3] as a secondary, separately timed measurement: 256 1080p frames (seeds 0..255), contiguous
    blocks of 256 / world frames per rank (parallel.frames_of_rank), detect+describe only, in B-frame calls spread over
    the detect streams; no data-path collective. Not part of `value`.
    Steady state (round 4): one untimed warm pass, then `passes` passes, each bracketed by the barrier and timed by itself
    -- the MEDIAN pass is what `frames_per_s` reports -- and the same passes once more back to back without a barrier in
    between (`frames_per_s_sustained`: a single pass pays the fill and drain of its 4 streams, which start their octave-0
    launches together and end in their description kernels together; 16 calls are only 4 per stream). Nothing but the
    detect calls is issued inside a timed region: the keypoints are counted afterwards by ONE device-side sum per call of
    the last pass (an untimed pass of its own). `after_call(c, b, e, arenas_of_the_call)` is the test hook
    (tests/test_gpu_bench_config.py snapshots counts and outputs there, on the call's stream)."""
    import statistics
    from niftymatch_amd import parallel
    if frames is None:
        frames = make_frames(nm, torch, dev, parallel.frames_of_rank(256, world, rank))
    B = max(1, min(B, len(arenas)))
    S = max(1, min(len(streams), len(arenas) // B))
    calls = [(k, min(k + B, len(frames))) for k in range(0, len(frames), B)]
    kps = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(S)]

    def run(hook=None):
        for c, (b, e) in enumerate(calls):
            s = c % S
            ar = arenas[s * B: s * B + (e - b)]             # a fixed arena set per stream: reuse is ordered by the stream
            with torch.cuda.stream(streams[s]):
                nm.detect_describe_batch(ar, frames[b:e])
                if hook:
                    hook(c, b, e, ar)

    def count(c, b, e, ar):
        kps[c % S] += torch.stack([a.num_items[0] for a in ar]).sum()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run()
    barrier()
    per_pass = []
    for _ in range(passes):
        t0 = time.perf_counter()
        run()
        barrier()
        per_pass.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    for _ in range(passes):
        run()
    barrier()
    dt_sus = (time.perf_counter() - t0) / passes
    run(count if after_call is None else (lambda c, b, e, ar: (count(c, b, e, ar), after_call(c, b, e, ar))))
    barrier()
    dt = statistics.median(per_pass)
    ranks_ms = per_rank(torch, dist, cdev, world, 1e3 * dt)
    kp = sum(int(k.item()) for k in kps)
    tot = torch.tensor([dt, float(kp), float(len(frames)), dt_sus], dtype=torch.float64, device=cdev)
    if world > 1:
        mx = torch.stack([tot[0], tot[3]])
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        tot[0] = mx[0]
        tot[3] = mx[1]
    dt, kp_all, n_all, dt_sus = float(tot[0]), float(tot[1]), int(tot[2]), float(tot[3])
    return {"workload": "configs[3]: %d x 1080p frames, SIFT detect+describe only, %d per rank in %d-frame calls on %d streams, %d rank(s)"
                        % (n_all, len(frames), B, S, world),
            "frames_per_s": round(n_all / dt, 1), "keypoints_per_s": round(kp_all / dt, 1),
            "frames_per_s_sustained": round(n_all / dt_sus, 1),
            "timing": "median of %d barrier-bracketed passes after one warm pass; sustained = the same %d passes back to back"
                      % (passes, passes),
            "ms_per_pass": [round(1e3 * t, 3) for t in per_pass], "ms_total": round(1e3 * dt, 3),
            "ms_per_pass_sustained": round(1e3 * dt_sus, 3),
            "keypoints_total": int(kp_all), "collective": "none",
            "per_rank_ms": [round(v, 3) for v in ranks_ms]}


def load_traffic():
    try:
        return json.load(open(os.path.join(_ROOT, "profiles", "pmc_traffic.json")))
    except Exception:
        return {}


def stored(traffic, key, field, tail):
    """A stored PMC constant of profiles/pmc_traffic.json, or None unless it was measured on the launch shape it is about to be
    divided by: the LAST element of the entry's `launch_shape` is the frames / pairs per launch (a per-16-pair
    figure was once multiplied by 16 again). tests/test_bench_host.py checks every entry against bench.py's defaults."""
    e = traffic.get(key) or {}
    shape = e.get("launch_shape") or []
    if not shape or shape[-1] != tail:
        return None
    return e.get(field)


def roofline_pyramid(B, o0_ms, all_ms, traffic, nodog_ms=None, dogonly_ms=None):
    """Whole scale-space chain of one B-frame detect call against HBM. `achieved` follows the bench contract: ALGORITHMIC
    bytes (108 B per octave-pixel = 48 Gaussian + 60 DoG; the fused gradient planes add 36) over the
    measured duration. `traffic` is the HBM-side byte count of the same sequence from the rocprofv3 PMC passes in
    profiles/ (fusion keeps it below the algorithmic bytes), and `physical_GBps` / `physical_frac` are what actually
    crossed the fabric per second -- the figure to hold against the HBM peak when asking how busy the memory system is."""
    sum_px = sum((W >> o) * (H >> o) for o in range(6))            # 2 764 020 octave-pixels at 1080p
    alg108 = 108.0 * sum_px * B
    alg144 = 144.0 * sum_px * B
    t_all = stored(traffic, "pyramid_all", "hbm_bytes_per_frame", B)
    t_o0 = None
    out = {"kernel": "scale-space chain of one detect call: base blur + 6 octaves x 5 fused Gaussian+DoG(+gradient,"
                     " +decimation) launches, %d frames per launch" % B,
           "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "algorithmic_bytes": alg108,
           "algorithmic_bytes_with_gradients": alg144}
    if all_ms:
        out.update({"achieved": round(alg108 / (all_ms * 1e-3) / 1e9, 1), "frac": round(alg108 / (all_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "achieved_with_gradients": round(alg144 / (all_ms * 1e-3) / 1e9, 1),
                    "frac_with_gradients": round(alg144 / (all_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "avg_ms": round(all_ms, 4), "us_per_frame": round(1e3 * all_ms / B, 2),
                    "traffic": (t_all * B if t_all else None)})
        if t_all:
            out["physical_GBps"] = round(t_all * B / (all_ms * 1e-3) / 1e9, 1)
            out["physical_frac"] = round(t_all * B / (all_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
    if dogonly_ms:
        # exactly the reference's convolve + compute_dog work: Gaussian levels + DoG planes, WITHOUT the gradient planes the
        # chain above also produces in the same launches (36 B/px that the 108 B/px yardstick does not count)
        out["levels_dog_only"] = {"note": "the same chain without the fused gradient planes: the 108 B per octave-pixel workload alone",
                                  "avg_ms": round(dogonly_ms, 4), "us_per_frame": round(1e3 * dogonly_ms / B, 2),
                                  "achieved": round(alg108 / (dogonly_ms * 1e-3) / 1e9, 1),
                                  "frac": round(alg108 / (dogonly_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        t_do = stored(traffic, "pyramid_levels_dog_only", "hbm_bytes_per_frame", B)
        if t_do:
            out["levels_dog_only"].update({"traffic": t_do * B, "physical_GBps": round(t_do * B / (dogonly_ms * 1e-3) / 1e9, 1)})
    if nodog_ms:
        # what nm_sift_detect_describe_batch itself runs since round 2: the same chain WITHOUT materialised DoG planes (its
        # detection kernel subtracts consecutive levels): 48 B/px of Gaussian levels (+ 4 for level 5) + 36 of gradients
        out["frame_driver_chain"] = {"note": "the chain the frame driver issues: no DoG planes (detection forms them from the levels)",
                                     "avg_ms": round(nodog_ms, 4), "us_per_frame": round(1e3 * nodog_ms / B, 2),
                                     "algorithmic_bytes": 84.0 * sum_px * B,
                                     "algorithmic_GBps": round(84.0 * sum_px * B / (nodog_ms * 1e-3) / 1e9, 1)}
        t_fd = stored(traffic, "pyramid_frame_driver", "hbm_bytes_per_frame", B)
        if t_fd:
            out["frame_driver_chain"].update({"traffic": t_fd * B, "physical_GBps": round(t_fd * B / (nodog_ms * 1e-3) / 1e9, 1)})
    o0_alg = 136.0 * W * H * B           # octave 0, levels 1..5: 40 (Gaussian) + 60 (DoG) + 36 (gradients) B/px
    if o0_ms == o0_ms:                   # not NaN
        out["octave0"] = {"kernel": "octave-0 part of the same chain (5 launches), from the library profile hook during the probe",
                          "algorithmic_GBps": round(o0_alg / (o0_ms * 1e-3) / 1e9, 1), "algorithmic_bytes": o0_alg,
                          "avg_ms": round(o0_ms, 4), "traffic": (t_o0 * B if t_o0 else None),
                          "physical_GBps": (round(t_o0 * B / (o0_ms * 1e-3) / 1e9, 1) if t_o0 else None),
                          "physical_frac": (round(t_o0 * B / (o0_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if t_o0 else None)}
    return out


def describe_report(B, ms, keypoints, traffic):
    """frame_desc_kernel of one B-frame detect call alone on the chip. SURVEY.md 8(d): orientation / descriptor are gather +
    latency / VALU bound, reported as keypoints/s with NO roofline claim (rounds 4-5 printed a fraction of a "VALU issue
    peak" that the kernel exceeded: the yardstick was wrong). The instruction counts per keypoint are the PMC
    pass of tools/pmc_collect.py on this launch shape (stored, cited); the duration is live (library profile site)."""
    out = {"kernel": "frame_desc_kernel, %d frames per launch" % B, "avg_ms": round(ms, 4), "us_per_frame": round(1e3 * ms / B, 2),
           "keypoints": int(keypoints), "keypoints_per_s": round(keypoints / (ms * 1e-3), 1), "roofline": None}
    e = traffic.get("frame_desc_kernel") or {}
    kp = e.get("keypoints_per_launch")
    if kp and (e.get("launch_shape") or [None])[-1] == B:
        out.update({"valu_instructions_per_keypoint": round(e.get("sq_insts_valu_per_launch", 0) / kp, 1),
                    "lds_instructions_per_keypoint": round(e.get("sq_insts_lds_per_launch", 0) / kp, 1),
                    "counters_from": e.get("profile")})
    return out


# what the arithmetic of a step runs in: every image stage, every distance a match is DECIDED on and every output is fp32; the
# matcher's screens (which only select the candidates that are recomputed exactly) run on the MFMA pipe in the named types
DTYPE = {"f16": "f32 decisions; f16/bf16 MFMA screens", "bf16x3": "f32 decisions; bf16x3 MFMA screen", "f32": "f32"}

_ROOF_KEYS = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_ms", "launches_timed", "avg_launch_flops",
              "pairs_per_launch", "avg_ms_per_pair", "frac_executed", "share_of_step", "screen", "same_matches_as_default_screen",
              "algorithmic_bytes", "frac_with_gradients", "physical_GBps", "physical_frac", "us_per_frame", "write_GBps", "shape",
              "error")


def compact_line(out, detail_path):
    """The stdout line: the contract's keys, `summary`, the roofline objects cut down to their figures, `cpu_baseline`, and the
    name of the side file that holds everything (notes, secondary measurements, per-rank lists). Pure; tests/test_bench_host.py
    holds it under 6 KB on a full-size record."""
    cut = lambda r: ({k: r[k] for k in _ROOF_KEYS if k in r and r[k] is not None} if isinstance(r, dict) else r)
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                "vs_baseline", "dtype", "data") if k in out}
    cfg = out.get("config") or {}
    line["config"] = {k: cfg[k] for k in ("workload", "match_screen", "pairs_per_gpu_per_step", "frames_per_detect_call",
                                          "pairs_per_match_call", "capacity", "parallelism") if k in cfg}
    line["summary"] = out.get("summary")
    for k in ("keypoints_per_s", "descriptor_comparisons_per_s", "value_f32_screen", "verified_pair0_vs_oracle", "rccl_ranks_seen",
              "ranks_seen_by_communicator"):
        if out.get(k) is not None:
            line[k] = out[k]
    line["roofline"] = cut(out.get("roofline"))
    for k in ("roofline_f32_screen", "roofline_distance"):
        if out.get(k) is not None:
            line[k] = cut(out[k])
    rp = out.get("roofline_pyramid")
    if isinstance(rp, dict):
        line["roofline_pyramid"] = cut(rp)
        for sub in ("levels_dog_only", "frame_driver_chain"):
            if isinstance(rp.get(sub), dict):
                line["roofline_pyramid"][sub] = {k: rp[sub][k] for k in ("us_per_frame", "frac", "traffic", "physical_GBps") if k in rp[sub]}
    d = out.get("describe")
    if isinstance(d, dict):
        line["describe"] = {k: d[k] for k in ("kernel", "us_per_frame", "keypoints_per_s", "valu_instructions_per_keypoint",
                                              "lds_instructions_per_keypoint", "share_of_step", "error") if k in d}
    for k, keys in (("detect_256", ("frames_per_s", "keypoints_per_s", "error")),
                    ("allpairs_100k", ("ms_per_step", "tflops_2NM128_aggregate", "verified_sample_vs_fp64", "collective", "error"))):
        if isinstance(out.get(k), dict):
            line[k] = {q: out[k][q] for q in keys if q in out[k]}
    if out.get("cpu_baseline") is not None:
        line["cpu_baseline"] = out["cpu_baseline"]
    line["detail_file"] = detail_path
    return line


def seeds_of_set(s, world, rank, P):
    """Frame seeds of set s on `rank`: (2i, 2i + 1) is a pair; distinct per set, rank and pair (tests/test_bench_host.py
    checks the 8-rank partition: no seed is shared by two ranks or two sets)."""
    return [2 * ((s * world + rank) * P + i) + k for i in range(P) for k in (0, 1)]


def launcher_command(gpus, argv, environ):
    """`python bench.py --gpus N` with N > 1 outside torchrun: the command that re-runs this script as N ranks (one per
    GPU), or None when this process is already a rank (WORLD_SIZE set) or N == 1. Pure function: no GPU, no torch."""
    if gpus <= 1 or "WORLD_SIZE" in environ:
        return None
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=32, help="frame pairs per GPU per ROUND (one detect call sequence + its matches)")
    ap.add_argument("--rounds", type=int, default=5,
                    help="rounds per step: a step is `rounds` x `pairs` frame pairs per GPU, every round on frames of its own "
                         "(the arenas are reused round after round, as they are step after step). 5 x 32 pairs = 55 ms per step: "
                         "the driver's 20 timed steps then span > 1 s (a 0.2 s window sits inside the clock ramp)")
    ap.add_argument("--frame-sets", type=int, default=128,
                    help="distinct sets of 2 x pairs synthetic frames held in HBM (8.3 MB per frame, 68 GB at the default); round "
                         "q runs on set q mod frame-sets, so with (warmup + steps) x rounds <= frame-sets no frame is ever seen twice")
    ap.add_argument("--streams", type=int, default=4)
    ap.add_argument("--batch", type=int, default=64,
                    help="frames per nm_sift_detect_describe_batch call (64 = NM_SIFT_MAX_BATCH: the whole step's 32 pairs in ONE "
                         "launch sequence; the per-octave launches of the small octaves and the book-keeping launches cost the same "
                         "for 64 frames as for 16. Round 4, same box: 2 776 frame-pairs/s with 16-frame calls on 4 streams, 2 821-2 885 "
                         "with 32 on 2, 2 893-2 905 with 64 on 1)")
    ap.add_argument("--host-threads", type=int, default=1, help="host threads that enqueue the detect calls")
    ap.add_argument("--match-batch", type=int, default=16, help="pairs per nm_sift_match_batch_dev_f32 call")
    ap.add_argument("--match-streams", type=int, default=1,
                    help="streams the fused matches alternate over. 2 hides the small norms/finalize/fallback launches of "
                         "one match under the next match's MFMA kernel (+7 %% frame-pairs/s), but the MFMA kernels of the "
                         "two streams then also contend for CUs and each reads ~20 %% longer: default 1 keeps the "
                         "event-timed matcher launches the isolated-kernel figure")
    ap.add_argument("--match-pipeline", action="store_true",
                    help="pipeline the match calls of a step by phase: norms / split images (PREP) and the exact finalize + "
                         "fallback (FINISH) of neighbouring calls run on a helper stream beside the MFMA launches (SCREEN), which "
                         "stay back to back on one stream. Measured (round 3): no gain -- 2 199 vs 2 196 frame-pairs/s, the MFMA "
                         "launches read 7 us longer each, exactly what the hidden helpers took -- so not the default.")
    ap.add_argument("--overlap", action="store_true",
                    help="pipeline the matches of a detect call with the next call's detection (higher throughput; the "
                         "matcher then shares the chip, so its roofline reading drops -- not the default)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for N > 1. nccl (= RCCL over xGMI) is the real thing; gloo exists to "
                         "rehearse the multi-rank logic with several ranks sharing one GPU (timing tensors then live on the "
                         "host; combine with --no-allpairs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-allpairs", action="store_true", help="skip the secondary config-5 measurement")
    ap.add_argument("--no-detect256", action="store_true", help="skip the secondary configs[3] measurement")
    ap.add_argument("--no-dropin", action="store_true", help="skip the secondary measurement of the drop-in C++ API loop")
    ap.add_argument("--no-latency", action="store_true", help="skip the batch-1 latency probe")
    ap.add_argument("--no-f32-loop", action="store_true", help="skip the second timed loop with the fp32 MFMA screen")
    return ap.parse_args(argv)


def latency_probe(nm, torch, dev, frames4):
    """Batch-1 figures (the reference's use case is one frame at a time, siftfunctions.cu:42-181): wall
    time per call of back-to-back calls on ONE stream -- a call is one dependent launch chain, so this is its latency
    (or the host's enqueue time where that is longer) -- for one 1080p frame and for one pair (one 2-frame detect call
    + the device-sized match), issued eagerly and replayed as a captured HIP graph. frames4: two pairs; the graphs are
    replayed on the second pair as well (different keypoint counts) and must give what the eager calls give."""
    a = [nm.SiftArena(W, H, CAP, device=dev) for _ in range(2)]
    buf = [frames4[0].clone(), frames4[1].clone()]
    ws = nm.MatchBatchDevWorkspace(1, CAP, CAP, dev)
    res = torch.full((CAP,), -1, dtype=torch.int32, device=dev)
    s = torch.cuda.Stream(device=dev)

    def frame():
        a[0].detect_describe(buf[0])

    def pair():
        nm.detect_describe_batch(a, buf)
        nm.sift_match_batch_dev([a[0].desc], [a[0].num_items], [a[1].desc], [a[1].num_items], [res], 0.8, workspace=ws)

    def timeit(fn, n):
        with torch.cuda.stream(s):
            for _ in range(5):
                fn()
            s.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            s.synchronize()
        return (time.perf_counter() - t0) / n * 1e6

    out = {"what": "batch 1, one stream, back-to-back calls, us per call (1080p); pair = one 2-frame detect call + "
                   "device-sized match"}
    try:
        out["launches_per_frame_call"] = nm.lib().nm_sift_arena_launches_per_call(a[0]._h, 1)
        out["launches_per_16_frame_call"] = nm.lib().nm_sift_arena_launches_per_call(a[0]._h, 16)
        out["frame_us_eager"] = round(timeit(frame, 50), 1)
        out["pair_us_eager"] = round(timeit(pair, 30), 1)
        gf, gp = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(gf, stream=s):
            frame()
        with torch.cuda.graph(gp, stream=s):
            pair()
        out["frame_us_graph"] = round(timeit(gf.replay, 50), 1)
        out["pair_us_graph"] = round(timeit(gp.replay, 30), 1)
        # what a latency-bound client does with two frames: one single-frame call per frame on a stream of its own (each is
        # the 23-launch tail path and leaves most of the chip idle), the match behind both on the first stream
        s2, ev = torch.cuda.Stream(device=dev), torch.cuda.Event()

        def pair_two_streams():
            ev0 = torch.cuda.Event()
            ev0.record(s)
            with torch.cuda.stream(s2):
                s2.wait_event(ev0)                       # ordered behind the previous match (it read a[1])
                a[1].detect_describe(buf[1])
                ev.record(s2)
            a[0].detect_describe(buf[0])
            s.wait_event(ev)
            nm.sift_match_batch_dev([a[0].desc], [a[0].num_items], [a[1].desc], [a[1].num_items], [res], 0.8, workspace=ws)

        out["pair_us_eager_two_streams"] = round(timeit(pair_two_streams, 30), 1)
        # the pair graph on OTHER frames (their keypoint counts differ): must equal the eager calls on those frames
        buf[0].copy_(frames4[2]); buf[1].copy_(frames4[3])
        res.fill_(-1)
        with torch.cuda.stream(s):
            pair()
        s.synchronize()
        n_eager = (int(a[0].num_items.item()), int(a[1].num_items.item()))
        want = res.clone()
        res.fill_(-1)
        for x in a:
            x.num_items.zero_()
        torch.cuda.synchronize()
        gp.replay()
        torch.cuda.synchronize()
        n_graph = (int(a[0].num_items.item()), int(a[1].num_items.item()))
        out["keypoints_second_pair"] = list(n_graph)
        out["graph_replay_on_other_frames_equals_eager"] = bool(n_graph == n_eager and torch.equal(res, want))
    except Exception as exc:
        out["error"] = repr(exc)
    for x in a:
        x.close()
    return out


def main():
    args = parse_args()
    # Before anything touches the GPU: the driver's plain `python bench.py --gpus N` form becomes N ranks as a CHILD
    # process (a process that has initialised HIP must never exec another program on this pool).
    cmd = launcher_command(args.gpus, sys.argv[1:], os.environ)
    if cmd is not None:
        import subprocess
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.exit(subprocess.run(cmd, env=env).returncode)

    import torch
    import torch.distributed as dist
    import niftymatch_amd as nm

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, "--gpus %d but WORLD_SIZE=%d: launch one rank per GPU" % (args.gpus, world)
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    dev_index = local_rank % max(1, torch.cuda.device_count())     # rehearsals may put several ranks on one GPU
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
    cdev = dev if args.backend == "nccl" else torch.device("cpu")   # where the few collective payload tensors live
    nm.lib()
    # how many ranks the communicator really reaches (RCCL when the backend is nccl): a sum of ones over it
    ranks_seen = 1
    if world > 1:
        one = torch.ones(1, dtype=torch.int32, device=cdev)
        dist.all_reduce(one, op=dist.ReduceOp.SUM)
        ranks_seen = int(one.item())

    P = args.pairs
    B = max(1, min(args.batch, nm.SIFT_MAX_BATCH))
    while (2 * P) % B:
        B -= 1
    NB = 2 * P // B                                   # detect calls per step, B frames each
    S = max(1, min(args.streams, NB))
    # Fresh frames every step: n_sets distinct sets of 2P frames; step k (warm-up included) runs on set k mod n_sets.
    # Seeds are distinct per set, rank and pair: (2i, 2i+1) is a pair; set 0 of rank 0 starts at seed 0.
    R = max(1, args.rounds)                           # rounds per step; everything below counts ROUNDS ("sub-steps")
    total_steps = (args.warmup + args.steps) * R
    n_sets = max(1, min(total_steps, args.frame_sets))

    def seeds_of(s):
        return seeds_of_set(s, world, rank, P)
    frame_sets = [make_frames(nm, torch, dev, seeds_of(s)) for s in range(n_sets)]
    # The batch-1 latency probe runs FIRST, while the process holds the streams a latency-bound client holds (its own and two
    # arenas'): HIP spreads streams over a few hardware queues, and among the ~130 streams the throughput loop's 64 arenas
    # create, an arena's side stream can land on the caller's queue -- the call's two branches then run one after the other
    # (434 instead of 297 us per frame in the same process; the captured graph does not care).
    latency = None
    if rank == 0 and not args.no_latency:
        latency = latency_probe(nm, torch, dev, (frame_sets[0] + frame_sets[-1])[:2] + (frame_sets[0] + frame_sets[-1])[-2:])
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    mstream = torch.cuda.Stream(device=dev)
    # one arena per frame of the batch (0.4 GB each): nothing on the hot path is reused before it has been consumed.
    # Their descriptor counts live in ONE device table (the matcher reads them there; nothing comes back to the host).
    counts_all = torch.zeros(2 * P, dtype=torch.int32, device=dev)
    arenas = [nm.SiftArena(W, H, CAP, device=dev, num_items=counts_all[k:k + 1]) for k in range(2 * P)]
    MB = max(1, min(args.match_batch, nm.MATCH_MAX_BATCH, P))
    MS = max(1, args.match_streams)
    mstreams = [mstream] + [torch.cuda.Stream(device=dev) for _ in range(MS - 1)]
    NCALLS = (P + MB - 1) // MB
    pipeline = args.match_pipeline and MS == 1 and not args.overlap
    bwss = [nm.MatchBatchDevWorkspace(MB, CAP, CAP, dev) for _ in range(NCALLS if pipeline else MS)]
    hstream = torch.cuda.Stream(device=dev)             # helper stream of the phase pipeline
    ev_prep = [torch.cuda.Event() for _ in range(NCALLS)]
    ev_screen = [torch.cuda.Event() for _ in range(NCALLS)]
    results = [torch.full((CAP,), -1, dtype=torch.int32, device=dev) for _ in range(P)]
    NSUB = args.steps * R                             # timed rounds
    EV_STEPS = min(NSUB, 128)                         # rounds whose match launches are event-timed
    hist = torch.zeros((max(1, NSUB), 2 * P), dtype=torch.int32, device=dev)    # the counts of every timed round

    def mk_events(n):
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for a, b in evs:
            a.record(); b.record()
        return evs
    ev_pyr = mk_events(1)
    torch.cuda.synchronize()
    pyr_ms = []
    done = [torch.cuda.Event() for _ in range(S)]

    from concurrent.futures import ThreadPoolExecutor
    T = max(1, min(args.host_threads, NB))
    pool = ThreadPoolExecutor(T) if T > 1 else None
    call_done = [torch.cuda.Event() for _ in range(NB)]

    def match_call(idx, evs, stream, ws, out, phases=7):
        """One nm_sift_match_batch_dev_f32 call (or some of its phases) for the pairs idx: the set sizes are the frame
        driver's d_num_items, read on the device (no host read-back anywhere in a step)."""
        with torch.cuda.stream(stream):
            timed = evs and (phases & nm.MATCH_PHASE_SCREEN)
            keep = nm.profile_event_pairs(nm.PROF_MATCH_TOP2, [evs[i] for i in idx]) if timed else None
            nm.sift_match_batch_dev([arenas[2 * i].desc for i in idx], [arenas[2 * i].num_items for i in idx],
                                    [arenas[2 * i + 1].desc for i in idx], [arenas[2 * i + 1].num_items for i in idx],
                                    [out[i] for i in idx], 0.8, workspace=ws, capA=CAP, capB=CAP, phases=phases)
            if timed:
                nm.profile_event_pairs(nm.PROF_MATCH_TOP2, [])
            del keep

    def step(k, evs, hist_row):
        """One batch on frame set k mod n_sets. Detect+describe of the 2P frames = NB calls of B frames each, spread over
        S streams (all in flight together: the latency-bound small-octave and book-keeping launches of one call hide
        under the others'); the P fused matches then run back to back as MB-pair device-sized calls (a match launch
        fills the chip by itself); every MFMA launch of an event-timed step is bracketed by HIP events on its stream."""
        fr = frame_sets[k % n_sets]

        def enqueue_detect(t):
            # calls t, t+T, ... of the step, each on its stream (torch's current stream is per host thread; the C ABI
            # itself takes the stream as an argument). With 16-frame calls one host thread is enough.
            for c in range(t, NB, T):
                with torch.cuda.stream(streams[c % S]):
                    nm.detect_describe_batch(arenas[c * B:(c + 1) * B], fr[c * B:(c + 1) * B])
                    call_done[c].record()
        if args.overlap and B % 2 == 0:
            # the matches of call c are queued as soon as call c has finished, while the next calls' detection proceeds
            for c in range(NB):
                enqueue_detect_one = streams[c % S]
                with torch.cuda.stream(enqueue_detect_one):
                    nm.detect_describe_batch(arenas[c * B:(c + 1) * B], fr[c * B:(c + 1) * B])
                    call_done[c].record()
                mstream.wait_event(call_done[c])
                pairs_c = list(range(c * B // 2, (c + 1) * B // 2))
                for i0 in range(0, len(pairs_c), MB):
                    match_call(pairs_c[i0:i0 + MB], evs, mstream, bwss[0], results)
        else:
            if pool is None:
                enqueue_detect(0)
            else:
                list(pool.map(enqueue_detect, range(T)))
            for s in range(S):
                done[s].record(streams[s])
                mstream.wait_event(done[s])
            calls = [list(range(i0, min(i0 + MB, P))) for i0 in range(0, P, MB)]
            if pipeline:
                # phase pipeline: PREP of every call up front on the helper stream, the MFMA launches back to back on the
                # match stream (each call's behind its PREP), FINISH of call c on the helper stream beside SCREEN of c + 1
                hstream.wait_stream(mstream)
                for ci, idx in enumerate(calls):
                    match_call(idx, None, hstream, bwss[ci], results, nm.MATCH_PHASE_PREP)
                    ev_prep[ci].record(hstream)
                for ci, idx in enumerate(calls):
                    mstream.wait_event(ev_prep[ci])
                    match_call(idx, evs, mstream, bwss[ci], results, nm.MATCH_PHASE_SCREEN)
                    ev_screen[ci].record(mstream)
                for ci, idx in enumerate(calls):
                    hstream.wait_event(ev_screen[ci])
                    match_call(idx, None, hstream, bwss[ci], results, nm.MATCH_PHASE_FINISH)
                mstream.wait_stream(hstream)
            else:
                for q in range(1, MS):
                    mstreams[q].wait_stream(mstream)        # the matches start when the last detect call has finished
                for ci, idx in enumerate(calls):
                    match_call(idx, evs, mstreams[ci % MS], bwss[ci % MS], results)
                for q in range(1, MS):
                    mstream.wait_stream(mstreams[q])
        if hist_row is not None:
            with torch.cuda.stream(mstream):
                hist_row.copy_(counts_all)
        for s in range(S):                      # the next step's detects overwrite the arenas: wait for the matches
            streams[s].wait_stream(mstream)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_loop(k0, evs_all, hist_rows):
        """EXACTLY args.steps steps (= args.steps x R rounds) bracketed by barrier + synchronize; returns (max-over-ranks
        seconds, per-rank ms)."""
        barrier()
        t0 = time.perf_counter()
        for j in range(NSUB):
            step(k0 + j, evs_all[j * P:(j + 1) * P] if (evs_all and j < EV_STEPS) else None,
                 hist_rows[j] if hist_rows is not None else None)
        barrier()
        dt = time.perf_counter() - t0
        ranks_ms = per_rank(torch, dist, cdev, world, 1e3 * dt)
        return max(ranks_ms) / 1e3, ranks_ms

    def launch_times(evs_all, counts_host):
        """(ms, flops) of every event-timed MFMA launch: 2*N*M*128 flop with the sizes the step really had. The two-stage
        screen's coarse pass is ONE launch per match call (persistent over the call's pairs; the library records the
        call's first event pair around it and the others empty): its launches are the calls, with the flops of their pairs."""
        ms = [a.elapsed_time(b) for a, b in evs_all[:EV_STEPS * P]]
        fl = [256.0 * float(counts_host[j][2 * i]) * float(counts_host[j][2 * i + 1]) for j in range(EV_STEPS) for i in range(P)]
        # pairs per MFMA launch under the current screen (the library says: nm_sift_match_pairs_per_launch): the launch's first
        # pair carries its events, the others are recorded back to back
        ppl = pairs_per_launch()
        if ppl > 1:
            calls = [(j * P + k + g, j * P + min(k + g + ppl, k + MB, P)) for j in range(EV_STEPS) for k in range(0, P, MB)
                     for g in range(0, min(MB, P - k), ppl)]
            ms = [sum(ms[a:b]) for a, b in calls]
            fl = [sum(fl[a:b]) for a, b in calls]
        return ms, fl

    def pairs_per_launch():
        return max(1, nm.match_pairs_per_launch(min(MB, P)))

    screen = nm.get_match_screen()
    evs = mk_events(EV_STEPS * P)
    for k in range(args.warmup * R):
        step(k, None, None)
    dt, ranks_ms = timed_loop(args.warmup * R, evs, hist)
    torch.cuda.synchronize()
    counts_host = hist.cpu().tolist()                 # [step][frame]: read AFTER the timed region
    match_ms, match_fl = launch_times(evs, counts_host)
    ppl_default = pairs_per_launch()
    last_set = (total_steps - 1) % n_sets
    # what the timed loop left in the arenas / results of pair 0, for the oracle check below
    snap = None
    if rank == 0:
        nA, nB = counts_host[NSUB - 1][0], counts_host[NSUB - 1][1]
        snap = {"n": (nA, nB), "seeds": tuple(seeds_of(last_set)[:2]),
                "kpts": [arenas[k].kpts[:n].cpu().numpy() for k, n in ((0, nA), (1, nB))],
                "desc": [arenas[k].desc[:n].cpu().numpy() for k, n in ((0, nA), (1, nB))],
                "match": results[0][:nA].cpu().numpy()}
        try:                                              # pair 0 of the last call: rows the second pass / the exact fallback took
            snap["rows"] = bwss[0].row_counts(0)
        except Exception:
            snap["rows"] = None

    kp_rank = float(sum(sum(row) for row in counts_host[:NSUB]))
    cmp_rank = float(sum(row[2 * i] * row[2 * i + 1] for row in counts_host[:NSUB] for i in range(P)))
    tot = torch.tensor([kp_rank, cmp_rank], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    kp_all, cmp_all = float(tot[0].item()), float(tot[1].item())

    # the SAME timed loop with the fp32 MFMA screen (NM_MATCH_SCREEN=f32: north_star's "matcher on MFMA fp32"): its
    # whole-job value and its per-launch roofline come from a timed region of their own, not from a probe
    f32 = None
    if screen != "f32" and not args.no_f32_loop:
        try:
            nm.set_match_screen("f32")
            evs32 = mk_events(EV_STEPS * P)
            hist32 = torch.zeros_like(hist)
            step(total_steps, None, None)                                   # one warm-up step with this screen
            dt32, _ = timed_loop(total_steps + 1, evs32, hist32)
            torch.cuda.synchronize()
            c32 = hist32.cpu().tolist()
            ms32, fl32 = launch_times(evs32, c32)
            ppl32 = pairs_per_launch()
            # the last step again through the default screen, matches only: both screens must emit the same indexes
            nm.set_match_screen(screen)
            res2 = [torch.full((CAP,), -1, dtype=torch.int32, device=dev) for _ in range(P)]
            for i0 in range(0, P, MB):
                match_call(list(range(i0, min(i0 + MB, P))), None, mstream, bwss[0], res2)
            torch.cuda.synchronize()
            row = c32[NSUB - 1]
            same = all(torch.equal(res2[i][:row[2 * i]], results[i][:row[2 * i]]) for i in range(P))
            f32 = {"dt": dt32, "ms": ms32, "fl": fl32, "same": bool(same), "ppl": ppl32}
        except Exception as exc:
            f32 = {"error": repr(exc)}
        finally:
            nm.set_match_screen(screen)

    # whole-pyramid probe (after the timed regions, chip otherwise idle): the scale-space launches of one B-frame detect
    # call, every octave, timed with events on the stream they run on
    frames = frame_sets[0]
    pyr_all_ms = pyr_nodog_ms = pyr_dogonly_ms = None
    try:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(mstream):
            for _ in range(3):
                nm.scale_space_batch(arenas[:B], frames[:B])
            reps = 20
            e0.record()
            for _ in range(reps):
                nm.scale_space_batch(arenas[:B], frames[:B])
            e1.record()
            for _ in range(5):                  # the octave-0 part of the same chain, through the library's profile hook
                nm.profile_events(nm.PROF_PYRAMID_O0, ev_pyr[0][0], ev_pyr[0][1])
                nm.scale_space_batch(arenas[:B], frames[:B])
                nm.profile_events(nm.PROF_PYRAMID_O0, None, None)
                mstream.synchronize()
                pyr_ms.append(ev_pyr[0][0].elapsed_time(ev_pyr[0][1]))
            e2, e3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            nm.scale_space_batch(arenas[:B], frames[:B], write_dog=False)
            e2.record()
            for _ in range(reps):
                nm.scale_space_batch(arenas[:B], frames[:B], write_dog=False)
            e3.record()
            e4, e5 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            nm.scale_space_batch(arenas[:B], frames[:B], write_dog=True, write_grad=False)
            e4.record()
            for _ in range(reps):
                nm.scale_space_batch(arenas[:B], frames[:B], write_dog=True, write_grad=False)
            e5.record()
        mstream.synchronize()
        pyr_all_ms = e0.elapsed_time(e1) / reps
        pyr_nodog_ms = e2.elapsed_time(e3) / reps
        pyr_dogonly_ms = e4.elapsed_time(e5) / reps
    except Exception:
        pyr_all_ms = None

    # descriptor kernel of one B-frame detect call alone (library profile hook; chip otherwise idle), against the VALU issue
    # rate: it is the largest single kernel of a step and has no memory or MFMA yardstick
    desc_roof = None
    try:
        ed = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(6)]
        ms_d = []
        with torch.cuda.stream(mstream):
            for a, b in ed:                      # torch creates the HIP event at its first record
                a.record(); b.record()
            for a, b in ed:
                nm.profile_events(nm.PROF_DESCRIBE, a, b)
                nm.detect_describe_batch(arenas[:B], frames[:B])
                nm.profile_events(nm.PROF_DESCRIBE, None, None)
        mstream.synchronize()
        ms_d = sorted(a.elapsed_time(b) for a, b in ed[1:])
        kp_call = float(sum(int(a.num_items.item()) for a in arenas[:B]))
        desc_roof = describe_report(B, ms_d[len(ms_d) // 2], kp_call, load_traffic())
    except Exception as e:
        desc_roof = {"error": repr(e)}

    # the reference's own C++ API driven the way a NiftyMatch application drives it (SiftParams / PyramidData / SiftData +
    # the per-octave compute_* calls + compute_sift_matches), on one pair, one host thread, one stream; not part of `value`
    dropin = None
    if rank == 0 and not args.no_dropin:
        try:
            import ctypes as C
            dropin = {"workload": "drop-in C++ API client loop on the 1080p pair (nm/src/nm_client.cpp: 2 x per-octave "
                                  "detect+describe + compute_sift_matches), single host thread, NULL stream",
                      "distance_mode": nm.get_distance_mode()}
            for key, wd, reps, nstreams in (("distance_null", 0, 10, 1), ("distance_materialised", 1, 4, 1),
                                            ("distance_null_two_streams", 0, 10, 2), ("distance_materialised_two_streams", 1, 4, 2)):
                n3 = (C.c_int * 3)()
                us = nm.lib().nm_client_pair_loop_ex(frames[0].data_ptr(), frames[1].data_ptr(), W, H, CAP, reps, wd, nstreams, n3)
                dropin[key] = {"us_per_pair": round(us, 1), "pairs_per_s": round(1e6 / us, 1), "reps": reps,
                               "host_threads_and_streams": nstreams, "keypoints": [n3[0], n3[1]], "matches": n3[2]}
        except Exception as exc:
            dropin = {"error": repr(exc)}

    # the materialised N x M distance matrix of compute_sift_matches on pair 0's real descriptors (nm_sift_match_f32 with
    # `distance`): the fp32 MFMA pass (distance_mfma_kernel, library profile site) and the whole call, beside the exact VALU
    # kernel's call; its own roofline entry -- MFMA fp32 by the 2 N M 128 flops, and the N x M x 4 bytes it writes
    dist_roof = None
    if rank == 0 and not args.no_dropin:
        try:
            nm.detect_describe_batch(arenas[:2], frames[:2])
            torch.cuda.synchronize()
            dA, dB = int(arenas[0].num_items.item()), int(arenas[1].num_items.item())
            wsd = nm.MatchWorkspace(dA, dB, dev)
            Dm = torch.empty((dA, dB), dtype=torch.float32, device=dev)
            rd = torch.full((dA,), -1, dtype=torch.int32, device=dev)
            before = nm.get_distance_mode()
            times = {}
            for mode in ("mfma", "exact"):
                nm.set_distance_mode(mode)
                ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(12)]
                for q in ev:
                    for e in q:
                        e.record()
                for q in ev:
                    nm.profile_events(nm.PROF_DISTANCE, q[2], q[3])
                    q[0].record()
                    nm._check(nm.lib().nm_sift_match_f32(arenas[0].desc.data_ptr(), dA, arenas[1].desc.data_ptr(), dB, Dm.data_ptr(),
                                                         rd.data_ptr(), 0.8, wsd.buf.data_ptr(), torch.cuda.current_stream().cuda_stream),
                              "nm_sift_match_f32")
                    q[1].record()
                    nm.profile_events(nm.PROF_DISTANCE, None, None)
                torch.cuda.synchronize()
                call = sorted(q[0].elapsed_time(q[1]) for q in ev[2:])
                kern = sorted(q[2].elapsed_time(q[3]) for q in ev[2:])
                times[mode] = (call[len(call) // 2], kern[len(kern) // 2])
            nm.set_distance_mode(before)
            listed, cap = nm.match_distance_listed(wsd, dA, dB)
            fl, k_ms = 256.0 * dA * dB, times["mfma"][1]
            dist_roof = {"kernel": "distance_mfma_kernel", "bound": "mfma", "achieved": round(fl / (k_ms * 1e-3) / 1e12, 3),
                         "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(fl / (k_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                         "avg_ms": round(k_ms, 4), "shape": [dA, dB, 128], "bytes_written": 4.0 * dA * dB,
                         "write_GBps": round(4.0 * dA * dB / (k_ms * 1e-3) / 1e9, 1),
                         "call_us_mfma": round(1e3 * times["mfma"][0], 1), "call_us_exact_kernel": round(1e3 * times["exact"][0], 1),
                         "blocks_listed_for_exact_recompute": listed, "blocks_total": ((dA + 31) // 32) * ((dB + 31) // 32),
                         "note": "median of 10 nm_sift_match_f32 calls with `distance` on pair 0's descriptors; kernel = the library's "
                                 "profile site around distance_mfma_kernel; call = centring + MFMA pass + fix-up + the fused match. Every "
                                 "entry within 1e-4 relative of the reference's chain (tests/test_gpu_match.py)"}
            del Dm, wsd
        except Exception as exc:
            dist_roof = {"error": repr(exc)}

    detect256 = None
    if not args.no_detect256:
        try:
            detect256 = detect_256(nm, torch, dist, dev, cdev, rank, world, arenas, streams, B)
        except Exception as exc:
            detect256 = {"error": repr(exc)}

    extra = None
    if not args.no_allpairs:
        for a in arenas:                     # give the memory back before the 100k x 100k workspaces
            a.close()
        frame_sets = frames = None
        try:
            extra = allpairs_100k(nm, torch, dist, dev, cdev, rank, world)
        except Exception as exc:             # never lose the headline line to the secondary measurement
            extra = {"error": repr(exc)}

    if rank == 0:
        pairs_total = P * R * world * args.steps
        nA, nB = snap["n"]
        m_ms = sum(match_ms) / len(match_ms)            # every event-timed MFMA launch of the timed region
        p_ms = sum(pyr_ms) / len(pyr_ms) if pyr_ms else float("nan")
        traffic = load_traffic()

        def roof_of(scr, ms, fl, ppl):
            # ALGORITHMIC flops (2NM128, with every launch's own N and M) over the summed launch time
            ach = sum(fl) / (sum(ms) * 1e-3) / 1e12
            if scr == "f32":
                r = {"kernel": "match_top2_group_kernel<f32>" if ppl > 1 else "match_top2_kernel<f32>", "bound": "mfma",
                     "achieved": round(ach, 3),
                     "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_F32_PEAK_TFLOPS, 4),
                     "traffic": stored(traffic, "match_top2_group_kernel_f32" if ppl > 1 else "match_top2_kernel_f32",
                                       "hbm_bytes_per_launch", ppl)}
            elif scr == "f16":
                # two-stage screen: the timed kernel is its coarse pass, ONE fp16 product per k (+ one 16-deep k-slot step for
                # the norms: 1.125 executed flops per algorithmic flop) against the dense fp16 peak. The bf16x3 second pass
                # over the ~1 % of the rows the coarse pass cannot prove is one launch per CALL and is in `value`, not here.
                r = {"kernel": "match_coarse_kernel", "bound": "mfma", "achieved": round(ach, 3),
                     "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4),
                     "executed_TFLOPs": round(ach * F16_EXECUTED_PER_ALGORITHMIC, 3),
                     "frac_executed": round(ach * F16_EXECUTED_PER_ALGORITHMIC / MFMA_BF16_PEAK_TFLOPS, 4),
                     "vs_f32_mfma_peak": round(ach / MFMA_F32_PEAK_TFLOPS, 4),
                     "traffic": stored(traffic, "match_coarse_kernel", "hbm_bytes_per_launch", MB),
                     "note": "coarse pass on fp16 images (a_h.b_h, v_mfma_f32_32x32x16_f16); rows whose coarse result the "
                             "residual-norm bound cannot prove are screened again on split bf16 operands; match decisions are "
                             "made on distances recomputed exactly in fp32 (results bit-identical to the other screens and "
                             "the oracle)"}
            else:
                # against the dense peak of the dtype the MFMAs run in (bf16). The screen executes 3.125 bf16 flops per
                # algorithmic flop, so the pipe is `frac_executed` busy; against the fp32-MFMA roofline the path's
                # arithmetic is specified in, the same launch reads `vs_f32_mfma_peak`.
                r = {"kernel": "match_top2_kernel<bf16x3>", "bound": "mfma", "achieved": round(ach, 3),
                     "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4),
                     "executed_TFLOPs": round(ach * BF16X3_EXECUTED_PER_ALGORITHMIC, 3),
                     "frac_executed": round(ach * BF16X3_EXECUTED_PER_ALGORITHMIC / MFMA_BF16_PEAK_TFLOPS, 4),
                     "vs_f32_mfma_peak": round(ach / MFMA_F32_PEAK_TFLOPS, 4),
                     "sustained_bf16_mfma_measured": MFMA_BF16_SUSTAINED_MEASURED_TFLOPS,
                     "frac_executed_of_sustained": round(ach * BF16X3_EXECUTED_PER_ALGORITHMIC / MFMA_BF16_SUSTAINED_MEASURED_TFLOPS, 4),
                     "traffic": stored(traffic, "match_top2_kernel_bf16x3", "hbm_bytes_per_launch", 1) if ppl == 1 else None,
                     "note": "screen on split bf16 operands (a_h.b_h + a_h.b_l + a_l.b_h); match decisions are made on "
                             "distances recomputed exactly in fp32 (results bit-identical to the fp32 screen and the oracle)"}
            r.update({"traffic_note": "HBM bytes per launch (as launched here: `pairs_per_launch` pairs for the coarse pass) from the "
                                      "rocprofv3 PMC passes of tools/pmc_collect.py in profiles/ (stored, not live; null if the stored "
                                      "launch shape is not this run's)",
                      "avg_ms": round(sum(ms) / len(ms), 4), "launches_timed": len(ms),
                      "avg_launch_flops": round(sum(fl) / len(fl), 1), "screen": scr})
            if ppl > 1:                                   # one launch = ppl pairs of a match call (the coarse pass: all MB of them)
                r.update({"pairs_per_launch": ppl, "avg_ms_per_pair": round(sum(ms) / len(ms) / ppl, 4)})
            return r
        roof = roof_of(screen, match_ms, match_fl, ppl_default)
        roof["launch_shape_last_step_pair0"] = [nA, nB, 128]
        if snap.get("rows"):
            roof["rows_second_pass_sample"], roof["rows_exact_fallback_sample"] = snap["rows"]     # first pair of the last match call
        head = {}
        out = {
            "metric": METRIC, "value": round(pairs_total / dt, 3), "unit": "frame-pairs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE.get(screen, "f32"), "data": "synthetic",
            "config": {"workload": "configs[2]: SIFT detect+describe x2 + fused BF L2 match per 1920x1080 pair",
                       "match_screen": {"f16": "two-stage: f16 coarse pass + bf16x3 second pass on the unproven rows"}.get(screen, screen),
                       "match_sizes": "device",
                       "frames": "fresh every round: %d distinct sets of %d frames per rank, round q on set q mod %d "
                                 "((warmup + steps) x rounds = %d)" % (n_sets, 2 * P, n_sets, total_steps),
                       "pairs_per_gpu_per_step": P * R, "rounds_per_step": R, "pairs_per_round": P,
                       "detect_streams": S, "frames_per_detect_call": B,
                       "host_enqueue_threads": T, "match_streams": MS, "pairs_per_match_call": MB,
                       "phases": "overlapped" if (args.overlap and B % 2 == 0) else "detect then match",
                       "match_pipeline": ("PREP / FINISH of neighbouring calls on a helper stream beside the MFMA launches" if pipeline
                                          else "whole calls on %d stream(s)" % MS),
                       "keypoints_pair0_last_step": [nA, nB], "capacity": CAP,
                       "parallelism": "frame-pair sharding, %d rank(s), no data-path collective" % world},
            "keypoints_per_s": round(kp_all / dt, 1),
            "descriptor_comparisons_per_s": round(cmp_all / dt, 1),
            "per_rank_ms": [round(v, 3) for v in ranks_ms],
            "ranks_seen_by_communicator": ranks_seen, "backend": (args.backend if world > 1 else None),
            "roofline": roof,
            "roofline_pyramid": roofline_pyramid(B, p_ms, pyr_all_ms, traffic, pyr_nodog_ms, pyr_dogonly_ms),
            "describe": desc_roof,
        }
        if world > 1 and args.backend == "nccl":
            out["rccl_ranks_seen"] = ranks_seen
        if f32 is not None:
            if "error" in f32:
                out["roofline_f32_screen"] = {"error": f32["error"]}
            else:
                out["value_f32_screen"] = round(pairs_total / f32["dt"], 3)
                out["roofline_f32_screen"] = roof_of("f32", f32["ms"], f32["fl"], f32["ppl"])
                # also in the HEAD of the line (a truncated tail of the line still shows them)
                head["value_f32_screen"] = out["value_f32_screen"]
                head["roofline_f32_screen_frac"] = out["roofline_f32_screen"]["frac"]
                out["roofline_f32_screen"].update({"same_matches_as_default_screen": f32["same"],
                                                   "note": "the same timed loop (EXACTLY --steps steps, barrier + synchronize "
                                                           "both sides) with the fp32 MFMA screen, run after the headline loop"})
        if latency is not None:
            out["latency"] = latency
        if dropin is not None:
            out["dropin_api"] = dropin
        if dist_roof is not None:
            if "error" not in dist_roof:
                dist_roof["traffic"] = stored(traffic, "distance_mfma_kernel", "hbm_bytes_per_launch", 1)
                dist_roof["traffic_note"] = "HBM bytes per launch from the rocprofv3 PMC passes in profiles/ (stored, not live)"
            out["roofline_distance"] = dist_roof
            head["roofline_distance_frac"] = dist_roof.get("frac")
            head["distance_12k_us"] = (round(1e3 * dist_roof["avg_ms"], 1) if dist_roof.get("avg_ms") else None)
        if detect256 is not None:
            out["detect_256"] = detect256
        if extra is not None:
            out["allpairs_100k"] = extra
        if not args.no_cpu_baseline and world == 1:        # reported at N = 1 only (the other ranks would wait for it)
            import numpy as np
            out["cpu_baseline"], (r0, r1, m) = cpu_baseline(*snap["seeds"])
            # the oracle's pair against what the timed loop produced, bit for bit (siftfunctions.cu:100-181, match.cu:83-117)
            ok = snap["n"] == (r0["n"], r1["n"])
            ok = ok and all(np.array_equal(snap["kpts"][k], r["kpts"]) and np.array_equal(snap["desc"][k], r["desc"])
                            for k, r in ((0, r0), (1, r1)))
            ok = ok and np.array_equal(snap["match"], m)
            out["verified_pair0_vs_oracle"] = bool(ok)
        # share of the step each roofline object's kernel (family) accounts for: its solo duration x launches per step over
        # the measured step (the kernels of a step run one after the other: the step is the sum of their solo times)
        step_ms = 1e3 * dt / args.steps
        roof["share_of_step"] = round(sum(match_ms) / max(1, EV_STEPS) * NSUB / args.steps / step_ms, 4)
        rp = out["roofline_pyramid"]
        if isinstance(rp, dict) and rp.get("frame_driver_chain"):
            rp["share_of_step"] = round(rp["frame_driver_chain"]["avg_ms"] * NB * R / step_ms, 4)
            rp["share_of_step_note"] = "the chain the frame driver runs (frame_driver_chain.avg_ms) x detect calls per step / ms_per_step"
        if isinstance(desc_roof, dict) and desc_roof.get("avg_ms"):
            desc_roof["share_of_step"] = round(desc_roof["avg_ms"] * NB * R / step_ms, 4)
        # the figures a reader needs first go to the head of the line: the driver keeps only a tail of a long stdout
        head.update({"roofline_pyramid_frac": (rp or {}).get("frac"),
                     "roofline_pyramid_levels_dog_only_frac": ((rp or {}).get("levels_dog_only") or {}).get("frac"),
                     "frame_driver_chain_us_per_frame": ((rp or {}).get("frame_driver_chain") or {}).get("us_per_frame"),
                     "frame_desc_us_per_frame": (desc_roof or {}).get("us_per_frame"),
                     "dropin_api_pairs_per_s": ({k: (dropin.get(k) or {}).get("pairs_per_s") for k in
                                                 ("distance_null", "distance_materialised", "distance_null_two_streams",
                                                  "distance_materialised_two_streams")} if isinstance(dropin, dict) and "error" not in dropin else None),
                     "value_is": "batch ABI (nm_sift_detect_describe_batch + nm_sift_match_batch_dev_f32); the reference's own C++ API "
                                 "call sequence is dropin_api_pairs_per_s"})
        head.update({"roofline_frac": roof.get("frac"), "roofline_kernel": roof.get("kernel"),
                     "verified_pair0_vs_oracle": out.get("verified_pair0_vs_oracle"),
                     "latency_us": ({k: latency.get(k) for k in ("launches_per_frame_call", "frame_us_eager", "frame_us_graph", "pair_us_eager", "pair_us_graph", "pair_us_eager_two_streams")}
                                    if isinstance(latency, dict) else None),
                     "detect_256_frames_per_s": (detect256 or {}).get("frames_per_s")})
        out["summary"] = head
        # ONE short line on stdout (the driver keeps a tail of stdout: VERDICT r5 asked for < 6 KB); everything else in a side file
        detail = os.environ.get("NM_BENCH_DETAIL", os.path.join(_ROOT, "gpurun_out", "bench_detail.json"))
        try:
            os.makedirs(os.path.dirname(detail), exist_ok=True)
            json.dump(out, open(detail, "w"), indent=1)
        except OSError as exc:
            detail = "not written: %r" % (exc,)
        print(json.dumps(compact_line(out, os.path.relpath(detail, _ROOT) if os.path.isabs(detail) else detail)))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
