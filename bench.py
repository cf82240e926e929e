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


def cpu_baseline():
    """The CPU oracle (a port of the reference's semantics; the reference has no CPU path, src/utils/macros.h:1-8 is the
    whole directory) on one 1080p pair, BASELINE.md section 2: median of 3 repetitions at all host threads (the whole
    pair, nothing scaled) and at 1 thread (both frames whole + the first 1024 query rows of the match, scaled to all
    rows: the scan is linear in the rows). Also returns the oracle's outputs for pair 0 so that the bench can check
    what it timed."""
    sys.path.insert(0, os.path.join(_ROOT, "tests"))
    import statistics
    import oracle_lib as O
    import helpers as Hh
    f0, f1 = Hh.blurred_frame(0, W, H), Hh.blurred_frame(1, W, H)

    def timed(fn):
        t0 = time.time()
        r = fn()
        return time.time() - t0, r

    def one(rows):
        td, (r0, r1) = timed(lambda: (O.sift_detect_describe(f0, CAP), O.sift_detect_describe(f1, CAP)))
        n = r0["n"] if rows is None else min(rows, r0["n"])
        tm, m = timed(lambda: O.sift_matches(r0["desc"][:n], r1["desc"], 0.8, want_distance=False))
        return td, tm * (r0["n"] / float(n)), r0, r1, m

    threads = O.set_threads(0)
    reps_all = [one(None) for _ in range(3)]
    O.set_threads(1)
    reps_1 = [one(1024) for _ in range(3)]
    O.set_threads(0)
    med = lambda reps: statistics.median(td + tm for td, tm, *_ in reps)
    td, tm, r0, r1, m = sorted(reps_all, key=lambda r: r[0] + r[1])[1]
    out = {"value": round(1.0 / med(reps_all), 4), "unit": "frame-pairs/s", "cores": int(threads), "kind": "port",
           "value_1_thread": round(1.0 / med(reps_1), 5), "reps": 3,
           "sample": "median of 3: one whole 1080p pair, nothing scaled, %d threads: both frames detect+describe (%.2fs) + "
                     "%d x %d match (%.2fs); 1 thread: both frames whole + 1024 of the query rows, scaled to all rows"
                     % (threads, td, r0["n"], r1["n"], tm)}
    return out, (r0, r1, m[0])


def allpairs_100k(nm, torch, dist, dev, rank, world, steps=3):
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
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
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
            "verified_sample_vs_fp64": ok}


def detect_256(nm, torch, dist, dev, cdev, rank, world, arenas, streams, B):
    """BASELINE configs[3] as a secondary, separately timed measurement: 256 1080p frames (seeds 0..255), contiguous
    blocks of 256 / world frames per rank (parallel.frames_of_rank), detect+describe only, in B-frame calls spread over
    the detect streams; no data-path collective. Not part of `value`."""
    from niftymatch_amd import parallel
    mine = parallel.frames_of_rank(256, world, rank)
    frames = make_frames(nm, torch, dev, mine)
    B = max(1, min(B, len(arenas)))
    S = max(1, min(len(streams), len(arenas) // B))
    calls = [(k, min(k + B, len(frames))) for k in range(0, len(frames), B)]
    kp = torch.zeros(1, dtype=torch.int64, device=dev)
    kps = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(S)]

    def run(count):
        for c, (b, e) in enumerate(calls):
            s = c % S
            ar = arenas[s * B: s * B + (e - b)]             # a fixed arena set per stream: reuse is ordered by the stream
            with torch.cuda.stream(streams[s]):
                nm.detect_describe_batch(ar, frames[b:e])
                if count:
                    kps[s] += torch.stack([a.num_items[0] for a in ar]).sum()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    run(False)
    barrier()
    t0 = time.perf_counter()
    run(True)
    barrier()
    dt = time.perf_counter() - t0
    for k in kps:
        kp += k
    tot = torch.tensor([dt, float(kp.item()), float(len(frames))], dtype=torch.float64, device=cdev)
    if world > 1:
        mx = tot[:1].clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        tot[0] = mx[0]
    dt, kp_all, n_all = float(tot[0]), float(tot[1]), int(tot[2])
    return {"workload": "configs[3]: %d x 1080p frames, SIFT detect+describe only, %d per rank in %d-frame calls, %d rank(s)"
                        % (n_all, len(frames), B, world),
            "frames_per_s": round(n_all / dt, 1), "keypoints_per_s": round(kp_all / dt, 1),
            "ms_total": round(1e3 * dt, 3), "keypoints_total": int(kp_all), "collective": "none"}


def roofline_pyramid(B, o0_ms, all_ms, traffic, nodog_ms=None):
    """Whole scale-space chain of one B-frame detect call against HBM. `achieved` follows the bench contract: ALGORITHMIC
    bytes (108 B per octave-pixel = 48 Gaussian + 60 DoG; the fused gradient planes add 36) over the
    measured duration. `traffic` is the HBM-side byte count of the same sequence from the rocprofv3 PMC passes in
    profiles/ (fusion keeps it below the algorithmic bytes), and `physical_GBps` / `physical_frac` are what actually
    crossed the fabric per second -- the figure to hold against the HBM peak when asking how busy the memory system is."""
    sum_px = sum((W >> o) * (H >> o) for o in range(6))            # 2 764 020 octave-pixels at 1080p
    alg108 = 108.0 * sum_px * B
    alg144 = 144.0 * sum_px * B
    t_all = traffic.get("pyramid_all", {}).get("hbm_bytes_per_frame")
    t_o0 = traffic.get("pyramid_o0", {}).get("hbm_bytes_per_sequence")
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
            # the chain writes 2.4 bytes per byte it reads; what plain streaming kernels of such mixes move on this device
            # (tools/micro/hbm_mix.hip, profiles/r02_l_hbm_mix_microbench.txt): 1 read : 2 written 3.7-4.3 TB/s, 1 : 3
            # 3.2-5.0 TB/s, copy 4.6-4.7, read-only 5.5-6.5 -- the practical ceiling for `physical_GBps`, not 8 TB/s
            out["write_heavy_streaming_measured_GBps"] = {"1r:2w": [3680, 4340], "1r:3w": [3180, 5010], "copy": [4620, 4730],
                                                          "read_only": [5520, 6510]}
    if nodog_ms:
        # what nm_sift_detect_describe_batch itself runs since round 2: the same chain WITHOUT materialised DoG planes (its
        # detection kernel subtracts consecutive levels): 48 B/px of Gaussian levels (+ 4 for level 5) + 36 of gradients
        out["frame_driver_chain"] = {"note": "the chain the frame driver issues: no DoG planes (detection forms them from the levels)",
                                     "avg_ms": round(nodog_ms, 4), "us_per_frame": round(1e3 * nodog_ms / B, 2),
                                     "algorithmic_bytes": 84.0 * sum_px * B,
                                     "algorithmic_GBps": round(84.0 * sum_px * B / (nodog_ms * 1e-3) / 1e9, 1)}
        t_fd = traffic.get("pyramid_frame_driver", {}).get("hbm_bytes_per_frame")
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
    ap.add_argument("--pairs", type=int, default=32, help="frame pairs per GPU per step")
    ap.add_argument("--streams", type=int, default=4)
    ap.add_argument("--batch", type=int, default=16, help="frames per nm_sift_detect_describe_batch call (16 = 8 pairs)")
    ap.add_argument("--host-threads", type=int, default=1, help="host threads that enqueue the detect calls")
    ap.add_argument("--match-batch", type=int, default=16,
                    help="pairs per nm_sift_match_batch_f32 call (1 = one nm_sift_match_f32 call per pair)")
    ap.add_argument("--match-streams", type=int, default=1,
                    help="streams the fused matches alternate over. 2 hides the small norms/finalize/fallback launches of "
                         "one match under the next match's MFMA kernel (+7 %% frame-pairs/s), but the MFMA kernels of the "
                         "two streams then also contend for CUs and each reads ~20 %% longer: default 1 keeps the "
                         "event-timed matcher launches the isolated-kernel figure")
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
    return ap.parse_args(argv)


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

    P = args.pairs
    B = max(1, min(args.batch, nm.SIFT_MAX_BATCH))
    while (2 * P) % B:
        B -= 1
    NB = 2 * P // B                                   # detect calls per step, B frames each
    S = max(1, min(args.streams, NB))
    # distinct seeds per rank and pair: (2i, 2i+1) is a pair
    seeds = [2 * (rank * P + i) + k for i in range(P) for k in (0, 1)]
    frames = make_frames(nm, torch, dev, seeds)
    streams = [torch.cuda.Stream(device=dev) for _ in range(S)]
    mstream = torch.cuda.Stream(device=dev)
    # one arena per frame of the batch (0.4 GB each): nothing on the hot path is reused before it has been consumed
    arenas = [nm.SiftArena(W, H, CAP, device=dev) for _ in range(2 * P)]
    MB = max(1, min(args.match_batch, nm.MATCH_MAX_BATCH, P))
    bws = nm.MatchBatchWorkspace(MB, CAP, CAP, dev) if MB > 1 else None
    MS = max(1, args.match_streams)
    mstreams = [mstream] + [torch.cuda.Stream(device=dev) for _ in range(MS - 1)]
    wss = [nm.MatchWorkspace(CAP, CAP, dev) for _ in range(MS)]
    ws = wss[0]
    results = [torch.full((CAP,), -1, dtype=torch.int32, device=dev) for _ in range(P)]

    # keypoint counts are data-dependent but deterministic: one untimed pass gives the host-side sizes of each pair
    counts = []
    for c in range(NB):
        nm.detect_describe_batch(arenas[c * B:(c + 1) * B], frames[c * B:(c + 1) * B])
    torch.cuda.synchronize()
    for i in range(P):
        counts.append((int(arenas[2 * i].num_items.item()), int(arenas[2 * i + 1].num_items.item())))

    def mk_events(n):
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for a, b in evs:
            a.record(); b.record()
        return evs
    ev_match = mk_events(P)
    ev_pyr = mk_events(1)
    torch.cuda.synchronize()
    match_ms, pyr_ms = [], []
    done = [torch.cuda.Event() for _ in range(S)]

    from concurrent.futures import ThreadPoolExecutor
    T = max(1, min(args.host_threads, NB))
    pool = ThreadPoolExecutor(T) if T > 1 else None

    def enqueue_detect(t):
        # calls t, t+T, ... of the step, each on its stream (torch's current stream is per host thread; the C ABI
        # itself takes the stream as an argument). With 16-frame calls one host thread is enough (~16 us of host time
        # per frame); several threads matter for small batches, where interleaved issue mixes the calls' kernels.
        for c in range(t, NB, T):
            with torch.cuda.stream(streams[c % S]):
                nm.detect_describe_batch(arenas[c * B:(c + 1) * B], frames[c * B:(c + 1) * B])

    call_done = [torch.cuda.Event() for _ in range(NB)]

    def step_overlapped(timed):
        """--overlap: the matches of the pairs of call c are queued (one match stream) as soon as call c has finished,
        while the next calls' detection proceeds on the other streams. Same work, no probes of isolated sequences."""
        for c in range(NB):
            with torch.cuda.stream(streams[c % S]):
                nm.detect_describe_batch(arenas[c * B:(c + 1) * B], frames[c * B:(c + 1) * B])
                call_done[c].record()
            mstream.wait_event(call_done[c])
            with torch.cuda.stream(mstream):
                for i in range(c * B // 2, (c + 1) * B // 2):
                    nA, nB = counts[i]
                    if timed:
                        nm.profile_events(nm.PROF_MATCH_TOP2, ev_match[i][0], ev_match[i][1])
                    nm.sift_match(arenas[2 * i].desc, arenas[2 * i + 1].desc, 0.8, prior=results[i], workspace=ws, nA=nA, nB=nB)
                if timed:
                    nm.profile_events(nm.PROF_MATCH_TOP2, None, None)
        for s in range(S):
            streams[s].wait_stream(mstream)
        if timed:
            mstream.synchronize()
            match_ms.extend(a.elapsed_time(b) for a, b in ev_match)

    def step(timed):
        """One batch. Detect+describe of the 2P frames = NB calls of B frames each, spread over S streams (all in flight
        together: the latency-bound small-octave and book-keeping launches of one call hide under the others'); the P
        fused matches then run back to back on one stream (a match launch fills the chip by itself); every match launch
        is event-timed. The scale-space chain is timed alone after the timed region (whole-pyramid probe below)."""
        if pool is None:
            enqueue_detect(0)
        else:
            list(pool.map(enqueue_detect, range(T)))
        for s in range(S):
            done[s].record(streams[s])
            mstream.wait_event(done[s])
        if MB > 1:
            # batched matches: norms / finalize / fallback once per call for all its pairs, the MFMA kernel once per pair
            with torch.cuda.stream(mstream):
                for i0 in range(0, P, MB):
                    idx = list(range(i0, min(i0 + MB, P)))
                    keep = nm.profile_event_pairs(nm.PROF_MATCH_TOP2, [ev_match[i] for i in idx]) if timed else None
                    nm.sift_match_batch([arenas[2 * i].desc for i in idx], [arenas[2 * i + 1].desc for i in idx],
                                        [counts[i][0] for i in idx], [counts[i][1] for i in idx],
                                        [results[i] for i in idx], 0.8, workspace=bws)
                    if timed:
                        nm.profile_event_pairs(nm.PROF_MATCH_TOP2, [])
                    del keep
        else:
            for k in range(1, MS):
                mstreams[k].wait_stream(mstream)        # the matches start when the last detect call has finished
            for i in range(P):
                nA, nB = counts[i]
                with torch.cuda.stream(mstreams[i % MS]):
                    if timed:
                        nm.profile_events(nm.PROF_MATCH_TOP2, ev_match[i][0], ev_match[i][1])
                    nm.sift_match(arenas[2 * i].desc, arenas[2 * i + 1].desc, 0.8, prior=results[i], workspace=wss[i % MS],
                                  nA=nA, nB=nB)
            if timed:
                nm.profile_events(nm.PROF_MATCH_TOP2, None, None)
            for k in range(1, MS):
                mstream.wait_stream(mstreams[k])
        for s in range(S):                      # the next step's detects overwrite the arenas: wait for the matches
            streams[s].wait_stream(mstream)
        if timed:
            mstream.synchronize()
            match_ms.extend(a.elapsed_time(b) for a, b in ev_match)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    overlap = args.overlap and B % 2 == 0
    run_step = step_overlapped if overlap else step
    for _ in range(args.warmup):
        run_step(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step(True)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    kp_rank = sum(a + b for a, b in counts)
    cmp_rank = sum(a * b for a, b in counts)
    tot = torch.tensor([float(kp_rank), float(cmp_rank)], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    kp_all, cmp_all = float(tot[0].item()), float(tot[1].item())

    # whole-pyramid probe (after the timed region, chip otherwise idle): the scale-space launches of one B-frame detect call,
    # every octave, timed with events on the stream they run on
    pyr_all_ms = pyr_nodog_ms = None
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
        mstream.synchronize()
        pyr_all_ms = e0.elapsed_time(e1) / reps
        pyr_nodog_ms = e2.elapsed_time(e3) / reps
    except Exception as exc:
        pyr_all_ms = None

    # the same 16-pair batched match calls with the fp32 screen (the round-1/2 kernel), after the timed region, on the
    # otherwise idle chip: its per-launch time is the fp32-MFMA roofline reading that the default screen is compared with
    screen = nm.get_match_screen()
    f32_ms = []
    if rank == 0 and screen != "f32" and MB > 1:
        try:
            nm.set_match_screen("f32")
            res2 = [torch.full((CAP,), -1, dtype=torch.int32, device=dev) for _ in range(MB)]
            idx = list(range(min(MB, P)))
            with torch.cuda.stream(mstream):
                for rep in range(5):                 # the first two calls are warm-up (clock, caches)
                    evs = mk_events(len(idx))
                    keep = nm.profile_event_pairs(nm.PROF_MATCH_TOP2, evs)
                    nm.sift_match_batch([arenas[2 * i].desc for i in idx], [arenas[2 * i + 1].desc for i in idx],
                                        [counts[i][0] for i in idx], [counts[i][1] for i in idx],
                                        [res2[k] for k in range(len(idx))], 0.8, workspace=bws)
                    nm.profile_event_pairs(nm.PROF_MATCH_TOP2, [])
                    mstream.synchronize()
                    del keep
                    if rep >= 2:
                        f32_ms.extend(a.elapsed_time(b) for a, b in evs)
            same = all(torch.equal(res2[k][:counts[i][0]], results[i][:counts[i][0]]) for k, i in enumerate(idx))
            f32_ms = (f32_ms, bool(same), [counts[i] for i in idx])
        except Exception as exc:
            f32_ms = repr(exc)
        finally:
            nm.set_match_screen(screen)

    # what the timed loop left in the arenas / results of pair 0 (rank 0: seeds 0 and 1), for the oracle check below
    snap = None
    if rank == 0:
        nA, nB = counts[0]
        snap = {"n": (nA, nB), "kpts": [arenas[k].kpts[:n].cpu().numpy() for k, n in ((0, nA), (1, nB))],
                "desc": [arenas[k].desc[:n].cpu().numpy() for k, n in ((0, nA), (1, nB))],
                "match": results[0][:nA].cpu().numpy()}

    # the reference's own C++ API driven the way a NiftyMatch application drives it (SiftParams / PyramidData / SiftData +
    # the per-octave compute_* calls + compute_sift_matches), on pair 0, one host thread, one stream; not part of `value`
    dropin = None
    if rank == 0 and not args.no_dropin:
        try:
            import ctypes as C
            dropin = {"workload": "drop-in C++ API client loop on the 1080p pair (nm/src/nm_client.cpp: 2 x per-octave "
                                  "detect+describe + compute_sift_matches), single host thread, NULL stream"}
            for key, wd, reps in (("distance_null", 0, 10), ("distance_materialised", 1, 4)):
                n3 = (C.c_int * 3)()
                us = nm.lib().nm_client_pair_loop(frames[0].data_ptr(), frames[1].data_ptr(), W, H, CAP, reps, wd, n3)
                dropin[key] = {"us_per_pair": round(us, 1), "pairs_per_s": round(1e6 / us, 1), "reps": reps,
                               "keypoints": [n3[0], n3[1]], "matches": n3[2]}
        except Exception as exc:
            dropin = {"error": repr(exc)}

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
        try:
            extra = allpairs_100k(nm, torch, dist, dev, rank, world)
        except Exception as exc:             # never lose the headline line to the secondary measurement
            extra = {"error": repr(exc)}

    if rank == 0:
        pairs_total = P * world * args.steps
        nA, nB = counts[0]
        m_ms = sum(match_ms) / len(match_ms)            # every match launch of the timed region
        p_ms = sum(pyr_ms) / len(pyr_ms) if pyr_ms else float("nan")
        flops = 256.0 * sum(a * b for a, b in counts) / len(counts)      # 2*N*M*128 per launch
        # octave 0, levels 1..5 of the B frames of one call: 40 (Gaussian) + 60 (DoG) + 36 (gradients) B/px
        pyr_bytes = 136.0 * W * H * B
        traffic = {}
        try:
            traffic = json.load(open(os.path.join(_ROOT, "profiles", "pmc_traffic.json")))
        except Exception:
            pass
        t_match = traffic.get("match_top2_kernel", {}).get("hbm_bytes_per_launch")
        ach = flops / (m_ms * 1e-3) / 1e12
        if screen == "f32":
            roof = {"kernel": "match_top2_kernel<f32>", "bound": "mfma", "achieved": round(ach, 3),
                    "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_F32_PEAK_TFLOPS, 4)}
        else:
            # The contract's reading: ALGORITHMIC flops (2NM128) over the launch time, against the dense peak of the
            # dtype the MFMAs run in (bf16). The screen executes 3.125 bf16 flops per algorithmic flop, so the pipe is
            # `frac_executed` busy; against the fp32-MFMA roofline the path's arithmetic is specified in, the same
            # launch reads `vs_f32_mfma_peak` (> 1: faster than any fp32-MFMA formulation can be).
            roof = {"kernel": "match_top2_kernel<bf16x3>", "bound": "mfma", "achieved": round(ach, 3),
                    "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / MFMA_BF16_PEAK_TFLOPS, 4),
                    "executed_TFLOPs": round(ach * BF16X3_EXECUTED_PER_ALGORITHMIC, 3),
                    "frac_executed": round(ach * BF16X3_EXECUTED_PER_ALGORITHMIC / MFMA_BF16_PEAK_TFLOPS, 4),
                    "vs_f32_mfma_peak": round(ach / MFMA_F32_PEAK_TFLOPS, 4),
                    "sustained_bf16_mfma_measured": MFMA_BF16_SUSTAINED_MEASURED_TFLOPS,
                    "frac_executed_of_sustained": round(ach * BF16X3_EXECUTED_PER_ALGORITHMIC / MFMA_BF16_SUSTAINED_MEASURED_TFLOPS, 4),
                    "note": "screen on split bf16 operands (a_h.b_h + a_h.b_l + a_l.b_h); match decisions are made on "
                            "distances recomputed exactly in fp32 (results bit-identical to the fp32 screen and the oracle)"}
        roof.update({"traffic": t_match if screen == "f32" else traffic.get("match_top2_kernel_bf16x3", {}).get("hbm_bytes_per_launch"),
                     "traffic_note": "HBM bytes per launch from the rocprofv3 PMC passes in profiles/ (not live)",
                     "avg_ms": round(m_ms, 4), "launches_timed": len(match_ms), "launch_shape": [nA, nB, 128],
                     "screen": screen})
        roof_f32 = None
        if isinstance(f32_ms, tuple) and f32_ms[0]:
            ms32 = sum(f32_ms[0]) / len(f32_ms[0])
            fl32 = 256.0 * sum(a * b for a, b in f32_ms[2]) / len(f32_ms[2])
            roof_f32 = {"kernel": "match_top2_kernel<f32>", "bound": "mfma", "achieved": round(fl32 / (ms32 * 1e-3) / 1e12, 3),
                        "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(fl32 / (ms32 * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4), "traffic": t_match,
                        "avg_ms": round(ms32, 4), "launches_timed": len(f32_ms[0]),
                        "same_matches_as_default_screen": f32_ms[1],
                        "note": "the fp32 screen (NM_MATCH_SCREEN=f32) on the same pairs, after the timed region"}
        elif isinstance(f32_ms, str):
            roof_f32 = {"error": f32_ms}
        out = {
            "metric": METRIC, "value": round(pairs_total / dt, 3), "unit": "frame-pairs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[2]: SIFT detect+describe x2 + fused BF L2 match per 1920x1080 pair", "match_screen": screen,
                       "pairs_per_gpu_per_step": P, "detect_streams": S, "frames_per_detect_call": B,
                       "host_enqueue_threads": T, "match_streams": MS, "pairs_per_match_call": MB, "phases": "overlapped" if overlap else "detect then match",
                       "keypoints_pair0": [nA, nB], "capacity": CAP,
                       "parallelism": "frame-pair sharding, %d rank(s), no data-path collective" % world},
            "keypoints_per_s": round(kp_all * args.steps / dt, 1),
            "descriptor_comparisons_per_s": round(cmp_all * args.steps / dt, 1),
            "roofline": roof,
            "roofline_pyramid": roofline_pyramid(B, p_ms, pyr_all_ms, traffic, pyr_nodog_ms),
        }
        if roof_f32 is not None:
            out["roofline_f32_screen"] = roof_f32
        if dropin is not None:
            out["dropin_api"] = dropin
        if detect256 is not None:
            out["detect_256"] = detect256
        if extra is not None:
            out["allpairs_100k"] = extra
        if not args.no_cpu_baseline and world == 1:        # reported at N = 1 only (the other ranks would wait for it)
            import numpy as np
            out["cpu_baseline"], (r0, r1, m) = cpu_baseline()
            # the oracle's pair 0 against what the timed loop produced, bit for bit (siftfunctions.cu:100-181, match.cu:83-117)
            ok = snap["n"] == (r0["n"], r1["n"])
            ok = ok and all(np.array_equal(snap["kpts"][k], r["kpts"]) and np.array_equal(snap["desc"][k], r["desc"])
                            for k, r in ((0, r0), (1, r1)))
            ok = ok and np.array_equal(snap["match"], m)
            out["verified_pair0_vs_oracle"] = bool(ok)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
