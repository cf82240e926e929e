"""Host logic of bench.py that needs no GPU: the `--gpus N` self-launcher and the on-device frame generator."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_launcher_branch_is_taken_for_plain_gpus_n():
    """`python bench.py --gpus 8 ...` (the driver's form when it does not use torchrun itself) must become 8 ranks: the
    launcher command is built before anything touches the GPU; inside a rank (WORLD_SIZE set) or at N = 1 it is None."""
    import bench
    argv = ["--gpus", "8", "--steps", "5", "--warmup", "2"]
    cmd = bench.launcher_command(8, argv, {})
    assert cmd is not None and cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-len(argv) - 1] == os.path.join(ROOT, "bench.py") and cmd[-len(argv):] == argv
    assert bench.launcher_command(8, argv, {"WORLD_SIZE": "8"}) is None       # already a rank
    assert bench.launcher_command(1, ["--gpus", "1"], {}) is None


def test_plain_gpus_n_really_spawns_ranks(tmp_path):
    """End to end without a GPU: with --gpus 2 and no WORLD_SIZE the script must come back as two torchrun ranks. The
    ranks stop at `bench.py needs a GPU`; seeing that assertion from a process with RANK set proves the branch."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline", "--no-allpairs"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "bench.py needs a GPU" in r.stderr and "torch.distributed" in r.stderr


def test_world_size_mismatch_fails_loudly():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in r.stderr


def test_synth_torch_equals_numpy():
    from niftymatch_amd import synth
    for seed in (0, 1, 7, 255, 511, 123456):
        a = synth.noise_frame(seed, 333, 77)
        b = synth.noise_frame_torch(seed, 333, 77, "cpu").numpy()
        assert np.array_equal(a, b), seed


def test_eight_rank_partitions_without_a_gpu():
    """What the driver's 8-GPU run relies on (no scaling curve has been measured on hardware): the launcher
    command for N = 8, frame seeds disjoint over 8 ranks and over the sets of a run, configs[3]'s 256 frames as 32 per rank,
    configs[4]'s 100 000 candidates as 12 500 per rank."""
    import bench
    from niftymatch_amd import parallel
    cmd = bench.launcher_command(8, ["--gpus", "8"], {})
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8" and "--nnodes=1" in cmd
    P, world, sets = 32, 8, 125
    seen = set()
    for s in range(sets):
        for r in range(world):
            seeds = bench.seeds_of_set(s, world, r, P)
            assert len(seeds) == 2 * P and all(seeds[2 * i + 1] == seeds[2 * i] + 1 and seeds[2 * i] % 2 == 0 for i in range(P))
            assert not (seen & set(seeds))
            seen |= set(seeds)
    assert len(seen) == sets * world * 2 * P
    assert bench.seeds_of_set(0, 1, 0, P)[:2] == [0, 1]
    for r in range(8):
        assert parallel.frames_of_rank(256, 8, r) == list(range(32 * r, 32 * r + 32))
        assert parallel.block_range(100000, 8, r) == (12500 * r, 12500 * r + 12500)


def test_stored_pmc_constants_carry_the_launch_shape_bench_divides_by():
    """Every entry of profiles/pmc_traffic.json names its launch shape, its round and a committed summary file; bench.py takes
    an entry only when the entry's frames / pairs per launch equal what the bench launches (a per-16-pair figure was once
    multiplied by 16 again: BENCH_r05 roofline.traffic 10.3 GB instead of 0.64 GB)."""
    import json
    import bench
    t = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    args = bench.parse_args([])
    tails = {"match_coarse_kernel": args.match_batch, "match_top2_kernel_f32": 1, "match_top2_kernel_bf16x3": 1,
             "match_top2_group_kernel_f32": 8,
             "distance_mfma_kernel": 1, "pyramid_all": args.batch, "pyramid_frame_driver": args.batch,
             "pyramid_levels_dog_only": args.batch, "frame_desc_kernel": args.batch, "frame_orient_kernel": args.batch,
             "detect_stage_kernel": args.batch}
    for k, e in t.items():
        if k.startswith("_"):
            continue
        assert k in tails, "unknown entry %s: bench.py would not know its launch shape" % k
        assert isinstance(e.get("launch_shape"), list) and e.get("round") and e.get("profile"), k
        assert os.path.exists(os.path.join(ROOT, e["profile"])), (k, e["profile"])
        field = [f for f in ("hbm_bytes_per_launch", "hbm_bytes_per_frame", "sq_insts_valu_per_launch") if f in e]
        assert field, k
        want = e[field[0]] if e["launch_shape"][-1] == tails[k] else None
        assert bench.stored(t, k, field[0], tails[k]) == want
        assert bench.stored(t, k, field[0], tails[k] + 1) is None            # any other launch shape: not used
    assert bench.stored(t, "no_such_kernel", "hbm_bytes_per_launch", 16) is None
    # the coarse pass's figure is per 16-pair launch and must come through unmultiplied
    c = t["match_coarse_kernel"]
    # (round 6, one XCD per pair: 177 MB per 16-pair launch; rounds 3-5: 644 MB; never the 10.3 GB of the x16 bug)
    assert c["launch_shape"][-1] == 16 and 0.14e9 < c["hbm_bytes_per_launch"] < 1.0e9


def test_stdout_line_is_short_and_complete():
    """compact_line on a full-size record (the last committed detail file, or a synthetic one with every object populated):
    under 6 KB, every contract key present, no roofline fraction above 1, dtype names the MFMA arithmetic."""
    import glob
    import json
    import bench
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0[6-9]_*bench_detail*.json")))
    if files:
        full = json.load(open(files[-1]))
    else:
        roof = {"kernel": "match_coarse_kernel", "bound": "mfma", "achieved": 832.6, "peak": 2500.0, "unit": "TFLOP/s", "frac": 0.333,
                "traffic": 643630000, "note": "x" * 400, "traffic_note": "y" * 200, "avg_ms": 0.7295, "launches_timed": 200,
                "avg_launch_flops": 6.07e11, "screen": "f16", "pairs_per_launch": 16, "avg_ms_per_pair": 0.0456, "share_of_step": 0.14}
        full = {"metric": bench.METRIC, "value": 3120.9, "unit": "frame-pairs/s", "n_gpus": 1, "steps": 20, "warmup": 5,
                "ms_per_step": 51.3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": bench.DTYPE["f16"],
                "data": "synthetic", "config": {"workload": "configs[2]: ...", "frames": "z" * 300, "capacity": 16384},
                "summary": {"roofline_frac": 0.333, "value_is": "w" * 200}, "roofline": roof, "roofline_f32_screen": dict(roof, frac=0.8),
                "roofline_pyramid": {"bound": "hbm", "frac": 0.65, "peak": 8000.0, "unit": "GB/s", "achieved": 5200.0, "traffic": 1.6e10,
                                     "avg_ms": 3.67, "levels_dog_only": {"frac": 0.92, "us_per_frame": 40.0, "note": "n" * 200},
                                     "frame_driver_chain": {"us_per_frame": 50.9, "note": "n" * 200}},
                "describe": {"kernel": "frame_desc_kernel", "us_per_frame": 43.4, "keypoints_per_s": 2.8e8},
                "latency": {"what": "q" * 500}, "dropin_api": {"workload": "q" * 500}, "detect_256": {"frames_per_s": 7300.0, "timing": "q" * 300},
                "allpairs_100k": {"ms_per_step": 4.1, "workload": "q" * 200},
                "cpu_baseline": {"value": 1.03, "unit": "frame-pairs/s", "cores": 16, "kind": "port", "sample": "s" * 300}}
    line = bench.compact_line(full, "gpurun_out/bench_detail.json")
    text = json.dumps(line)
    assert len(text) < 6144, len(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "detail_file"):
        assert k in line, k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    assert "MFMA" in bench.DTYPE["f16"] and "f16" in bench.DTYPE["f16"]

    def fracs(o):
        if isinstance(o, dict):
            for k, v in o.items():
                if k.startswith("frac") and isinstance(v, (int, float)):
                    yield k, v
                yield from fracs(v)
    assert all(v <= 1.0 for _, v in fracs(line)), list(fracs(line))
