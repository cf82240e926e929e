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
