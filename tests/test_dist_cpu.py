"""world_size-2 gloo tests of the multi-GPU host logic (sharding arithmetic + the single all-gather + merge order).
The shard/merge compute is injected (oracle on CPU) because the HIP kernels need a GPU; the collective path is real."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers as H
from niftymatch_amd import parallel

MIN2_INIT = np.float32(2139095040.0)


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _cpu_merge(m1_all, ix_all, m2_all, ambiguity, prior):
    """Reference merge (ascending shard order, strict <) in plain torch; the scan's initial min2 survives only while the
    minimum sits at global candidate 0 (match.cu:91,97)."""
    n_shards, nA = m1_all.shape
    res = torch.full((nA,), -1, dtype=torch.int32) if prior is None else prior.clone()
    for i in range(nA):
        m1, ix, m2 = float(m1_all[0, i]), int(ix_all[0, i]), float(m2_all[0, i])
        for g in range(1, n_shards):
            a1, ai, a2 = float(m1_all[g, i]), int(ix_all[g, i]), float(m2_all[g, i])
            if a1 < m1:
                m2 = min(m1, a2); m1 = a1; ix = ai
            else:
                m2 = min(m2, a1)
        if ix <= 0:
            m2 = min(m2, float(MIN2_INIT))
        if m2 > 0:
            res[i] = ix if np.float32(m1) / np.float32(m2) < np.float32(ambiguity) else -1
    return res


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as O
    O.set_threads(2)
    A = H.synth.descriptors(1, 300)
    B = H.synth.descriptors(2, 501)
    B[37] = B[400]                                   # duplicate across the shard boundary: lower index must win
    A[5] = B[400] + np.float32(1e-3)
    b, e = parallel.block_range(len(B), world, rank)

    def shard_fn(Aq, Bs, off):
        m1, ix, m2 = O.sift_match_shard(Aq.numpy(), Bs.numpy(), off)
        return torch.from_numpy(m1), torch.from_numpy(ix), torch.from_numpy(m2)

    # ambiguity 1.5: the duplicated candidate (ratio exactly 1) is reported, so the tie-break is observable
    res = parallel.match_sharded(torch.from_numpy(A), torch.from_numpy(B[b:e].copy()), b, 1.5, shard_fn=shard_fn,
                                 merge_fn=_cpu_merge)
    ref, _, _ = O.sift_matches(A, B, 1.5, want_distance=False)
    frames = parallel.frames_of_rank(7, world, rank)
    q.put((rank, res.numpy().tolist() == ref.tolist(), int(ref[5]), frames))
    dist.destroy_process_group()


def test_block_ranges_partition():
    for n in (0, 1, 7, 256, 100000):
        for world in (1, 2, 3, 8):
            ranges = [parallel.block_range(n, world, r) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
            assert max(e - b for b, e in ranges) - min(e - b for b, e in ranges) <= 1
    assert parallel.block_range(100000, 8, 3) == (37500, 50000)


def test_sharded_match_two_ranks_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=90) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert out[0][1] and out[1][1]
    assert out[0][2] == 37                            # the duplicate's LOWER global index, owned by rank 0
    assert out[0][3] == [0, 1, 2, 3] and out[1][3] == [4, 5, 6]


def _worker4(rank, world, port, q):
    """Four ranks: (a) 3 candidates over 4 ranks -- the last shard is EMPTY (neutral triple: +inf, -1, +inf) and candidates 1
    and 2 (ranks 1 and 2) are equal: the lower global index wins; (b) 1 003 candidates, a tie between shards 0 and 3 and an
    exact copy of a query in shard 2 (min1 = 0)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as O
    O.set_threads(1)

    def shard_fn(Aq, Bs, off):
        m1, ix, m2 = O.sift_match_shard(Aq.numpy(), Bs.numpy(), off)
        return torch.from_numpy(m1), torch.from_numpy(ix), torch.from_numpy(m2)

    out = []
    for nA, nB, dups, copies in ((40, 3, [(1, 2)], []), (200, 1003, [(17, 900)], [(9, 600)])):
        A = H.synth.descriptors(11, nA)
        B = H.synth.descriptors(12, nB)
        for lo, hi in dups:
            B[lo] = B[hi]
            A[3] = B[hi] + np.float32(1e-3)
        for qa, jb in copies:
            A[qa] = B[jb]
        b, e = parallel.block_range(nB, world, rank)
        res = parallel.match_sharded(torch.from_numpy(A), torch.from_numpy(B[b:e].copy()), b, 1.5, shard_fn=shard_fn,
                                     merge_fn=_cpu_merge)
        ref, _, _ = O.sift_matches(A, B, 1.5, want_distance=False)
        out.append((res.numpy().tolist() == ref.tolist(), int(ref[3]), e - b, [int(ref[qa]) for qa, _ in copies]))
    q.put((rank, out))
    dist.destroy_process_group()


def test_sharded_match_four_ranks_gloo_empty_shard_and_cross_shard_tie():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker4, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [o[1][0][2] for o in out] == [1, 1, 1, 0]            # case (a): the fourth shard is empty
    for rank, (a, b) in out:
        assert a[0] and b[0], rank                              # every rank holds the unsharded answer
        assert a[1] == 1                                        # tie between shards 1 and 2: the lower global index
        assert b[1] == 17 and b[3] == [600]                     # tie between shards 0 and 3; the exact copy (min1 = 0) in shard 2


def test_single_process_path_without_init():
    import oracle_lib as O
    A = H.synth.descriptors(3, 50); B = H.synth.descriptors(4, 60)

    def shard_fn(Aq, Bs, off):
        m1, ix, m2 = O.sift_match_shard(Aq.numpy(), Bs.numpy(), off)
        return torch.from_numpy(m1), torch.from_numpy(ix), torch.from_numpy(m2)
    res = parallel.match_sharded(torch.from_numpy(A), torch.from_numpy(B), 0, 0.8, shard_fn=shard_fn, merge_fn=_cpu_merge)
    ref, _, _ = O.sift_matches(A, B, 0.8, want_distance=False)
    assert res.numpy().tolist() == ref.tolist()


@pytest.mark.gpu
def test_sharded_match_single_gpu_world1(nm, oracle, cuda):
    A = H.synth.descriptors(5, 700); B = H.synth.descriptors(6, 900)
    res = parallel.match_sharded(torch.from_numpy(A).to(cuda), torch.from_numpy(B).to(cuda), 0, 0.8)
    ref, _, _ = oracle.sift_matches(A, B, 0.8, want_distance=False)
    assert np.array_equal(res.cpu().numpy(), ref)
