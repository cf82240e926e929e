"""Multi-GPU sharding of the hot path (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

The reference is single-GPU (its only device code is CudaUtils::setup_CUDA, utils/cudautils.cpp:19-28); this module is
new. Two workloads shard:

* detect+describe of a batch of frames: frames are independent -> contiguous blocks of frames per rank, NO collective.
* all-pairs matching: the query set A is replicated, the candidate set B is split row-wise. Every rank computes the
  exact (min1, global index, min2) of each query over its shard; ONE all-gather of 12 bytes per row per rank follows;
  every rank merges the triples in ascending shard order so that the lowest global index wins ties (match.cu:94-105).
"""
import torch
import torch.distributed as dist


def block_range(n, world, rank):
    """Contiguous [begin, end) of n items owned by `rank` (first n % world ranks get one extra item)."""
    base, extra = divmod(n, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


def frames_of_rank(n_frames, world, rank):
    b, e = block_range(n_frames, world, rank)
    return list(range(b, e))


def _hip_shard(A, B_shard, offset):
    import niftymatch_amd as nm
    return nm.sift_match_shard(A, B_shard, offset)


def _hip_merge(m1, ix, m2, ambiguity, prior):
    import niftymatch_amd as nm
    return nm.sift_match_merge(m1, ix, m2, ambiguity, prior=prior)


def match_sharded(A, B_shard, index_offset, ambiguity=0.8, prior=None, group=None, shard_fn=None, merge_fn=None):
    """Match every row of A (replicated) against the row-sharded B. Returns int32[nA] match indexes into the global B.

    shard_fn / merge_fn default to the HIP kernels; tests inject CPU implementations to run the collective logic on
    gloo. Exactly one collective (all_gather of a (3, nA) int32 tensor) is issued.
    """
    shard_fn = shard_fn or _hip_shard
    merge_fn = merge_fn or _hip_merge
    m1, ix, m2 = shard_fn(A, B_shard, index_offset)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    packed = torch.stack([m1.view(torch.int32), ix.to(torch.int32), m2.view(torch.int32)]).contiguous()   # (3, nA)
    if world > 1:
        flat = torch.empty((world * 3, packed.shape[1]), dtype=torch.int32, device=packed.device)
        dist.all_gather_into_tensor(flat, packed, group=group)      # rank-major concatenation along dim 0
        gathered = flat.view(world, 3, packed.shape[1])
    else:
        gathered = packed.unsqueeze(0)
    m1_all = gathered[:, 0].contiguous().view(torch.float32)
    ix_all = gathered[:, 1].contiguous()
    m2_all = gathered[:, 2].contiguous().view(torch.float32)
    return merge_fn(m1_all, ix_all, m2_all, ambiguity, prior)
