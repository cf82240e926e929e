"""niftymatch_amd -- MI355X (gfx950) drop-in for NiftyMatch's SIFT detect/describe + brute-force L2 match path.

The product is the HIP library niftymatch_amd/lib/libnm_hip.so (C ABI: include/nm_abi.h) and the C++ headers under
niftymatch_amd/nm/ that mirror the reference's API. This Python module is host plumbing for tests, bench.py and the
multi-GPU launcher: it binds the C ABI with ctypes and uses torch only for device memory, streams and
torch.distributed. There is NO CPU fallback: if the library is missing, every entry point raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libnm_hip.so")
# A/B diagnostics (tools/build_variant.py builds libraries with pieces compiled out: wrong results by design). A stray
# NM_HIP_LIB alone must never redirect the product or its tests, so the override also needs NM_DIAGNOSTIC=1.
if os.environ.get("NM_HIP_LIB"):
    if os.environ.get("NM_DIAGNOSTIC") != "1":
        raise RuntimeError("NM_HIP_LIB is set without NM_DIAGNOSTIC=1: refusing to load a diagnostic library build")
    LIB_PATH = os.environ["NM_HIP_LIB"]
_lib = None

_F = C.c_float
_I = C.c_int
_P = C.c_void_p
_SZ = C.c_size_t

_SIGNATURES = {
    "DivUp": (_I, [_I, _I]), "DivDown": (_I, [_I, _I]), "AlignUp": (_I, [_I, _I]), "AlignDown": (_I, [_I, _I]),
    "nm_version": (C.c_char_p, []), "nm_device_count": (_I, [_P]), "nm_set_device": (_I, [_I]),
    "nm_error_string": (C.c_char_p, [_I]),
    "nm_fill_u32": (_I, [_P, _SZ, C.c_uint, _P]),
    "nm_profile_events": (_I, [_I, _P, _P]),
    "nm_create_kernel_for_sigma": (_I, [_F, _P]),
    "nm_convolve_f32": (_I, [_P, _P, _P, _I, _I, _P, _I, _P]),
    "nm_downsample2_f32": (_I, [_P, _I, _I, _P, _I, _I, _P]),
    "nm_subtract_f32": (_I, [_P, _P, _P, _I, _I, _P]),
    "nm_gradient_f32": (_I, [_P, _P, _I, _I, _P]),
    "nm_find_keypoints_f32": (_I, [_P, _P, _P, _I, _I, _F, _F, _F, _F, _I, _I, _P, _P]),
    "nm_find_keypoints_masked_f32": (_I, [_P, _P, _I, _I, _P, _P, _I, _I, _F, _F, _F, _F, _I, _I, _P, _P]),
    "nm_subtract_batch_f32": (_I, [_I, _P, _P, _P, _I, _I, _P]),
    "nm_gradient_batch_f32": (_I, [_I, _P, _P, _I, _I, _P]),
    "nm_find_keypoints3_f32": (_I, [_P, _P, _I, _I, _I, _I, _F, _F, _F, _F, _I, _P, _P]),
    "nm_find_keypoints3_reset_f32": (_I, [_P, _P, _I, _I, _I, _I, _F, _F, _F, _F, _I, _P, _P, _P]),
    "nm_find_keypoints3_compact_workspace_bytes": (_SZ, [_I, _I]),
    "nm_find_keypoints3_compact_f32": (_I, [_P, _I, _I, _F, _F, _F, _F, _I, _I, _P, _P, _P, _P]),
    "nm_compact3_workspace_bytes": (_SZ, [_I]),
    "nm_compact_keypoints3": (_I, [_P, _I, _P, _P, _P, _P]),
    "nm_detect_orientations_levels": (_I, [_I, _P, _P, _P, _I, _I, _F, _F, _P, _P]),
    "nm_compute_sift_descriptors_levels": (_I, [_I, _P, _P, _P, _P, _I, _I, _I, _F, _P, _P, _P, _P]),
    "nm_detect_orientations_levels_dev": (_I, [_P, _P, _I, _P, _I, _I, _F, _F, _P, _P, _P]),
    "nm_compute_sift_descriptors_levels_dev": (_I, [_P, _P, _P, _I, _P, _I, _I, _P, _P, _P, _I, _I, _I, _F, _P, _P, _P, _P]),
    "nm_compact_workspace_bytes": (_SZ, [_I]),
    "nm_compact_keypoints": (_I, [_P, _I, _P, _P, _P, _P]),
    "nm_detect_orientations": (_I, [_P, _P, _I, _I, _I, _F, _F, _P, _P]),
    "nm_compute_sift_descriptors": (_I, [_P, _P, _P, _I, _I, _I, _I, _F, _P, _P, _P, _P]),
    "nm_transpose_f32": (_I, [_P, _P, _I, _I, _P]),
    "nm_bf_distance_f32": (_I, [_P, _I, _P, _I, _I, _P, _P]),
    "nm_get_sift_matches_f32": (_I, [_P, _I, _I, _I, _P, _F, _P]),
    "nm_sift_match_plan": (_I, [_I, _I, _P]),
    "nm_sift_match_plan_segments": (_I, [_I, _I, _I, _P, _I]),
    "nm_profile_event_pairs": (_I, [_I, _P, _I]),
    "nm_sift_match_batch_workspace_bytes": (_SZ, [_I, _P, _P]),
    "nm_sift_match_batch_f32": (_I, [_I, _P, _P, _P, _P, _P, _F, _P, _P]),
    "nm_sift_match_batch_dev_workspace_bytes": (_SZ, [_I, _I, _I]),
    "nm_sift_match_batch_dev_f32": (_I, [_I, _P, _P, _P, _P, _I, _I, _P, _F, _P, _P]),
    "nm_sift_match_batch_dev_phases_f32": (_I, [_I, _I, _P, _P, _P, _P, _I, _I, _P, _F, _P, _P]),
    "nm_sift_match_workspace_bytes": (_SZ, [_I, _I]),
    "nm_sift_match_set_screen": (_I, [_I]),
    "nm_sift_match_get_screen": (_I, []),
    "nm_sift_match_pairs_per_launch": (_I, [_I]),
    "nm_sift_match_set_distance_mode": (_I, [_I]),
    "nm_sift_set_detect_tall_min": (_I, [_I]),
    "nm_sift_match_get_distance_mode": (_I, []),
    "nm_sift_match_distance_listed": (_I, [_P, _I, _I, _P, _P, _P]),
    "nm_sift_match_f32": (_I, [_P, _I, _P, _I, _P, _P, _F, _P, _P]),
    "nm_sift_match_fallback_count": (_I, [_P, _I, _I, _P, _P]),
    "nm_sift_match_second_pass_count": (_I, [_P, _I, _I, _P, _P]),
    "nm_sift_match_shard_f32": (_I, [_P, _I, _P, _I, _I, _P, _P, _P, _P, _P]),
    "nm_sift_match_merge_f32": (_I, [_P, _P, _P, _I, _I, _P, _F, _P]),
    "nm_sift_match_merge_packed_f32": (_I, [_P, _I, _I, _P, _F, _P]),
    "nm_sift_match_allgather_workspace_bytes": (_SZ, [_I, _I, _I]),
    "nm_sift_match_allgather_f32": (_I, [_P, _I, _P, _I, _I, _I, _P, _F, _P, _P, _P]),
    "nm_grayscale_f32": (_I, [_P, _P, _I, _I, _P]),
    "nm_extract_channel_f32": (_I, [_P, _P, _I, _I, _I, _P]),
    "nm_put_channel_f32": (_I, [_P, _P, _I, _I, _I, _P]),
    "nm_set_alpha_to_const": (_I, [_P, _I, _I, C.c_ubyte, _P]),
    "nm_cast_f32_u8": (_I, [_P, _SZ, _SZ, _P, C.c_ubyte, _P]),
    "nm_downsample2_u8x4": (_I, [_P, _I, _I, _P, _I, _I, _P]),
    "nm_align_points": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "nm_selftest_sqrt": (_I, [_P, _P]),
    "nm_selftest_expw": (_I, [_P, _P]),
    "nm_selftest_orient": (_I, [_P, _P]),
    "nm_selftest_mfma_model": (_I, [_I, _I, _I, _P, _P]),
    "nm_sift_match_accum_budget": (_F, [_I]),
    "nm_selftest_mfma_f32": (_I, [_I, _I, _P, _P]),
    "nm_sift_match_distance_budget": (_F, []),
    "nm_undistort_map_f32": (_I, [_P, _P, _SZ, _SZ, _P, _P, _P, _P, _P]),
    "nm_resample_undistort_f32": (_I, [_P, _I, _I, _I, _P, _P, _SZ, _SZ, _P, _P]),
    "nm_resample_mask_u8": (_I, [_P, _P, _I, _I, _I, _I, _I, _P, _P, _F, _P]),
    "nm_resample_perspective_u8x4": (_I, [_P, _P, _I, _I, _I, _I, _P, _P, _P, _I, _P]),
    "nm_transform_blend": (_I, [_P, _I, _I, _P, _I, _I, _I, _I, _P, _I, _I, _P, _I, _P, _P, _I, _P]),
    "nm_ransac_f32": (_I, [_I, _P, _P, _P, _P, _I, _P, _I, _F, _P, _P, _P, _P, _P]),
    "nm_ransac_seed": (None, [C.c_uint]),
    "nm_sift_arena_create": (_I, [_I, _I, _I, _P]),
    "nm_sift_arena_destroy": (None, [_P]),
    "nm_sift_arena_bytes": (_SZ, [_P]),
    "nm_sift_arena_tail_trace": (_I, [_P, _P, _I, _P, _I]),
    "nm_sift_arena_tail_segments": (_I, [_P]),
    "nm_sift_arena_tail_status": (_I, [_P, _P, _P]),
    "nm_sift_arena_tail_inject_error": (_I, [_P]),
    "nm_sift_arena_launches_per_call": (_I, [_P, _I]),
    "nm_sift_tail_plan": (_I, [_I, _I, _I, _P, _I, _P]),
    "nm_sift_arena_set_params": (_I, [_P, _F, _F]),
    "nm_sift_arena_get_params": (_I, [_P, _P, _P]),
    "nm_sift_arena_set_mask": (_I, [_P, _P, _I, _I]),
    "nm_sift_detect_describe": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "nm_sift_detect_describe_batch": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "nm_sift_scale_space_batch": (_I, [_P, _I, _P, _P]),
    "nm_sift_scale_space_batch_ex": (_I, [_P, _I, _P, _I, _P]),
    "nm_sift_arena_level": (_P, [_P, _I]), "nm_sift_arena_dog": (_P, [_P, _I]), "nm_sift_arena_grad": (_P, [_P]),
    "nm_sift_octave_pyramid": (_I, [_P, _I, _I, _P]),
    "nm_client_detect_describe": (_I, [_P, _I, _I, _I, _P, _P, _P]),
    "nm_client_match": (_I, [_P, _I, _P, _I, _P, _P, _F]),
    "nm_client_copy_semantics": (_I, [_P, _I, _I, _I]),
    "nm_client_lazy_counts": (_I, [_P, _I, _I, _I, _P, _I, _P]),
    "nm_client_pair_loop_ex": (C.c_double, [_P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "nm_client_pair_loop": (C.c_double, [_P, _P, _I, _I, _I, _I, _I, _P]),
    "nm_client_ransac": (_I, [_I, _P, _P, _P, _P, _I, _F, _I, C.c_uint, _P]),
}

ABI_SYMBOLS = tuple(k for k in _SIGNATURES if not k.startswith("nm_client_"))


class NmError(RuntimeError):
    pass


def lib():
    """Load libnm_hip.so (built by `python -m niftymatch_amd.build`). Fails loudly when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NmError("libnm_hip.so not found at %s: build it with `python -m niftymatch_amd.build` "
                          "(there is no CPU fallback)" % LIB_PATH)
        # torch bundles its own HIP runtime (same SONAME libamdhip64.so.7). Import it FIRST so that the dynamic loader
        # binds libnm_hip.so to that copy: two HIP runtimes in one process cannot both own the device.
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _check(status, what):
    if status != 0:
        raise NmError("%s failed: (%d) %s" % (what, status, lib().nm_error_string(status).decode()))


def _torch():
    import torch
    return torch


def _stream():
    return _torch().cuda.current_stream().cuda_stream


def _dev(t, dtype=None):
    torch = _torch()
    if not t.is_cuda:
        raise NmError("expected a device tensor")
    if dtype is not None and t.dtype != dtype:
        raise NmError("expected dtype %s, got %s" % (dtype, t.dtype))
    if not t.is_contiguous():
        raise NmError("expected a contiguous tensor")
    return t.data_ptr()


PROF_MATCH_TOP2 = 0
PROF_PYRAMID_O0 = 1
PROF_DESCRIBE = 2
PROF_ORIENT = 3
PROF_DETECT_O0 = 4
PROF_DISTANCE = 5


def profile_events(site, start=None, stop=None):
    """Register (torch.cuda.Event, torch.cuda.Event) to be recorded around a kernel site; None clears."""
    _check(lib().nm_profile_events(site, start.cuda_event if start is not None else None,
                                   stop.cuda_event if stop is not None else None), "nm_profile_events")


def profile_event_pairs(site, pairs):
    """pairs: list of (start, stop) torch.cuda.Event, consumed one per launch of the site; [] clears. The returned ctypes
    array must stay alive until the hook is cleared."""
    n = len(pairs)
    arr = (C.c_void_p * (2 * n))(*[e.cuda_event for p in pairs for e in p]) if n else None
    _check(lib().nm_profile_event_pairs(site, arr, n), "nm_profile_event_pairs")
    return arr


def create_kernel_for_sigma(sigma):
    """Host taps of PyramidData::create_kernel_for_sigma -> (numpy float32 taps, radius)."""
    import numpy as np
    r = lib().nm_create_kernel_for_sigma(sigma, None)
    taps = np.zeros(2 * r + 1, np.float32)
    lib().nm_create_kernel_for_sigma(sigma, taps.ctypes.data)
    return taps, r


# ---- stage wrappers (device tensors in, device tensors out) ----------------------------------------------------
def convolve(image, taps_dev, radius, want_buffer=False):
    torch = _torch()
    h, w = image.shape
    out = torch.empty_like(image)
    buf = torch.empty_like(image)
    _check(lib().nm_convolve_f32(_dev(out), _dev(image, torch.float32), _dev(buf), w, h, _dev(taps_dev), radius,
                                 _stream()), "nm_convolve_f32")
    return (out, buf) if want_buffer else out


def downsample2(src, rw, rh):
    torch = _torch()
    sh, sw = src.shape
    out = torch.empty((rh, rw), dtype=torch.float32, device=src.device)
    _check(lib().nm_downsample2_f32(_dev(out), rw, rh, _dev(src, torch.float32), sw, sh, _stream()), "nm_downsample2_f32")
    return out


def subtract(a, b):
    torch = _torch()
    out = torch.empty_like(a)
    _check(lib().nm_subtract_f32(_dev(a, torch.float32), _dev(b, torch.float32), _dev(out), a.shape[1], a.shape[0],
                                 _stream()), "nm_subtract_f32")
    return out


def gradient(src):
    torch = _torch()
    h, w = src.shape
    out = torch.empty((h, w, 2), dtype=torch.float32, device=src.device)
    _check(lib().nm_gradient_f32(_dev(src, torch.float32), _dev(out), w, h, _stream()), "nm_gradient_f32")
    return out


def find_keypoints(cur, dn, up, peak, edge, xper, sigma0, num_dogs, level, mask=None):
    torch = _torch()
    h, w = cur.shape
    res = torch.full((h, w, 4), -1.0, dtype=torch.float32, device=cur.device)
    if mask is None:
        _check(lib().nm_find_keypoints_f32(_dev(cur), _dev(dn), _dev(up), w, h, peak, edge, xper, sigma0, num_dogs,
                                           level, _dev(res), _stream()), "nm_find_keypoints_f32")
    else:
        mh, mw = mask.shape
        _check(lib().nm_find_keypoints_masked_f32(_dev(cur), _dev(mask, torch.float32), mw, mh, _dev(dn), _dev(up), w,
                                                  h, peak, edge, xper, sigma0, num_dogs, level, _dev(res), _stream()),
               "nm_find_keypoints_masked_f32")
    return res


def compact_keypoints(dense):
    torch = _torch()
    flat = dense.reshape(-1, 4)
    n = flat.shape[0]
    out = torch.full_like(flat, -1.0)
    cnt = torch.zeros(1, dtype=torch.int32, device=dense.device)
    ws = torch.empty(lib().nm_compact_workspace_bytes(n) + 16, dtype=torch.uint8, device=dense.device)
    _check(lib().nm_compact_keypoints(_dev(flat), n, _dev(out), _dev(cnt), _dev(ws), _stream()), "nm_compact_keypoints")
    return out[: int(cnt.item())]


def detect_orientations(kpts, grad, ow, oh, gauss_factor, xper):
    torch = _torch()
    n = kpts.shape[0]
    res = torch.full((n, 2), -1.0, dtype=torch.float32, device=kpts.device)
    _check(lib().nm_detect_orientations(_dev(kpts, torch.float32), _dev(grad, torch.float32), n, ow, oh, gauss_factor,
                                        xper, _dev(res), _stream()), "nm_detect_orientations")
    return res


def compute_sift_descriptors(kpts, orients, grad, ow, oh, num_dogs, xper):
    torch = _torch()
    n = kpts.shape[0]
    desc = torch.zeros((n, 128), dtype=torch.float32, device=kpts.device)
    x = torch.zeros(n, dtype=torch.float32, device=kpts.device)
    y = torch.zeros(n, dtype=torch.float32, device=kpts.device)
    _check(lib().nm_compute_sift_descriptors(_dev(kpts), _dev(orients), _dev(grad), n, ow, oh, num_dogs, xper,
                                             _dev(desc), _dev(x), _dev(y), _stream()), "nm_compute_sift_descriptors")
    return desc, x, y


def transpose(a):
    torch = _torch()
    h, w = a.shape
    out = torch.empty((w, h), dtype=torch.float32, device=a.device)
    _check(lib().nm_transpose_f32(_dev(out), _dev(a, torch.float32), w, h, _stream()), "nm_transpose_f32")
    return out


def bf_distance(At, B):
    torch = _torch()
    dim, na = At.shape
    nb = B.shape[0]
    D = torch.empty((nb, na), dtype=torch.float32, device=At.device)
    _check(lib().nm_bf_distance_f32(_dev(At, torch.float32), na, _dev(B, torch.float32), nb, dim, _dev(D), _stream()),
           "nm_bf_distance_f32")
    return D


def get_sift_matches(distance, ambiguity=0.8, prior=None, cols=None):
    torch = _torch()
    rows, bw = distance.shape
    cols = bw if cols is None else cols
    res = torch.full((rows,), -1, dtype=torch.int32, device=distance.device) if prior is None else prior.clone()
    _check(lib().nm_get_sift_matches_f32(_dev(distance, torch.float32), rows, cols, bw, _dev(res, torch.int32),
                                         ambiguity, _stream()), "nm_get_sift_matches_f32")
    return res


class MatchWorkspace:
    """Device scratch of the fused matcher, sized for (nA, nB); reusable across calls of at most that size."""

    def __init__(self, nA, nB, device):
        torch = _torch()
        self.nA, self.nB = nA, nB
        self.buf = torch.empty(lib().nm_sift_match_workspace_bytes(nA, nB) + 256, dtype=torch.uint8, device=device)


def sift_match(A, B, ambiguity=0.8, want_distance=False, prior=None, workspace=None, nA=None, nB=None):
    """compute_sift_matches on raw descriptor tensors (n x 128). Returns (result int32[nA], distance or None)."""
    torch = _torch()
    nA = A.shape[0] if nA is None else nA
    nB = B.shape[0] if nB is None else nB
    ws = workspace or MatchWorkspace(nA, nB, A.device)
    D = torch.empty((nA, nB), dtype=torch.float32, device=A.device) if want_distance else None
    res = torch.full((nA,), -1, dtype=torch.int32, device=A.device) if prior is None else prior
    _check(lib().nm_sift_match_f32(_dev(A, torch.float32), nA, _dev(B, torch.float32), nB,
                                   _dev(D) if D is not None else None, _dev(res, torch.int32), ambiguity,
                                   _dev(ws.buf), _stream()), "nm_sift_match_f32")
    return res, D


MATCH_MAX_BATCH = 16


class MatchBatchWorkspace:
    """Device scratch for nm_sift_match_batch_f32 with at most n pairs of at most (nA, nB) rows."""

    def __init__(self, n, nA, nB, device):
        torch = _torch()
        self.n, self.nA, self.nB = n, nA, nB
        self.buf = torch.empty(n * lib().nm_sift_match_workspace_bytes(nA, nB), dtype=torch.uint8, device=device)


def sift_match_batch(As, Bs, nAs, nBs, results, ambiguity=0.8, workspace=None):
    """len(As) <= MATCH_MAX_BATCH matches in one call; results[k] (int32, >= nAs[k]) is updated in place like the
    `prior` of sift_match."""
    torch = _torch()
    n = len(As)
    if not (n == len(Bs) == len(nAs) == len(nBs) == len(results)) or not 0 < n <= MATCH_MAX_BATCH:
        raise NmError("bad batch")
    ia = (C.c_int * n)(*nAs)
    ib = (C.c_int * n)(*nBs)
    need = lib().nm_sift_match_batch_workspace_bytes(n, ia, ib)
    if workspace is None:
        workspace = MatchBatchWorkspace(n, max(nAs), max(nBs), As[0].device)
    if workspace.buf.numel() < need:
        raise NmError("batch workspace too small")
    arr = lambda vals: (C.c_void_p * n)(*vals)
    _check(lib().nm_sift_match_batch_f32(n, arr([_dev(a, torch.float32) for a in As]), ia,
                                         arr([_dev(b, torch.float32) for b in Bs]), ib,
                                         arr([_dev(r, torch.int32) for r in results]), ambiguity, _dev(workspace.buf),
                                         _stream()), "nm_sift_match_batch_f32")
    return workspace


class MatchBatchDevWorkspace:
    """Device scratch for nm_sift_match_batch_dev_f32: n pairs of at most (capA, capB) rows, sizes read on the device."""

    def __init__(self, n, capA, capB, device):
        torch = _torch()
        self.n, self.capA, self.capB = n, capA, capB
        self.pair_bytes = lib().nm_sift_match_batch_dev_workspace_bytes(1, capA, capB)
        self.buf = torch.empty(n * self.pair_bytes, dtype=torch.uint8, device=device)

    def row_counts(self, k):
        """(rows of pair k that the bf16x3 second pass screened again, rows that took the exact fallback) in the last call
        on this workspace; the first is meaningful under the two-stage screen only. Synchronises the stream."""
        class _View:                                     # the pair's slice, shaped like a single-pair workspace
            pass
        v = _View()
        v.buf = self.buf[k * self.pair_bytes:(k + 1) * self.pair_bytes]
        return match_second_pass_count(v, self.capA, self.capB), match_fallback_count(v, self.capA, self.capB)


MATCH_PHASE_PREP, MATCH_PHASE_SCREEN, MATCH_PHASE_FINISH = 1, 2, 4


def sift_match_batch_dev(As, d_nAs, Bs, d_nBs, results, ambiguity=0.8, workspace=None, capA=None, capB=None, phases=7):
    """len(As) <= MATCH_MAX_BATCH matches whose set sizes are int32 DEVICE tensors (e.g. SiftArena.num_items): no host
    read-back between detect and match. capA / capB default to the rows of the descriptor tensors."""
    torch = _torch()
    n = len(As)
    if not (n == len(Bs) == len(d_nAs) == len(d_nBs) == len(results)) or not 0 < n <= MATCH_MAX_BATCH:
        raise NmError("bad batch")
    capA = min(a.shape[0] for a in As) if capA is None else capA
    capB = min(b.shape[0] for b in Bs) if capB is None else capB
    if any(a.shape[0] < capA for a in As) or any(b.shape[0] < capB for b in Bs) or any(r.shape[0] < capA for r in results):
        raise NmError("a descriptor set or result is smaller than the capacity")
    if workspace is None:
        workspace = MatchBatchDevWorkspace(n, capA, capB, As[0].device)
    if workspace.buf.numel() < lib().nm_sift_match_batch_dev_workspace_bytes(n, capA, capB):
        raise NmError("batch workspace too small")
    arr = lambda vals: (C.c_void_p * n)(*vals)
    _check(lib().nm_sift_match_batch_dev_phases_f32(phases, n, arr([_dev(a, torch.float32) for a in As]),
                                                    arr([_dev(c, torch.int32) for c in d_nAs]),
                                                    arr([_dev(b, torch.float32) for b in Bs]),
                                                    arr([_dev(c, torch.int32) for c in d_nBs]), capA, capB,
                                                    arr([_dev(r, torch.int32) for r in results]), ambiguity,
                                                    _dev(workspace.buf), _stream()), "nm_sift_match_batch_dev_phases_f32")
    return workspace


MATCH_SCREENS = {"f32": 0, "bf16x3": 1, "f16": 2}


def set_match_screen(name):
    """Select the MFMA screen of the fused matcher ("f32", "bf16x3" or the two-stage "f16"; results are identical, see
    nm_abi.h)."""
    _check(lib().nm_sift_match_set_screen(MATCH_SCREENS[name]), "nm_sift_match_set_screen")


def get_match_screen():
    v = lib().nm_sift_match_get_screen()
    return [k for k, x in MATCH_SCREENS.items() if x == v][0]


def match_pairs_per_launch(n_pairs):
    """Pairs one launch of the screening kernel covers in a batched call of n_pairs pairs under the current screen (nm_abi.h)."""
    return int(lib().nm_sift_match_pairs_per_launch(int(n_pairs)))


def set_detect_tall_min(min_groups=-1):
    """Batched detection launches with at least `min_groups` 20-row unit groups take the tall form (default 2048; -1 restores
    it, 2**31 - 1 disables it). Returns the previous value. Results do not depend on it."""
    return lib().nm_sift_set_detect_tall_min(int(min_groups))


DISTANCE_MODES = {"exact": 0, "mfma": 1}


def set_distance_mode(name):
    """How sift_match(..., want_distance=True) fills the matrix: "mfma" (default: fp32 matrix cores, every entry within 1e-4
    relative of the reference's chain, near-duplicates recomputed exactly) or "exact" (VALU kernel, bit-equal)."""
    _check(lib().nm_sift_match_set_distance_mode(DISTANCE_MODES[name]), "nm_sift_match_set_distance_mode")


def get_distance_mode():
    v = lib().nm_sift_match_get_distance_mode()
    return [k for k, x in DISTANCE_MODES.items() if x == v][0]


def match_distance_listed(workspace, nA, nB):
    """(32 x 32 blocks of the matrix the last MFMA distance pass on `workspace` listed for re-examination, capacity of its
    list); (-1, 0) when the pass took the exact kernel for lack of scratch."""
    n, cap = C.c_int(0), C.c_int(0)
    _check(lib().nm_sift_match_distance_listed(_dev(workspace.buf), nA, nB, C.byref(n), C.byref(cap), _stream()),
           "nm_sift_match_distance_listed")
    return n.value, cap.value


def match_fallback_count(workspace, nA, nB):
    """Rows of the last match call on `workspace` (same sizes) that needed the exact full-scan fallback."""
    n = C.c_int(0)
    _check(lib().nm_sift_match_fallback_count(_dev(workspace.buf), nA, nB, C.byref(n), _stream()),
           "nm_sift_match_fallback_count")
    return n.value


def match_second_pass_count(workspace, nA, nB):
    """Two-stage screen: rows of the last match call on `workspace` (same sizes) that the bf16x3 pass screened again."""
    n = C.c_int(0)
    _check(lib().nm_sift_match_second_pass_count(_dev(workspace.buf), nA, nB, C.byref(n), _stream()),
           "nm_sift_match_second_pass_count")
    return n.value


def sift_match_shard(A, B_shard, index_offset, workspace=None):
    """Exact (min1, idx+offset, min2) of every row of A over the local shard of B."""
    torch = _torch()
    nA, nB = A.shape[0], B_shard.shape[0]
    ws = workspace or MatchWorkspace(nA, nB, A.device)
    m1 = torch.empty(nA, dtype=torch.float32, device=A.device)
    ix = torch.empty(nA, dtype=torch.int32, device=A.device)
    m2 = torch.empty(nA, dtype=torch.float32, device=A.device)
    _check(lib().nm_sift_match_shard_f32(_dev(A, torch.float32), nA, _dev(B_shard, torch.float32), nB, index_offset,
                                         _dev(m1), _dev(ix), _dev(m2), _dev(ws.buf), _stream()),
           "nm_sift_match_shard_f32")
    return m1, ix, m2


def sift_match_merge(m1_all, ix_all, m2_all, ambiguity=0.8, prior=None):
    """Merge shard-major (n_shards, nA) triples into match indexes."""
    torch = _torch()
    n_shards, nA = m1_all.shape
    res = torch.full((nA,), -1, dtype=torch.int32, device=m1_all.device) if prior is None else prior
    _check(lib().nm_sift_match_merge_f32(_dev(m1_all, torch.float32), _dev(ix_all, torch.int32),
                                         _dev(m2_all, torch.float32), n_shards, nA, _dev(res), ambiguity, _stream()),
           "nm_sift_match_merge_f32")
    return res


# ---- element-wise stages either side of the path --------------------------------------------
def grayscale(bgra):
    """uint8 (H, W, 4) BGRA -> float32 (H, W): 0.07 B + 0.72 G + 0.21 R."""
    torch = _torch()
    h, w, _ = bgra.shape
    out = torch.empty((h, w), dtype=torch.float32, device=bgra.device)
    _check(lib().nm_grayscale_f32(_dev(bgra, torch.uint8), _dev(out), w, h, _stream()), "nm_grayscale_f32")
    return out


def extract_channel(bgra, channel):
    torch = _torch()
    h, w, _ = bgra.shape
    out = torch.full((h, w), -7.0, dtype=torch.float32, device=bgra.device)
    _check(lib().nm_extract_channel_f32(_dev(bgra, torch.uint8), _dev(out), w, h, channel, _stream()),
           "nm_extract_channel_f32")
    return out


def put_channel(bgra, plane, channel):
    torch = _torch()
    out = bgra.clone()
    h, w, _ = out.shape
    _check(lib().nm_put_channel_f32(_dev(out, torch.uint8), _dev(plane, torch.float32), w, h, channel, _stream()),
           "nm_put_channel_f32")
    return out


def set_alpha(bgra, val=255):
    torch = _torch()
    out = bgra.clone()
    h, w, _ = out.shape
    _check(lib().nm_set_alpha_to_const(_dev(out, torch.uint8), w, h, val, _stream()), "nm_set_alpha_to_const")
    return out


def cast_f32_u8(src, max_val=0):
    torch = _torch()
    h, w = src.shape
    out = torch.empty((h, w), dtype=torch.uint8, device=src.device)
    _check(lib().nm_cast_f32_u8(_dev(src, torch.float32), w, h, _dev(out), max_val, _stream()), "nm_cast_f32_u8")
    return out


def downsample2_u8x4(src, rw, rh):
    torch = _torch()
    sh, sw, _ = src.shape
    out = torch.empty((rh, rw, 4), dtype=torch.uint8, device=src.device)
    _check(lib().nm_downsample2_u8x4(_dev(out), rw, rh, _dev(src, torch.uint8), sw, sh, _stream()), "nm_downsample2_u8x4")
    return out


def align_points(sx, sy, dx, dy, matches):
    torch = _torch()
    n = matches.shape[0]
    outs = [torch.empty(n, dtype=torch.float32, device=matches.device) for _ in range(4)]
    _check(lib().nm_align_points(_dev(sx), _dev(sy), _dev(dx), _dev(dy), *[_dev(o) for o in outs],
                                 _dev(matches, torch.int32), n, _stream()), "nm_align_points")
    return outs


def selftest_sqrt():
    """Number of inputs for which the gradient's fast sqrt differs from IEEE sqrt over its whole domain (must be 0)."""
    torch = _torch()
    out = torch.zeros(1, dtype=torch.int64, device="cuda")
    _check(lib().nm_selftest_sqrt(_dev(out), _stream()), "nm_selftest_sqrt")
    return int(out.item())


def selftest_expw():
    """nm_selftest_expw: (unreported differences, inputs reporting a nearby rounding boundary, inputs tested) of the descriptor
    weight's fast form against the spec sequence over its whole domain. The first must be 0."""
    torch = _torch()
    out = torch.zeros(3, dtype=torch.int64, device="cuda")
    _check(lib().nm_selftest_expw(_dev(out), _stream()), "nm_selftest_expw")
    return tuple(int(v) for v in out.cpu())


def selftest_orient():
    """nm_selftest_orient: (differing thirds, inputs the third's guard rejects, differing window tests, differing quotients,
    quotients tested) of the orientation kernel's hoisted arithmetic against the expressions it replaces. [0], [2], [3] must be 0."""
    torch = _torch()
    out = torch.zeros(5, dtype=torch.int64, device="cuda")
    _check(lib().nm_selftest_orient(_dev(out), _stream()), "nm_selftest_orient")
    return tuple(int(v) for v in out.cpu())


MFMA_BF16, MFMA_F16 = 0, 1


def selftest_mfma_model(instruction, n_random=1 << 20, n_chains=4096):
    """nm_selftest_mfma_model: the rounding model of the matrix instruction the matcher's screens issue, measured on this
    device (layout probe, directed cases, random instructions, the screens' own accumulator chains on adversarial rows
    incl. fp16 subnormals, all against binary64). Returns a dict of the NM_SELFTEST_MFMA_OUTPUTS figures."""
    torch = _torch()
    out = torch.zeros(16, dtype=torch.float32, device="cuda")
    _check(lib().nm_selftest_mfma_model(int(instruction), int(n_random), int(n_chains), _dev(out), _stream()),
           "nm_selftest_mfma_model")
    v = [float(x) for x in out.cpu()]
    return {"layout_mismatches": v[0], "c1_plus_16_small_ulp": v[1], "c1_plus_one_small_ulp": v[2],
            "one_plus_15_small_ulp": v[3], "c2p24_plus_16": v[4], "rel_u": v[5], "model_ratio": v[6],
            "instructions": v[7], "chain_coeff": v[8], "chain_coeff_subnormal": v[9], "chain_launches": v[10],
            "same_half_truncation_ulp": v[11]}


def selftest_mfma_f32(n_random=1 << 22, n_chains=2048):
    """nm_selftest_mfma_f32: rounding of v_mfma_f32_32x32x2_f32 on this device (single instructions and the two accumulation
    forms the library issues), against binary64."""
    torch = _torch()
    out = torch.zeros(8, dtype=torch.float32, device="cuda")
    _check(lib().nm_selftest_mfma_f32(int(n_random), int(n_chains), _dev(out), _stream()), "nm_selftest_mfma_f32")
    v = [float(x) for x in out.cpu()]
    return {"rel_u": v[0], "frac_fma_chain": v[1] / max(v[5], 1.0), "frac_correctly_rounded": v[2] / max(v[5], 1.0),
            "two_chain_coeff": v[3], "one_chain_coeff": v[4], "results": v[5]}


def match_distance_budget():
    return float(lib().nm_sift_match_distance_budget())


def match_accum_budget(screen):
    return float(lib().nm_sift_match_accum_budget(int(screen)))


TEX_U8N, TEX_U8X4N, TEX_F32 = 0, 1, 2


def _tex_format(t):
    torch = _torch()
    if t.dtype == torch.float32 and t.dim() == 2:
        return TEX_F32
    if t.dtype == torch.uint8 and t.dim() == 2:
        return TEX_U8N
    raise NmError("a scalar texture must be a 2-D float32 or uint8 device tensor")


def undistort_map(x, y, camera_matrix, distortion_coeffs):
    """cuda_undistort: per-pixel source coordinates (u, v) of the radial model. camera_matrix = (fx, fy, cx, cy) and
    distortion_coeffs = (k1, k2, k3) are device tensors, as in the reference."""
    torch = _torch()
    h, w = x.shape
    u, v = torch.empty_like(x), torch.empty_like(y)
    _check(lib().nm_undistort_map_f32(_dev(x, torch.float32), _dev(y, torch.float32), w, h,
                                      _dev(camera_matrix, torch.float32), _dev(distortion_coeffs, torch.float32), _dev(u),
                                      _dev(v), _stream()), "nm_undistort_map_f32")
    return u, v


def resample_undistort(tex, x, y):
    torch = _torch()
    h, w = x.shape
    out = torch.empty((h, w), dtype=torch.float32, device=x.device)
    _check(lib().nm_resample_undistort_f32(_dev(tex), tex.shape[1], tex.shape[0], _tex_format(tex), _dev(x, torch.float32),
                                           _dev(y, torch.float32), w, h, _dev(out), _stream()), "nm_resample_undistort_f32")
    return out


def resample_mask(tex, x_pos, y_pos, threshold=0.5):
    torch = _torch()
    h, w = x_pos.shape
    out = torch.empty((h, w), dtype=torch.uint8, device=x_pos.device)
    _check(lib().nm_resample_mask_u8(_dev(out), _dev(tex), tex.shape[1], tex.shape[0], _tex_format(tex), w, h,
                                     _dev(x_pos, torch.float32), _dev(y_pos, torch.float32), threshold, _stream()),
           "nm_resample_mask_u8")
    return out


def resample_perspective(tex_bgra, cols, rows, mat3x3, inverse=True):
    """resample_perspective_transform: returns (result (rows, cols, 4) uint8, x_pos, y_pos)."""
    torch = _torch()
    out = torch.empty((rows, cols, 4), dtype=torch.uint8, device=tex_bgra.device)
    xp = torch.empty((rows, cols), dtype=torch.float32, device=tex_bgra.device)
    yp = torch.empty_like(xp)
    _check(lib().nm_resample_perspective_u8x4(_dev(out), _dev(tex_bgra, torch.uint8), tex_bgra.shape[1], tex_bgra.shape[0],
                                              cols, rows, _dev(xp), _dev(yp), _dev(mat3x3, torch.float32),
                                              1 if inverse else 0, _stream()), "nm_resample_perspective_u8x4")
    return out, xp, yp


def transform_blend(canvas, canvas_wts, frame, nw, nh, mat3x3, tx, ty, frame_mask, frame_wts):
    """transform_blend: blends the warped frame into `canvas` / `canvas_wts` IN PLACE."""
    torch = _torch()
    ch, cw, _ = canvas.shape
    fh, fw, _ = frame.shape
    _check(lib().nm_transform_blend(_dev(canvas, torch.uint8), cw, ch, _dev(frame, torch.uint8), fw, fh, nw, nh,
                                    _dev(mat3x3, torch.float32), tx, ty, _dev(frame_mask), _tex_format(frame_mask),
                                    _dev(canvas_wts, torch.float32), _dev(frame_wts), _tex_format(frame_wts), _stream()),
           "nm_transform_blend")


def ransac(model, sx, sy, dx, dy, rand_list, thr):
    """Evaluate RANSAC hypotheses on the device. model 0/1/2 = translation/similarity/homography; rand_list int32
    (iterations, samples). Returns (position, H_best[9], homographies[iterations, 9], inliers[iterations])."""
    torch = _torch()
    it = rand_list.shape[0]
    H_all = torch.zeros((it, 9), dtype=torch.float32, device=sx.device)
    inl = torch.zeros(it, dtype=torch.int32, device=sx.device)
    Hb = torch.zeros(9, dtype=torch.float32, device=sx.device)
    pos = torch.zeros(1, dtype=torch.int32, device=sx.device)
    _check(lib().nm_ransac_f32(model, _dev(sx, torch.float32), _dev(sy), _dev(dx), _dev(dy), sx.shape[0],
                               _dev(rand_list, torch.int32), it, thr, _dev(H_all), _dev(inl), _dev(Hb), _dev(pos),
                               _stream()), "nm_ransac_f32")
    return pos, Hb, H_all, inl


SIFT_MAX_BATCH = 64


def detect_describe_batch(arenas, grays):
    """Enqueue len(arenas) <= SIFT_MAX_BATCH equally sized frames as ONE launch sequence on the current stream
    (nm_sift_detect_describe_batch); outputs land in each arena's own tensors, exactly as detect_describe would."""
    torch = _torch()
    n = len(arenas)
    if n != len(grays) or not 0 < n <= SIFT_MAX_BATCH:
        raise NmError("batch of %d arenas / %d frames (max %d)" % (n, len(grays), SIFT_MAX_BATCH))
    for a, g in zip(arenas, grays):
        if tuple(g.shape) != (a.height, a.width):
            raise NmError("frame shape %s does not match the arena (%d,%d)" % (tuple(g.shape), a.height, a.width))
        if g.device != a.device or torch.cuda.current_device() != a.device.index:
            raise NmError("arena lives on %s: frame on %s, current device %d" % (a.device, g.device,
                                                                                torch.cuda.current_device()))

    def arr(vals):
        return (C.c_void_p * n)(*vals)
    _check(lib().nm_sift_detect_describe_batch(
        arr([a._h.value for a in arenas]), n, arr([_dev(g, torch.float32) for g in grays]),
        arr([_dev(a.desc) for a in arenas]), arr([_dev(a.x) for a in arenas]), arr([_dev(a.y) for a in arenas]),
        arr([_dev(a.kpts) for a in arenas]), arr([_dev(a.orients) for a in arenas]),
        arr([_dev(a.num_items) for a in arenas]), _stream()), "nm_sift_detect_describe_batch")


def scale_space_batch(arenas, grays, write_dog=True, write_grad=True):
    """Only the scale-space launches (nm_sift_scale_space_batch_ex), on the current stream. write_dog=True: levels + DoG +
    gradient planes; False: what detect_describe_batch runs (no DoG planes). write_grad=False leaves the fused gradient
    planes out: with write_dog=True that is exactly the reference's convolve + compute_dog work (108 B per octave-pixel)."""
    torch = _torch()
    n = len(arenas)
    if n != len(grays) or not 0 < n <= SIFT_MAX_BATCH:
        raise NmError("batch of %d arenas / %d frames (max %d)" % (n, len(grays), SIFT_MAX_BATCH))
    _check(lib().nm_sift_scale_space_batch_ex((C.c_void_p * n)(*[a._h.value for a in arenas]), n,
                                              (C.c_void_p * n)(*[_dev(g, torch.float32) for g in grays]),
                                              (1 if write_dog else 0) | (0 if write_grad else 2), _stream()),
           "nm_sift_scale_space_batch_ex")


class SiftArena:
    """Per-stream frame arena + outputs of nm_sift_detect_describe (replaces PyramidData + SiftData)."""

    def __init__(self, width, height, capacity=16384, device="cuda", num_items=None):
        """num_items: optional 1-element int32 device tensor (e.g. a view into a table of all arenas' counts) that
        receives the descriptor count instead of a tensor of the arena's own."""
        torch = _torch()
        self.width, self.height, self.capacity = width, height, capacity
        self._h = C.c_void_p()
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        # the native arena (buffers, side stream, events) is created on the CURRENT HIP device: make that `device`,
        # where the output tensors live; the launchers refuse an arena of another device (hipErrorInvalidDevice)
        with torch.cuda.device(self.device):
            _check(lib().nm_sift_arena_create(width, height, capacity, C.byref(self._h)), "nm_sift_arena_create")
        device = self.device
        self.desc = torch.zeros((capacity, 128), dtype=torch.float32, device=device)
        self.x = torch.zeros(capacity, dtype=torch.float32, device=device)
        self.y = torch.zeros(capacity, dtype=torch.float32, device=device)
        self.kpts = torch.zeros((capacity, 4), dtype=torch.float32, device=device)
        self.orients = torch.zeros((capacity, 2), dtype=torch.float32, device=device)
        if num_items is not None and (num_items.dtype != torch.int32 or num_items.numel() != 1 or num_items.device != device):
            raise NmError("num_items must be a 1-element int32 tensor on %s" % device)
        self.num_items = torch.zeros(1, dtype=torch.int32, device=device) if num_items is None else num_items

    @property
    def bytes(self):
        return lib().nm_sift_arena_bytes(self._h)

    def detect_describe(self, gray):
        """Enqueue one frame (device fp32 (H,W)) on the current stream; no host synchronisation."""
        torch = _torch()
        if tuple(gray.shape) != (self.height, self.width):
            raise NmError("frame shape %s does not match the arena (%d,%d)" % (tuple(gray.shape), self.height, self.width))
        if gray.device != self.device or torch.cuda.current_device() != self.device.index:
            raise NmError("arena lives on %s: frame on %s, current device %d" % (self.device, gray.device,
                                                                                torch.cuda.current_device()))
        _check(lib().nm_sift_detect_describe(self._h, _dev(gray, torch.float32), _dev(self.desc), _dev(self.x),
                                             _dev(self.y), _dev(self.kpts), _dev(self.orients), _dev(self.num_items),
                                             _stream()), "nm_sift_detect_describe")

    def set_params(self, peak_threshold=0.0, edge_threshold=10.0):
        """SiftParams::_peak_threshold / _edge_threshold for the calls enqueued from now on."""
        _check(lib().nm_sift_arena_set_params(self._h, peak_threshold, edge_threshold), "nm_sift_arena_set_params")

    def set_mask(self, mask=None):
        """Full-resolution float32 device mask (height, width) as in compute_keypoints_with_mask, or None. The arena
        keeps a reference to the tensor."""
        torch = _torch()
        if mask is not None and (tuple(mask.shape) != (self.height, self.width) or mask.device != self.device):
            raise NmError("mask must be a (%d, %d) tensor on %s" % (self.height, self.width, self.device))
        self._mask = mask
        _check(lib().nm_sift_arena_set_mask(self._h, _dev(mask, torch.float32) if mask is not None else None,
                                            self.width, self.height), "nm_sift_arena_set_mask")

    def tail_status(self):
        """0 / 1: whether the last octave-tail launch on this arena's state words timed out (synchronises the stream)."""
        v = C.c_int(0)
        _check(lib().nm_sift_arena_tail_status(self._h, C.byref(v), _stream()), "nm_sift_arena_tail_status")
        return v.value

    def tail_inject_error(self):
        _check(lib().nm_sift_arena_tail_inject_error(self._h), "nm_sift_arena_tail_inject_error")

    def octave_pyramid(self, ow, oh):
        _check(lib().nm_sift_octave_pyramid(self._h, ow, oh, _stream()), "nm_sift_octave_pyramid")

    def level_ptr(self, l):
        return lib().nm_sift_arena_level(self._h, l)

    def dog_ptr(self, d):
        return lib().nm_sift_arena_dog(self._h, d)

    def grad_ptr(self):
        return lib().nm_sift_arena_grad(self._h)

    def close(self):
        if self._h:
            lib().nm_sift_arena_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
