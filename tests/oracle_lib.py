"""ctypes binding of the CPU oracle (oracle/libnm_oracle.so). Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB_PATH = os.path.join(_ORACLE_DIR, "libnm_oracle.so")


class Params(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("num_octaves", C.c_int), ("num_dog_levels", C.c_int),
                ("level_max", C.c_int), ("level_min", C.c_int), ("sigma_d_0", C.c_float), ("sigma_k", C.c_float),
                ("sigma_0", C.c_float), ("sigma_n", C.c_float), ("base_smooth", C.c_float),
                ("peak_threshold", C.c_float), ("edge_threshold", C.c_float), ("sigmas", C.c_float * 8),
                ("num_sigmas", C.c_int)]


def build():
    src = [os.path.join(_ORACLE_DIR, f) for f in ("nm_oracle.cpp", "nmo_math.h", "nmo_ransac.h", "nmo_warp.h", "Makefile")]
    if (not os.path.exists(_LIB_PATH)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src):
        subprocess.check_call(["make", "-C", _ORACLE_DIR], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def set_threads(n):
    return lib().nmo_set_threads(C.c_int(n))


def sift_params(w, h):
    p = Params()
    lib().nmo_sift_params(C.c_int(w), C.c_int(h), C.byref(p))
    return p


def create_kernel_for_sigma(sigma):
    r = lib().nmo_create_kernel_for_sigma(C.c_float(sigma), None)
    taps = np.zeros(2 * r + 1, np.float32)
    lib().nmo_create_kernel_for_sigma(C.c_float(sigma), _fp(taps))
    return taps, r


def convolve(image, taps, r):
    image = _f32(image)
    h, w = image.shape
    out = np.empty_like(image)
    buf = np.empty_like(image)
    lib().nmo_convolve(_fp(out), _fp(image), _fp(buf), C.c_int(w), C.c_int(h), _fp(_f32(taps)), C.c_int(r))
    return out, buf


def downsample2(src, rw, rh):
    src = _f32(src)
    sh, sw = src.shape
    out = np.empty((rh, rw), np.float32)
    lib().nmo_downsample2(_fp(out), C.c_int(rw), C.c_int(rh), _fp(src), C.c_int(sw), C.c_int(sh))
    return out


def subtract(a, b):
    a, b = _f32(a), _f32(b)
    out = np.empty_like(a)
    lib().nmo_subtract(_fp(a), _fp(b), _fp(out), C.c_int(a.shape[1]), C.c_int(a.shape[0]))
    return out


def gradient(src):
    src = _f32(src)
    h, w = src.shape
    out = np.empty((h, w, 2), np.float32)
    lib().nmo_gradient(_fp(src), _fp(out), C.c_int(w), C.c_int(h))
    return out


def find_keypoints(cur, dn, up, peak, edge, xper, sigma0, num_dogs, level, mask=None):
    cur, dn, up = _f32(cur), _f32(dn), _f32(up)
    h, w = cur.shape
    res = np.full((h, w, 4), -1.0, np.float32)
    if mask is not None:
        mask = _f32(mask)
        mh, mw = mask.shape
    else:
        mh = mw = 0
    lib().nmo_find_keypoints(_fp(cur), _fp(dn), _fp(up), _fp(mask), C.c_int(mw), C.c_int(mh), C.c_int(w), C.c_int(h),
                             C.c_float(peak), C.c_float(edge), C.c_float(xper), C.c_float(sigma0), C.c_int(num_dogs),
                             C.c_int(level), _fp(res))
    return res


def compact_keypoints(dense):
    dense = _f32(dense).reshape(-1, 4)
    out = np.full_like(dense, -1.0)
    n = lib().nmo_compact_keypoints(_fp(dense), C.c_int(dense.shape[0]), _fp(out))
    return out[:n].copy()


def detect_orientations(kpts, grad, ow, oh, gauss_factor, xper):
    kpts = _f32(kpts).reshape(-1, 4)
    grad = _f32(grad)
    res = np.full((kpts.shape[0], 2), -1.0, np.float32)
    lib().nmo_detect_orientations(_fp(kpts), _fp(grad), C.c_int(kpts.shape[0]), C.c_int(ow), C.c_int(oh),
                                  C.c_float(gauss_factor), C.c_float(xper), _fp(res))
    return res


def compute_sift_descriptors(kpts, orients, grad, ow, oh, num_dogs, xper):
    kpts = _f32(kpts).reshape(-1, 4)
    orients = _f32(orients).reshape(-1, 2)
    grad = _f32(grad)
    n = kpts.shape[0]
    desc = np.zeros((n, 128), np.float32)
    x = np.zeros(n, np.float32)
    y = np.zeros(n, np.float32)
    lib().nmo_compute_sift_descriptors(_fp(kpts), _fp(orients), _fp(grad), C.c_int(n), C.c_int(ow), C.c_int(oh),
                                       C.c_int(num_dogs), C.c_float(xper), _fp(desc), _fp(x), _fp(y))
    return desc, x, y


def transpose(a):
    a = _f32(a)
    h, w = a.shape
    out = np.empty((w, h), np.float32)
    lib().nmo_transpose(_fp(out), _fp(a), C.c_int(w), C.c_int(h))
    return out


def bf_distance(At, B):
    At, B = _f32(At), _f32(B)
    dim, na = At.shape
    nb = B.shape[0]
    D = np.empty((nb, na), np.float32)
    lib().nmo_bf_distance(_fp(At), C.c_int(na), _fp(B), C.c_int(nb), C.c_int(dim), _fp(D))
    return D


def get_sift_matches(distance, ambiguity=0.8, prior=None, cols=None, buffer_width=None):
    distance = _f32(distance)
    rows = distance.shape[0]
    bw = distance.shape[1] if buffer_width is None else buffer_width
    cols = distance.shape[1] if cols is None else cols
    res = np.full(rows, -1, np.int32) if prior is None else np.ascontiguousarray(prior, np.int32).copy()
    lib().nmo_get_sift_matches(_fp(distance), C.c_int(rows), C.c_int(cols), C.c_int(bw), _fp(res),
                               C.c_float(ambiguity))
    return res


def sift_matches(A, B, ambiguity=0.8, want_distance=True, prior=None):
    A, B = _f32(A), _f32(B)
    na, nb = A.shape[0], B.shape[0]
    D = np.empty((na, nb), np.float32) if want_distance else None
    res = np.full(na, -1, np.int32) if prior is None else np.ascontiguousarray(prior, np.int32).copy()
    m1 = np.empty(na, np.float32)
    ix = np.empty(na, np.int32)
    m2 = np.empty(na, np.float32)
    lib().nmo_sift_matches(_fp(A), C.c_int(na), _fp(B), C.c_int(nb), _fp(D), _fp(res), C.c_float(ambiguity),
                           _fp(m1), _fp(ix), _fp(m2))
    return res, D, (m1, ix, m2)


def sift_match_shard(A, B, index_offset=0):
    """Per-query (min1, global index, min2) of one candidate shard, min2 unclamped (see nmo_sift_match_shard)."""
    A, B = _f32(A), _f32(B)
    na, nb = A.shape[0], B.shape[0]
    m1 = np.empty(na, np.float32)
    ix = np.empty(na, np.int32)
    m2 = np.empty(na, np.float32)
    lib().nmo_sift_match_shard(_fp(A), C.c_int(na), _fp(B), C.c_int(nb), C.c_int(index_offset), _fp(m1), _fp(ix), _fp(m2))
    return m1, ix, m2


def sift_match_merge(m1_all, ix_all, m2_all, ambiguity=0.8, prior=None):
    m1_all, m2_all = _f32(m1_all), _f32(m2_all)
    ix_all = np.ascontiguousarray(ix_all, np.int32)
    n_shards, na = m1_all.shape
    res = np.full(na, -1, np.int32) if prior is None else np.ascontiguousarray(prior, np.int32).copy()
    lib().nmo_sift_match_merge(_fp(m1_all), _fp(ix_all), _fp(m2_all), C.c_int(n_shards), C.c_int(na), _fp(res),
                               C.c_float(ambiguity))
    return res


def sift_detect_describe(gray, capacity=16384, peak=None, edge=None, mask=None):
    """peak / edge: SiftParams::_peak_threshold / _edge_threshold (defaults 0 / 10); mask: full-resolution float plane."""
    gray = _f32(gray)
    h, w = gray.shape
    p = sift_params(w, h)
    if peak is not None or edge is not None or mask is not None:
        return _sift_detect_describe_ex(gray, capacity, p.peak_threshold if peak is None else peak,
                                        p.edge_threshold if edge is None else edge, mask, p)
    desc = np.zeros((capacity, 128), np.float32)
    xs = np.zeros(capacity, np.float32)
    ys = np.zeros(capacity, np.float32)
    kp = np.zeros((capacity, 4), np.float32)
    ori = np.zeros((capacity, 2), np.float32)
    counts = np.zeros(p.num_octaves * 3, np.int32)
    n = lib().nmo_sift_detect_describe(_fp(gray), C.c_int(w), C.c_int(h), C.c_int(capacity), _fp(desc), _fp(xs),
                                       _fp(ys), _fp(kp), _fp(ori), _fp(counts))
    return dict(n=n, desc=desc[:n], x=xs[:n], y=ys[:n], kpts=kp[:n], orient=ori[:n],
                counts=counts.reshape(-1, 3))


def sift_detect_describe_envelope(gray, capacity=16384):
    """nmo_sift_detect_describe_envelope: the frame driver's outputs plus, per output item, the ORDER-FREE binary64 sums and
    vote counts of the descriptor elements and of the raw orientation bins (and the oracle's own raw orientation bins)."""
    gray = _f32(gray)
    h, w = gray.shape
    desc = np.zeros((capacity, 128), np.float32)
    xs, ys = np.zeros(capacity, np.float32), np.zeros(capacity, np.float32)
    kp, ori = np.zeros((capacity, 4), np.float32), np.zeros((capacity, 2), np.float32)
    d64, dnv = np.zeros((capacity, 128), np.float64), np.zeros((capacity, 128), np.int32)
    o32, o64, onv = np.zeros((capacity, 36), np.float32), np.zeros((capacity, 36), np.float64), np.zeros((capacity, 36), np.int32)
    bad = C.c_int(-1)
    n = lib().nmo_sift_detect_describe_envelope(_fp(gray), C.c_int(w), C.c_int(h), C.c_int(capacity), _fp(desc), _fp(xs),
                                                _fp(ys), _fp(kp), _fp(ori), _fp(d64), _fp(dnv), _fp(o32), _fp(o64), _fp(onv),
                                                C.byref(bad))
    return dict(n=n, desc=desc[:n], kpts=kp[:n], orient=ori[:n], desc64=d64[:n], desc_nv=dnv[:n], ohist32=o32[:n],
                ohist64=o64[:n], ohist_nv=onv[:n], bad_bint=bad.value)


def _sift_detect_describe_ex(gray, capacity, peak, edge, mask, p):
    h, w = gray.shape
    desc = np.zeros((capacity, 128), np.float32)
    xs = np.zeros(capacity, np.float32)
    ys = np.zeros(capacity, np.float32)
    kp = np.zeros((capacity, 4), np.float32)
    ori = np.zeros((capacity, 2), np.float32)
    counts = np.zeros(p.num_octaves * 3, np.int32)
    m = None if mask is None else _f32(mask)
    assert m is None or m.shape == (h, w)
    n = lib().nmo_sift_detect_describe_ex(_fp(gray), C.c_int(w), C.c_int(h), C.c_int(capacity), C.c_float(peak),
                                          C.c_float(edge), _fp(m), _fp(desc), _fp(xs), _fp(ys), _fp(kp), _fp(ori),
                                          _fp(counts))
    return dict(n=n, desc=desc[:n], x=xs[:n], y=ys[:n], kpts=kp[:n], orient=ori[:n], counts=counts.reshape(-1, 3))


def octave_pyramid(level0, width, height, want_grad=True):
    level0 = _f32(level0)
    oh, ow = level0.shape
    levels = np.empty((6, oh, ow), np.float32)
    dogs = np.empty((5, oh, ow), np.float32)
    grad = np.empty((3, oh, ow, 2), np.float32) if want_grad else None
    lib().nmo_octave_pyramid(_fp(level0), C.c_int(ow), C.c_int(oh), C.c_int(width), C.c_int(height), _fp(levels),
                             _fp(dogs), _fp(grad))
    return levels, dogs, grad


def vec(fn, *arrs, dtype=np.float32):
    arrs = [np.ascontiguousarray(a, dtype) for a in arrs]
    out = np.empty_like(arrs[0])
    getattr(lib(), "nmo_vec_" + fn)(*[_fp(a) for a in arrs], _fp(out), C.c_int(arrs[0].size))
    return out


# ---- "next" rows: element-wise image stages ----------------------------------------------------------------------
def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def grayscale(bgra):
    bgra = _u8(bgra)
    h, w, _ = bgra.shape
    out = np.empty((h, w), np.float32)
    lib().nmo_grayscale(_fp(bgra), _fp(out), C.c_int(w), C.c_int(h))
    return out


def extract_channel(bgra, channel):
    bgra = _u8(bgra)
    h, w, _ = bgra.shape
    out = np.full((h, w), -7.0, np.float32)
    lib().nmo_extract_channel(_fp(bgra), _fp(out), C.c_int(w), C.c_int(h), C.c_int(channel))
    return out


def put_channel(bgra, plane, channel):
    out = _u8(bgra).copy()
    h, w, _ = out.shape
    lib().nmo_put_channel(_fp(out), _fp(_f32(plane)), C.c_int(w), C.c_int(h), C.c_int(channel))
    return out


def set_alpha(bgra, val):
    out = _u8(bgra).copy()
    h, w, _ = out.shape
    lib().nmo_set_alpha(_fp(out), C.c_int(w), C.c_int(h), C.c_ubyte(val))
    return out


def cast_f32_u8(src, max_val=0):
    src = _f32(src)
    h, w = src.shape
    out = np.empty((h, w), np.uint8)
    lib().nmo_cast_f32_u8(_fp(src), C.c_size_t(w), C.c_size_t(h), _fp(out), C.c_ubyte(max_val))
    return out


def downsample2_u8x4(src, rw, rh):
    src = _u8(src)
    sh, sw, _ = src.shape
    out = np.empty((rh, rw, 4), np.uint8)
    lib().nmo_downsample2_u8x4(_fp(out), C.c_int(rw), C.c_int(rh), _fp(src), C.c_int(sw), C.c_int(sh))
    return out


def align_points(sx, sy, dx, dy, matches):
    sx, sy, dx, dy = _f32(sx), _f32(sy), _f32(dx), _f32(dy)
    matches = np.ascontiguousarray(matches, np.int32)
    n = len(matches)
    outs = [np.empty(n, np.float32) for _ in range(4)]
    lib().nmo_align_points(_fp(sx), _fp(sy), _fp(dx), _fp(dy), *[_fp(o) for o in outs], _fp(matches), C.c_int(n))
    return outs


def ransac(model, sx, sy, dx, dy, rand_list, thr):
    """model: 0 translation, 1 similarity, 2 homography. rand_list: (iterations, samples) int32."""
    sx, sy, dx, dy = _f32(sx), _f32(sy), _f32(dx), _f32(dy)
    rl = np.ascontiguousarray(rand_list, np.int32)
    it = rl.shape[0]
    H_all = np.zeros((it, 9), np.float32)
    inl = np.zeros(it, np.int32)
    Hb = np.zeros(9, np.float32)
    pos = lib().nmo_ransac(C.c_int(model), _fp(sx), _fp(sy), _fp(dx), _fp(dy), C.c_int(len(sx)), _fp(rl), C.c_int(it),
                           C.c_float(thr), _fp(H_all), _fp(inl), _fp(Hb))
    return pos, Hb, H_all, inl


# ---- N3/N4 warps (oracle/nmo_warp.h) ----
TEX_U8N, TEX_U8X4N, TEX_F32 = 0, 1, 2


def _tex(t):
    t = np.ascontiguousarray(t)
    if t.dtype == np.float32 and t.ndim == 2:
        return t, TEX_F32
    if t.dtype == np.uint8 and t.ndim == 2:
        return t, TEX_U8N
    if t.dtype == np.uint8 and t.ndim == 3 and t.shape[2] == 4:
        return t, TEX_U8X4N
    raise ValueError("unsupported texture")


def undistort_map(x, y, cam, dist):
    x, y = _f32(x), _f32(y)
    h, w = x.shape
    u, v = np.empty_like(x), np.empty_like(y)
    lib().nmo_undistort_map(_fp(x), _fp(y), C.c_size_t(w), C.c_size_t(h), _fp(_f32(cam)), _fp(_f32(dist)), _fp(u), _fp(v))
    return u, v


def resample_undistort(tex, x, y):
    tex, fmt = _tex(tex)
    x, y = _f32(x), _f32(y)
    h, w = x.shape
    out = np.empty((h, w), np.float32)
    lib().nmo_resample_undistort(_fp(tex), C.c_int(tex.shape[1]), C.c_int(tex.shape[0]), C.c_int(fmt), _fp(x), _fp(y),
                                 C.c_size_t(w), C.c_size_t(h), _fp(out))
    return out


def resample_mask(tex, x, y, threshold=0.5):
    tex, fmt = _tex(tex)
    x, y = _f32(x), _f32(y)
    h, w = x.shape
    out = np.empty((h, w), np.uint8)
    lib().nmo_resample_mask(_fp(out), _fp(tex), C.c_int(tex.shape[1]), C.c_int(tex.shape[0]), C.c_int(fmt), C.c_int(w),
                            C.c_int(h), _fp(x), _fp(y), C.c_float(threshold))
    return out


def resample_perspective(tex, cols, rows, mat, inverse=True):
    tex, fmt = _tex(tex)
    assert fmt == TEX_U8X4N
    out = np.empty((rows, cols, 4), np.uint8)
    xp, yp = np.empty((rows, cols), np.float32), np.empty((rows, cols), np.float32)
    lib().nmo_resample_perspective(_fp(out), _fp(tex), C.c_int(tex.shape[1]), C.c_int(tex.shape[0]), C.c_int(cols),
                                   C.c_int(rows), _fp(xp), _fp(yp), _fp(_f32(mat).reshape(9)), C.c_int(1 if inverse else 0))
    return out, xp, yp


def transform_blend(canvas, canvas_wts, frame, nw, nh, mat, tx, ty, mask, wts):
    """Returns updated COPIES of (canvas, canvas_wts)."""
    canvas = np.ascontiguousarray(canvas, dtype=np.uint8).copy()
    canvas_wts = _f32(canvas_wts).copy()
    frame, ffmt = _tex(frame)
    assert ffmt == TEX_U8X4N
    mask, mfmt = _tex(mask)
    wts, wfmt = _tex(wts)
    ch, cw, _ = canvas.shape
    fh, fw, _ = frame.shape
    assert mask.shape == (fh, fw) and wts.shape == (fh, fw)
    lib().nmo_transform_blend(_fp(canvas), C.c_int(cw), C.c_int(ch), _fp(frame), C.c_int(fw), C.c_int(fh), C.c_int(nw),
                              C.c_int(nh), _fp(_f32(mat).reshape(9)), C.c_int(tx), C.c_int(ty), _fp(mask), C.c_int(mfmt),
                              _fp(canvas_wts), _fp(wts), C.c_int(wfmt))
    return canvas, canvas_wts
