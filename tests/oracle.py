"""ctypes binding of the CPU ORACLE (oracle/liborbref.so).  Test infrastructure: imported only by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB = os.path.join(ORACLE_DIR, "liborbref.so")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28


def build():
    if os.environ.get("ORBREF_LIB"):                   # e.g. oracle/liborbref_asan.so (tests/test_oracle_pins.py)
        return os.environ["ORBREF_LIB"]
    srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith((".cpp", ".hpp"))]
    if (not os.path.exists(_LIB)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liborbref.so"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        L = _lib
        L.orbref_create.restype = C.c_void_p
        L.orbref_create.argtypes = [C.c_int, C.c_float, C.c_int, C.c_int, C.c_int]
        L.orbref_destroy.argtypes = [C.c_void_p]
        L.orbref_extract.restype = C.c_int
        L.orbref_extract.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
        L.orbref_tables.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        L.orbref_compute_pyramid.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t]
        L.orbref_pyramid_dims.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orbref_pyramid_copy.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orbref_level_counts.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orbref_cell_candidates.restype = C.c_int
        L.orbref_cell_candidates.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        L.orbref_distribute.restype = C.c_int
        L.orbref_distribute.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_int] * 5 + [C.c_void_p, C.c_int]
        L.orbref_ic_angle.restype = C.c_float
        L.orbref_ic_angle.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orbref_descriptor.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]
        L.orbref_resize.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.orbref_resize_tab.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orbref_blur.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orbref_blur_taps.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orbref_set_gaussian.argtypes = [C.c_void_p, C.c_void_p]
        L.orbref_fast_vmap.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orbref_fast_atan2.restype = C.c_float
        L.orbref_fast_atan2.argtypes = [C.c_float, C.c_float]
        L.orbref_sincos.argtypes = [C.c_float, C.c_void_p, C.c_void_p]
        L.orbref_cvround.restype = C.c_int
        L.orbref_cvround.argtypes = [C.c_float]
        L.orbref_hamming.restype = C.c_int
        L.orbref_hamming.argtypes = [C.c_void_p, C.c_void_p]
        L.orbref_three_maxima.argtypes = [C.c_void_p, C.c_void_p]
        L.orbref_bow_transform.restype = C.c_int
        L.orbref_bow_transform.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orbref_search_by_bow.restype = C.c_int
        L.orbref_search_by_bow.argtypes = ([C.c_void_p] * 6 + [C.c_int] + [C.c_void_p] * 2 + [C.c_int] +
                                           [C.c_void_p] * 3 + [C.c_int, C.c_float, C.c_int, C.c_void_p])
        L.orbref_search_by_bow_kk.restype = C.c_int
        L.orbref_search_by_bow_kk.argtypes = ([C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int] +
                                              [C.c_void_p] * 3 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int] +
                                              [C.c_float, C.c_int, C.c_void_p])
        L.orbref_search_for_init.restype = C.c_int
        L.orbref_search_for_init.argtypes = ([C.c_void_p, C.c_void_p, C.c_int] * 2 + [C.c_float] * 4 +
                                             [C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p])
        L.orbref_stereo.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                    C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.orbref_search_by_projection.restype = C.c_int
        L.orbref_search_by_projection.argtypes = ([C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                    C.c_void_p, C.c_void_p, C.c_int] + [C.c_float] * 5 + [C.c_int, C.c_int, C.c_void_p])
        L.orbref_vocab_transform.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orbref_distinctive.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orbref_search_by_projection_best.argtypes = ([C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] +
                                                         [C.c_float] * 4 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p])
        L.orbref_search_for_triangulation.restype = C.c_int
        L.orbref_search_for_triangulation.argtypes = ([C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int]) * 2 + \
            [C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orbref_features_in_area.restype = C.c_int
        L.orbref_features_in_area.argtypes = ([C.c_void_p, C.c_int] + [C.c_float] * 7 + [C.c_int] * 2 +
                                              [C.c_void_p, C.c_int])
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class FeatVec:
    """CSR flattening of a DBoW2::FeatureVector."""

    def __init__(self, node_ids, offsets, indices):
        self.node_ids = np.ascontiguousarray(node_ids, dtype=np.uint32)
        self.offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        self.indices = np.ascontiguousarray(indices, dtype=np.int32)

    @property
    def nnodes(self):
        return int(self.node_ids.shape[0])


class Extractor:
    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7):
        self.L = lib()
        self.nlevels = nlevels
        self.nfeatures = nfeatures
        self.h = C.c_void_p(self.L.orbref_create(nfeatures, scale_factor, nlevels, ini_th, min_th))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orbref_destroy(self.h)
            self.h = None

    def set_gaussian(self, taps4):
        """taps k0..k3 of the symmetric 8.8 fixed-point 7-tap kernel: OpenCV-version dependent"""
        t = np.ascontiguousarray(taps4, np.int32)
        assert t.shape == (4,)
        self.L.orbref_set_gaussian(self.h, _p(t))

    def tables(self):
        n = self.nlevels
        sc, inv, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        quota = np.zeros(n, np.int32)
        umax = np.zeros(16, np.int32)
        self.L.orbref_tables(self.h, _p(sc), _p(inv), _p(s2), _p(is2), _p(quota), _p(umax))
        return dict(scale=sc, inv_scale=inv, sigma2=s2, inv_sigma2=is2, quota=quota, umax=umax)

    def extract(self, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        cap = self.nfeatures + 64 * self.nlevels + 64
        kps = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = self.L.orbref_extract(self.h, _p(img), img.shape[0], img.shape[1], img.strides[0], _p(kps), _p(desc), cap)
        assert n <= cap, n
        return kps[:n].copy(), desc[:n].copy()

    def compute_pyramid(self, img):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        self.L.orbref_compute_pyramid(self.h, _p(img), img.shape[0], img.shape[1], img.strides[0])

    def pyramid_level(self, level):
        w, h = C.c_int(), C.c_int()
        self.L.orbref_pyramid_dims(self.h, level, C.byref(w), C.byref(h))
        out = np.zeros((h.value, w.value), np.uint8)
        self.L.orbref_pyramid_copy(self.h, level, _p(out))
        return out

    def level_counts(self):
        kept = np.zeros(self.nlevels, np.int32)
        cands = np.zeros(self.nlevels, np.int32)
        self.L.orbref_level_counts(self.h, _p(kept), _p(cands))
        return kept, cands

    def cell_candidates(self, level, cap=200000):
        out = np.zeros((cap, 3), np.int32)
        n = self.L.orbref_cell_candidates(self.h, level, _p(out), cap)
        assert n <= cap
        return out[:n].copy()

    def distribute(self, cands, min_x, max_x, min_y, max_y, N):
        cands = np.ascontiguousarray(cands, dtype=np.int32)
        cap = max(N + 64, 4 * 64)
        out = np.zeros((cap, 3), np.int32)
        n = self.L.orbref_distribute(self.h, _p(cands), cands.shape[0], min_x, max_x, min_y, max_y, N, _p(out), cap)
        assert n <= cap
        return out[:n].copy()

    def ic_angle(self, level, x, y):
        return np.float32(self.L.orbref_ic_angle(self.h, level, x, y))

    def descriptor(self, blurred, x, y, angle):
        blurred = np.ascontiguousarray(blurred, dtype=np.uint8)
        out = np.zeros(32, np.uint8)
        self.L.orbref_descriptor(self.h, _p(blurred), blurred.shape[1], blurred.shape[0], x, y, float(angle), _p(out))
        return out


def resize(src, dw, dh):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    dst = np.zeros((dh, dw), np.uint8)
    lib().orbref_resize(_p(src), src.shape[1], src.shape[0], _p(dst), dw, dh)
    return dst


def resize_tab(src_len, dst_len):
    ofs = np.zeros(dst_len, np.int32)
    c0 = np.zeros(dst_len, np.int16)
    c1 = np.zeros(dst_len, np.int16)
    lib().orbref_resize_tab(src_len, dst_len, _p(ofs), _p(c0), _p(c1))
    return ofs, c0, c1


def blur(src, taps4=None):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    dst = np.zeros_like(src)
    if taps4 is None:
        lib().orbref_blur(_p(src), src.shape[1], src.shape[0], _p(dst))
    else:
        t = np.ascontiguousarray(taps4, np.int32)
        lib().orbref_blur_taps(_p(src), src.shape[1], src.shape[0], _p(dst), _p(t))
    return dst


def fast_vmap(img):
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.zeros(img.shape, np.int16)
    lib().orbref_fast_vmap(_p(img), img.shape[1], img.shape[0], _p(out))
    return out


def fast_atan2(y, x):
    return np.float32(lib().orbref_fast_atan2(float(y), float(x)))


def sincos(x):
    c, s = C.c_float(), C.c_float()
    lib().orbref_sincos(float(x), C.byref(c), C.byref(s))
    return np.float32(c.value), np.float32(s.value)


def cvround(v):
    return lib().orbref_cvround(float(v))


def hamming(a, b):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    b = np.ascontiguousarray(b, dtype=np.uint8)
    return lib().orbref_hamming(_p(a), _p(b))


def three_maxima(counts):
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    out = np.zeros(3, np.int32)
    lib().orbref_three_maxima(_p(counts), _p(out))
    return tuple(int(v) for v in out)


def bow_transform(desc, centroids):
    desc = np.ascontiguousarray(desc, dtype=np.uint8)
    centroids = np.ascontiguousarray(centroids, dtype=np.uint8)
    n = desc.shape[0]
    ids = np.zeros(128, np.uint32)
    offs = np.zeros(129, np.int32)
    idx = np.zeros(max(n, 1), np.int32)
    nn = lib().orbref_bow_transform(_p(desc), n, _p(centroids), _p(ids), _p(offs), _p(idx))
    return FeatVec(ids[:nn], offs[:nn + 1], idx[:n])


def featvec_from_nodes(node_id):
    """FeatureVector of DBoW2's transform: fv[nid].push_back(i) for i ascending (features whose descent ended above
    the level, node_id < 0, are in no node) -> ascending node ids, ascending indices inside a node."""
    node_id = np.asarray(node_id, np.int64)
    idx = np.nonzero(node_id >= 0)[0]
    order = idx[np.argsort(node_id[idx], kind="stable")]
    ids, counts = np.unique(node_id[idx], return_counts=True)
    offs = np.zeros(ids.size + 1, np.int32)
    np.cumsum(counts, out=offs[1:])
    return FeatVec(ids.astype(np.uint32), offs, order.astype(np.int32))


def search_by_bow(desc_kf, angle_kf, valid_kf, fv_kf, desc_f, angle_f, fv_f, ratio=0.7, check_ori=True):
    desc_kf = np.ascontiguousarray(desc_kf, np.uint8); desc_f = np.ascontiguousarray(desc_f, np.uint8)
    angle_kf = np.ascontiguousarray(angle_kf, np.float32); angle_f = np.ascontiguousarray(angle_f, np.float32)
    valid_kf = np.ascontiguousarray(valid_kf, np.uint8)
    nF = desc_f.shape[0]
    out = np.full(max(nF, 1), -1, np.int32)
    nm = lib().orbref_search_by_bow(_p(desc_kf), _p(angle_kf), _p(valid_kf), _p(fv_kf.node_ids), _p(fv_kf.offsets),
                                    _p(fv_kf.indices), fv_kf.nnodes, _p(desc_f), _p(angle_f), nF,
                                    _p(fv_f.node_ids), _p(fv_f.offsets), _p(fv_f.indices), fv_f.nnodes,
                                    ratio, int(check_ori), _p(out))
    return nm, out[:nF]


def search_by_bow_kk(d1, a1, v1, fv1, d2, a2, v2, fv2, ratio=0.75, check_ori=True):
    d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
    a1 = np.ascontiguousarray(a1, np.float32); a2 = np.ascontiguousarray(a2, np.float32)
    v1 = np.ascontiguousarray(v1, np.uint8); v2 = np.ascontiguousarray(v2, np.uint8)
    n1, n2 = d1.shape[0], d2.shape[0]
    out = np.full(max(n1, 1), -1, np.int32)
    nm = lib().orbref_search_by_bow_kk(_p(d1), _p(a1), _p(v1), n1, _p(fv1.node_ids), _p(fv1.offsets), _p(fv1.indices),
                                       fv1.nnodes, _p(d2), _p(a2), _p(v2), n2, _p(fv2.node_ids), _p(fv2.offsets),
                                       _p(fv2.indices), fv2.nnodes, ratio, int(check_ori), _p(out))
    return nm, out[:n1]


def search_for_init(k1, d1, k2, d2, grid, prev_xy, window=100, ratio=0.9, check_ori=True):
    """grid = (minX, minY, invW, invH); prev_xy (n1,2) float32 is updated in place."""
    k1 = np.ascontiguousarray(k1); k2 = np.ascontiguousarray(k2)
    d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
    assert prev_xy.dtype == np.float32 and prev_xy.flags.c_contiguous
    n1, n2 = k1.shape[0], k2.shape[0]
    m12 = np.full(max(n1, 1), -1, np.int32)
    nm = lib().orbref_search_for_init(_p(k1), _p(d1), n1, _p(k2), _p(d2), n2, *[float(g) for g in grid],
                                      _p(prev_xy), window, ratio, int(check_ori), _p(m12))
    return nm, m12[:n1]


def features_in_area(kps, grid, x, y, r, min_level=-1, max_level=-1):
    kps = np.ascontiguousarray(kps)
    cap = max(kps.shape[0], 1)
    out = np.zeros(cap, np.int32)
    n = lib().orbref_features_in_area(_p(kps), kps.shape[0], *[float(g) for g in grid], float(x), float(y), float(r),
                                      min_level, max_level, _p(out), cap)
    return out[:n].copy()


def stereo_matches(ex_left, ex_right, k_l, d_l, k_r, d_r, mb, mbf):
    """Frame::ComputeStereoMatches on the pyramids currently held by the two oracle extractors."""
    k_l = np.ascontiguousarray(k_l); k_r = np.ascontiguousarray(k_r)
    d_l = np.ascontiguousarray(d_l, np.uint8); d_r = np.ascontiguousarray(d_r, np.uint8)
    n = k_l.shape[0]
    u = np.zeros(max(n, 1), np.float32)
    z = np.zeros(max(n, 1), np.float32)
    lib().orbref_stereo(ex_left.h, ex_right.h, _p(k_l), _p(d_l), n, _p(k_r), _p(d_r), k_r.shape[0], mb, mbf, _p(u), _p(z))
    return u[:n], z[:n]


PROJ_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("r", "<f4"), ("min_level", "<i4"), ("max_level", "<i4"),
                       ("ur", "<f4"), ("er_max", "<f4"), ("flags", "<i4")])


def search_by_projection(mode, q, q_desc, q_angle, kps, desc, u_right, occupied, grid, ratio=0.8, check_ori=True, max_dist=100):
    """mode 0: SearchByProjection(CurrentFrame, LastFrame, ...); mode 1: SearchByProjection(Frame, MapPoints, ...)."""
    q = np.ascontiguousarray(q, PROJ_DTYPE); q_desc = np.ascontiguousarray(q_desc, np.uint8)
    q_angle = np.ascontiguousarray(q_angle, np.float32)
    kps = np.ascontiguousarray(kps); desc = np.ascontiguousarray(desc, np.uint8)
    u_right = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
    occupied = np.ascontiguousarray(occupied, np.uint8)
    n = kps.shape[0]
    out = np.full(max(n, 1), -1, np.int32)
    nm = lib().orbref_search_by_projection(mode, _p(q), _p(q_desc), _p(q_angle), q.shape[0], _p(kps), _p(desc), _p(u_right),
                                           _p(occupied), n, *[float(g) for g in grid], ratio, int(max_dist), int(check_ori),
                                           _p(out))
    return nm, out[:n]


def vocab_transform(tree, desc, levelsup=4):
    """tree = dict(node_desc, child_begin, children, word_id, L) (see orbhip.synth.synth_vocab_tree)."""
    desc = np.ascontiguousarray(desc, np.uint8)
    n = desc.shape[0]
    word = np.zeros(max(n, 1), np.int32)
    node = np.zeros(max(n, 1), np.int32)
    lib().orbref_vocab_transform(_p(tree["node_desc"]), _p(tree["child_begin"]), _p(tree["children"]), _p(tree["word_id"]),
                                 tree["node_desc"].shape[0], tree["L"], _p(desc), n, levelsup, _p(word), _p(node))
    return word[:n], node[:n]


def distinctive_descriptors(desc, offsets):
    desc = np.ascontiguousarray(desc, np.uint8); offsets = np.ascontiguousarray(offsets, np.int32)
    n = offsets.shape[0] - 1
    out = np.zeros(max(n, 1), np.int32)
    lib().orbref_distinctive(_p(desc), _p(offsets), n, _p(out))
    return out[:n]


def search_by_projection_best(q, q_desc, kps, desc, u_right, grid, max_dist=50, chi2=False, inv_sigma2=None):
    """Independent best candidate per projected point (search loops of Fuse x2 / SearchBySim3)."""
    q = np.ascontiguousarray(q, PROJ_DTYPE); q_desc = np.ascontiguousarray(q_desc, np.uint8)
    kps = np.ascontiguousarray(kps); desc = np.ascontiguousarray(desc, np.uint8)
    u_right = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
    inv_sigma2 = None if inv_sigma2 is None else np.ascontiguousarray(inv_sigma2, np.float32)
    nq = q.shape[0]
    best = np.full(max(nq, 1), -1, np.int32)
    dist = np.full(max(nq, 1), 256, np.int32)
    lib().orbref_search_by_projection_best(_p(q), _p(q_desc), nq, _p(kps), _p(desc), _p(u_right), kps.shape[0],
                                           *[float(g) for g in grid], int(max_dist), int(chi2), _p(inv_sigma2), _p(best), _p(dist))
    return best[:nq], dist[:nq]


def search_for_triangulation(k1, d1, mp1, ur1, fv1, k2, d2, mp2, ur2, fv2, F12, ex, ey, sf2, sig2, only_stereo=False,
                             check_ori=True):
    k1 = np.ascontiguousarray(k1); k2 = np.ascontiguousarray(k2)
    d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
    mp1 = np.ascontiguousarray(mp1, np.uint8); mp2 = np.ascontiguousarray(mp2, np.uint8)
    ur1 = None if ur1 is None else np.ascontiguousarray(ur1, np.float32)
    ur2 = None if ur2 is None else np.ascontiguousarray(ur2, np.float32)
    F12 = np.ascontiguousarray(F12, np.float32); sf2 = np.ascontiguousarray(sf2, np.float32); sig2 = np.ascontiguousarray(sig2, np.float32)
    n1 = k1.shape[0]
    out = np.full(max(n1, 1), -1, np.int32)
    nm = lib().orbref_search_for_triangulation(_p(k1), _p(d1), _p(mp1), _p(ur1), n1, _p(fv1.node_ids), _p(fv1.offsets),
                                               _p(fv1.indices), fv1.nnodes, _p(k2), _p(d2), _p(mp2), _p(ur2), k2.shape[0],
                                               _p(fv2.node_ids), _p(fv2.offsets), _p(fv2.indices), fv2.nnodes, _p(F12),
                                               float(ex), float(ey), _p(sf2), _p(sig2), int(only_stereo), int(check_ori), _p(out))
    return nm, out[:n1]
