"""ctypes binding of liborbhip.so (include/orb_hip.h) -- thin plumbing used by tests and bench.py.

The product is the C-ABI shared library; this module only marshals numpy arrays / raw device
pointers into it.  It fails loudly when the library is missing or no GPU is usable: there is no
CPU fallback anywhere on the product path.
"""
import ctypes as C
import os
import subprocess

import numpy as np

PKG_DIR = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # orb-slam2-chinesenotes_amd/
LIB_PATH = os.path.join(PKG_DIR, "liborbhip.so")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                     ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])

PROJ_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("r", "<f4"), ("min_level", "<i4"), ("max_level", "<i4"),
                       ("ur", "<f4"), ("er_max", "<f4"), ("flags", "<i4")])

ORB_OK = 0
STATUS = {0: "ORB_OK", -1: "ORB_ERR_INVALID", -2: "ORB_ERR_HIP", -3: "ORB_ERR_NO_DEVICE",
          -4: "ORB_ERR_CAPACITY", -5: "ORB_ERR_UNSUPPORTED", -6: "ORB_ERR_INTERNAL"}


class OrbError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (STATUS.get(code, "?"), code, msg))
        self.code = code


class Params(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("ini_th_fast", C.c_int32), ("min_th_fast", C.c_int32)]


class FeatVecC(C.Structure):
    _fields_ = [("node_ids", C.c_void_p), ("offsets", C.c_void_p), ("indices", C.c_void_p), ("n_nodes", C.c_int32)]


class FeatStoreC(C.Structure):
    _fields_ = [("desc", C.c_void_p), ("kps", C.c_void_p), ("valid", C.c_void_p), ("counts", C.c_void_p),
                ("node_of", C.c_void_p), ("cap", C.c_int32), ("n_frames", C.c_int32), ("n_nodes", C.c_int32),
                ("csr_keys", C.c_void_p), ("csr_start", C.c_void_p), ("csr_cnt", C.c_void_p), ("csr_desc", C.c_void_p)]


def build_library(force=False):
    """hipcc cross-compiles for gfx950 without a GPU; output stays in-tree."""
    args = ["make", "-C", PKG_DIR, "-j4", "liborbhip.so"]
    if force:
        subprocess.check_call(["make", "-C", PKG_DIR, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None

# every symbol include/orb_hip.h declares
SYMBOLS = [
    "orb_extractor_create", "orb_extractor_destroy", "orb_extractor_get_tables", "orb_extractor_max_keypoints",
    "orb_extractor_set_pattern", "orb_extractor_set_pattern_device", "orb_extractor_desc_plan", "orb_extractor_set_desc_stamps", "orb_extractor_set_pyr_stamps", "orb_extractor_set_qt_stamps", "orb_extractor_pyr_stamp_layout", "orb_extractor_pyr_persistent", "orb_builtin_pattern", "orb_extract",
    "orb_extract_batch", "orb_extract_batch_device", "orb_extractor_sync", "orb_get_pyramid_level",
    "orb_get_level_counts", "orb_get_fast_overflows", "orb_get_pyramid", "orb_host_alloc", "orb_host_free", "orb_extractor_set_profiling", "orb_extractor_get_stage_ms", "orb_extractor_profiled_frames", "orb_extractor_stream",
    "orb_hamming", "orb_three_maxima", "orb_matcher_create", "orb_matcher_destroy", "orb_matcher_sync",
    "orb_match_bow", "orb_match_bow_kk", "orb_match_init", "orb_match_projection", "orb_match_projection_best", "orb_match_triangulation", "orb_vocab_create", "orb_vocab_destroy", "orb_vocab_level_nodes", "orb_bow_transform",
    "orb_bow_transform_device", "orb_distinctive_descriptors", "orb_distinctive_descriptors_device", "orb_bow_assign_device", "orb_match_bow_batch_device", "orb_bow_build_csr_device",
    "orb_bow_build_csr_desc_device", "orb_match_bow_query_device", "orb_bow_query_frames_device", "orb_matcher_set_stage_stamps", "orb_gaussian_preset", "orb_extractor_set_gaussian",
    "orb_matcher_stream", "orb_stereo_match", "orb_stereo_match_device", "orb_stereo_match_batch_device", "orb_extractor_wait_for", "orb_matcher_wait_for", "orb_last_error", "orb_version", "orb_abi_version", "orb_sizeof_featstore", "orb_multi_create", "orb_multi_destroy", "orb_multi_devices", "orb_multi_handle",
    "orb_multi_set_pattern", "orb_multi_extract_batch", "orb_shard_range", "orb_multi_db_create", "orb_multi_db_destroy",
    "orb_multi_db_shards", "orb_multi_match_bow_batch",
]


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OrbError(-3, "liborbhip.so is not built (run __graft_entry__.build()); no CPU fallback exists")
    try:                       # one HIP runtime per process: let torch's copy win if torch is around
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(LIB_PATH)
    vp, ci, cf, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    L.orb_extractor_create.argtypes = [C.POINTER(Params), ci, C.POINTER(vp)]
    L.orb_extractor_destroy.argtypes = [vp]
    L.orb_extractor_destroy.restype = None
    L.orb_extractor_get_tables.argtypes = [vp] * 6
    L.orb_extractor_max_keypoints.argtypes = [vp]
    L.orb_extractor_set_pattern.argtypes = [vp, vp]
    L.orb_gaussian_preset.argtypes = [ci, vp]
    L.orb_extractor_set_gaussian.argtypes = [vp, vp]
    L.orb_extractor_set_pattern_device.argtypes = [vp, vp]
    L.orb_extractor_desc_plan.argtypes = [vp, vp, vp]
    L.orb_extractor_set_desc_stamps.argtypes = [vp, vp, sz]
    L.orb_extractor_set_pyr_stamps.argtypes = [vp, vp, sz]
    L.orb_extractor_set_qt_stamps.argtypes = [vp, vp, sz]
    L.orb_extractor_pyr_stamp_layout.argtypes = [vp, vp, vp, vp]
    L.orb_extractor_pyr_persistent.argtypes = [vp, vp]
    L.orb_builtin_pattern.argtypes = [vp]
    L.orb_extract.argtypes = [vp, vp, ci, ci, sz, vp, vp, ci, C.POINTER(ci)]
    L.orb_extract_batch.argtypes = [vp, vp, ci, ci, ci, sz, sz, vp, vp, ci, vp]
    L.orb_extract_batch_device.argtypes = [vp, vp, ci, ci, ci, sz, sz, vp, vp, ci, vp]
    L.orb_extractor_sync.argtypes = [vp]
    L.orb_get_pyramid_level.argtypes = [vp, ci, ci, vp, sz, C.POINTER(ci), C.POINTER(ci)]
    L.orb_get_level_counts.argtypes = [vp, ci, vp, vp]
    L.orb_get_fast_overflows.argtypes = [vp, vp, vp]
    L.orb_get_pyramid.argtypes = [vp, ci, vp, sz, C.POINTER(sz), vp, vp, vp, vp]
    L.orb_host_alloc.argtypes = [sz]
    L.orb_host_alloc.restype = vp
    L.orb_host_free.argtypes = [vp]
    L.orb_host_free.restype = None
    L.orb_extractor_set_profiling.argtypes = [vp, ci]
    L.orb_extractor_get_stage_ms.argtypes = [vp, vp]
    L.orb_extractor_profiled_frames.argtypes = [vp]
    L.orb_extractor_stream.argtypes = [vp]
    L.orb_extractor_stream.restype = vp
    L.orb_hamming.argtypes = [vp, vp]
    L.orb_three_maxima.argtypes = [vp, vp]
    L.orb_three_maxima.restype = None
    L.orb_matcher_create.argtypes = [ci, C.POINTER(vp)]
    L.orb_matcher_destroy.argtypes = [vp]
    L.orb_matcher_destroy.restype = None
    L.orb_matcher_sync.argtypes = [vp]
    L.orb_match_bow.argtypes = [vp, vp, vp, vp, ci, C.POINTER(FeatVecC), vp, vp, ci, C.POINTER(FeatVecC), cf, ci, vp,
                                C.POINTER(ci)]
    L.orb_match_bow_kk.argtypes = [vp, vp, vp, vp, ci, C.POINTER(FeatVecC), vp, vp, vp, ci, C.POINTER(FeatVecC), cf, ci,
                                   vp, C.POINTER(ci)]
    L.orb_match_init.argtypes = [vp, vp, vp, ci, vp, vp, ci, vp, vp, ci, cf, ci, vp, C.POINTER(ci)]
    L.orb_match_projection.argtypes = [vp, ci, vp, vp, vp, ci, vp, vp, vp, vp, ci, vp, cf, ci, ci, vp, C.POINTER(ci)]
    L.orb_match_projection_best.argtypes = [vp, vp, vp, ci, vp, vp, vp, ci, vp, ci, ci, vp, ci, vp, vp]
    L.orb_match_triangulation.argtypes = [vp, vp, vp, vp, vp, ci, C.POINTER(FeatVecC), vp, vp, vp, vp, ci, C.POINTER(FeatVecC), vp, cf, cf,
                                          vp, vp, ci, ci, ci, vp, C.POINTER(ci)]
    L.orb_vocab_create.argtypes = [ci, vp, vp, vp, vp, ci, ci, C.POINTER(vp)]
    L.orb_vocab_destroy.argtypes = [vp]
    L.orb_vocab_destroy.restype = None
    L.orb_vocab_level_nodes.argtypes = [vp, ci]
    L.orb_bow_transform.argtypes = [vp, vp, vp, ci, ci, vp, vp]
    L.orb_bow_transform_device.argtypes = [vp, vp, vp, vp, ci, ci, ci, vp, vp, vp]
    L.orb_distinctive_descriptors.argtypes = [vp, vp, vp, ci, vp]
    L.orb_distinctive_descriptors_device.argtypes = [vp, vp, vp, ci, vp]
    L.orb_bow_assign_device.argtypes = [vp, vp, vp, ci, ci, vp, vp]
    L.orb_match_bow_batch_device.argtypes = [vp, C.POINTER(FeatStoreC), vp, vp, ci, cf, ci, vp, vp]
    L.orb_bow_build_csr_device.argtypes = [vp, vp, vp, ci, ci, ci, vp, vp, vp]
    L.orb_bow_build_csr_desc_device.argtypes = [vp, vp, vp, vp, ci, ci, ci, vp, vp, vp, vp]
    L.orb_matcher_set_stage_stamps.argtypes = [vp, vp, C.c_size_t]
    L.orb_match_bow_query_device.argtypes = [vp, C.POINTER(FeatStoreC), vp, ci, vp, ci, cf, ci, vp, vp]
    L.orb_bow_query_frames_device.argtypes = [vp, vp, C.POINTER(FeatStoreC), ci, ci, ci, vp, ci, vp, cf, ci, vp, vp]
    L.orb_matcher_stream.argtypes = [vp]
    L.orb_matcher_stream.restype = vp
    L.orb_extractor_wait_for.argtypes = [vp, vp]
    L.orb_stereo_match.argtypes = [vp, vp, vp, vp, ci, vp, vp, ci, cf, cf, vp, vp]
    L.orb_stereo_match_device.argtypes = [vp, vp, ci, ci, vp, vp, ci, vp, vp, ci, cf, cf, vp, vp]
    L.orb_stereo_match_batch_device.argtypes = [vp, vp, ci, ci, ci, vp, vp, vp, vp, vp, vp, ci, cf, cf, vp, vp]
    L.orb_matcher_wait_for.argtypes = [vp, vp]
    L.orb_multi_create.argtypes = [C.POINTER(Params), vp, ci, C.POINTER(vp)]
    L.orb_multi_destroy.argtypes = [vp]
    L.orb_multi_db_create.argtypes = [vp, ci, vp, vp, vp, vp, vp, ci, ci, ci, C.POINTER(vp)]
    L.orb_multi_db_destroy.argtypes = [vp]
    L.orb_multi_db_destroy.restype = None
    L.orb_multi_db_shards.argtypes = [vp]
    L.orb_multi_match_bow_batch.argtypes = [vp, vp, vp, ci, vp, cf, ci, vp, vp]
    L.orb_multi_destroy.restype = None
    L.orb_multi_devices.argtypes = [vp]
    L.orb_multi_handle.argtypes = [vp, ci]
    L.orb_multi_handle.restype = vp
    L.orb_multi_set_pattern.argtypes = [vp, vp]
    L.orb_multi_extract_batch.argtypes = [vp, vp, ci, ci, ci, sz, sz, vp, vp, ci, vp]
    L.orb_shard_range.argtypes = [ci, ci, ci, C.POINTER(ci), C.POINTER(ci)]
    L.orb_shard_range.restype = None
    L.orb_last_error.restype = C.c_char_p
    L.orb_version.restype = C.c_char_p
    if ORDER_BEHIND_TORCH:
        _order_behind_torch(L)
    _lib = L
    return L


# The library's streams are non-blocking: nothing orders them behind work that torch has queued on ITS stream (a fill of an
# output buffer, an index tensor still being written).  The C contract is the caller's: buffers handed to a *_device entry
# are ready.  Tests set ORDER_BEHIND_TORCH before the first call, and every entry that takes device pointers then waits for
# torch's current stream first (bench.py does not: its timed loops order their buffers themselves).
ORDER_BEHIND_TORCH = False


def _order_behind_torch(L):
    import torch

    def wrap(fn):
        def call(*a):
            if torch.cuda.is_initialized():
                torch.cuda.current_stream().synchronize()
            return fn(*a)
        return call

    for name in SYMBOLS:
        if "_device" in name or name in ("orb_multi_extract_batch", "orb_multi_match_bow_batch", "orb_bow_transform_device"):
            setattr(L, name, wrap(getattr(L, name)))


def _check(rc):
    if rc != ORB_OK:
        raise OrbError(rc, lib().orb_last_error().decode(errors="replace"))


def _p(a):
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    return a.ctypes.data_as(C.c_void_p)


def builtin_pattern():
    out = np.zeros(1024, np.int8)
    _check(lib().orb_builtin_pattern(_p(out)))
    return out


def hamming(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return lib().orb_hamming(_p(a), _p(b))


def three_maxima(counts):
    counts = np.ascontiguousarray(counts, np.int32)
    out = np.zeros(3, np.int32)
    lib().orb_three_maxima(_p(counts), _p(out))
    return tuple(int(v) for v in out)


class Extractor:
    """Mirror of ORB_SLAM2::ORBextractor (reference include/ORBextractor.h:46-112) over the C ABI."""

    def __init__(self, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7, device=0):
        self.L = lib()
        self.h = C.c_void_p()
        self.nlevels = nlevels
        prm = Params(nfeatures, scale_factor, nlevels, ini_th, min_th)
        _check(self.L.orb_extractor_create(C.byref(prm), device, C.byref(self.h)))
        self.max_keypoints = self.L.orb_extractor_max_keypoints(self.h)

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.L.orb_extractor_destroy(self.h)
            self.h = C.c_void_p()

    __del__ = close

    def tables(self):
        n = self.nlevels
        sc, inv, s2, is2 = (np.zeros(n, np.float32) for _ in range(4))
        quota = np.zeros(n, np.int32)
        _check(self.L.orb_extractor_get_tables(self.h, _p(sc), _p(inv), _p(s2), _p(is2), _p(quota)))
        return dict(scale=sc, inv_scale=inv, sigma2=s2, inv_sigma2=is2, quota=quota)

    def set_gaussian(self, taps4_or_preset):
        """an int = one of the presets (0 legacy {18,34,49,55}, 1 error-diffused {18,34,48,56}); else the four taps k0..k3"""
        t = np.zeros(4, np.int32)
        if np.isscalar(taps4_or_preset):
            _check(self.L.orb_gaussian_preset(int(taps4_or_preset), _p(t)))
        else:
            t[:] = taps4_or_preset
        _check(self.L.orb_extractor_set_gaussian(self.h, _p(t)))
        return t

    def set_pattern(self, pattern):
        pattern = np.ascontiguousarray(pattern, np.int8)
        assert pattern.size == 1024
        _check(self.L.orb_extractor_set_pattern(self.h, _p(pattern)))

    def set_pattern_device(self, dptr):
        _check(self.L.orb_extractor_set_pattern_device(self.h, C.c_void_p(dptr)))

    def extract(self, img):
        """== operator()(image, mask, keypoints, descriptors) on host buffers."""
        img = np.asarray(img, dtype=np.uint8)
        if img.size and img.strides[1] != 1:
            img = np.ascontiguousarray(img)
        cap = self.max_keypoints
        kps = np.empty(cap, KP_DTYPE)                      # (only the first n entries are written and returned)
        desc = np.empty((cap, 32), np.uint8)
        n = C.c_int(0)
        rows, cols = (img.shape if img.ndim == 2 else (0, 0))
        stride = img.strides[0] if img.size else 0
        _check(self.L.orb_extract(self.h, _p(img) if img.size else None, rows, cols, stride, _p(kps), _p(desc), cap,
                                  C.byref(n)))
        return kps[:n.value].copy(), desc[:n.value].copy()

    def extract_batch(self, imgs):
        imgs = np.ascontiguousarray(imgs, dtype=np.uint8)
        f, rows, cols = imgs.shape
        cap = self.max_keypoints
        kps = np.zeros((f, cap), KP_DTYPE)
        desc = np.zeros((f, cap, 32), np.uint8)
        counts = np.zeros(f, np.int32)
        _check(self.L.orb_extract_batch(self.h, _p(imgs), f, rows, cols, imgs.strides[1], imgs.strides[0], _p(kps),
                                        _p(desc), cap, _p(counts)))
        return [(kps[i, :counts[i]].copy(), desc[i, :counts[i]].copy()) for i in range(f)]

    def extract_batch_into(self, imgs, kps, desc, counts):
        """Host batch into caller-owned arrays (pinned ones are copied to / from directly): imgs (f, rows, cols) u8,
        kps (f, cap) KP_DTYPE or raw bytes, desc (f, cap, 32) u8, counts (f) i32."""
        f, rows, cols = imgs.shape
        cap = self.max_keypoints
        _check(self.L.orb_extract_batch(self.h, _p(imgs), f, rows, cols, imgs.strides[1], imgs.strides[0], _p(kps),
                                        _p(desc), cap, _p(counts)))

    def extract_batch_device(self, d_imgs, n_frames, rows, cols, row_stride, frame_stride, d_kps, d_desc, cap, d_counts):
        """Raw device pointers (ints); asynchronous on the handle's stream."""
        _check(self.L.orb_extract_batch_device(self.h, C.c_void_p(d_imgs), n_frames, rows, cols, row_stride, frame_stride,
                                               C.c_void_p(d_kps), C.c_void_p(d_desc), cap, C.c_void_p(d_counts)))

    def sync(self):
        _check(self.L.orb_extractor_sync(self.h))

    def set_desc_stamps(self, d_ptr, capacity):
        _check(self.L.orb_extractor_set_desc_stamps(self.h, C.c_void_p(d_ptr), capacity))

    def set_qt_stamps(self, d_ptr, capacity):
        _check(self.L.orb_extractor_set_qt_stamps(self.h, C.c_void_p(d_ptr), capacity))

    def set_pyr_stamps(self, d_ptr, capacity):
        _check(self.L.orb_extractor_set_pyr_stamps(self.h, C.c_void_p(d_ptr), capacity))

    def pyr_stamp_layout(self):
        n = C.c_int32()
        bands, steps = np.zeros(8, np.int32), np.zeros(8, np.int32)
        _check(self.L.orb_extractor_pyr_stamp_layout(self.h, C.byref(n), _p(bands), _p(steps)))
        return [(int(bands[i]), int(steps[i])) for i in range(n.value)]

    def pyr_persistent(self):
        """Pyramid launches of the last batch that took the persistent form (orb_extractor_pyr_persistent)."""
        n = C.c_int32()
        _check(self.L.orb_extractor_pyr_persistent(self.h, C.byref(n)))
        return n.value

    def desc_plan(self):
        """(first level that runs level-resident, number of regions) of the current geometry (orb_extractor_desc_plan)."""
        a, b = C.c_int32(), C.c_int32()
        _check(self.L.orb_extractor_desc_plan(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def pyramid_level(self, frame, level):
        r, c = C.c_int(), C.c_int()
        _check(self.L.orb_get_pyramid_level(self.h, frame, level, None, 0, C.byref(r), C.byref(c)))
        out = np.zeros((r.value, c.value), np.uint8)
        _check(self.L.orb_get_pyramid_level(self.h, frame, level, _p(out), out.strides[0], C.byref(r), C.byref(c)))
        return out

    def pyramid(self, frame=0):
        """All levels of one frame with a single device-to-host copy: list of (rows, cols) u8 arrays (views)."""
        nl = self.nlevels
        need = C.c_size_t(0)
        off = np.zeros(nl, np.int32); pit = np.zeros(nl, np.int32); rw = np.zeros(nl, np.int32); cl = np.zeros(nl, np.int32)
        _check(self.L.orb_get_pyramid(self.h, frame, None, 0, C.byref(need), _p(off), _p(pit), _p(rw), _p(cl)))
        buf = np.zeros(need.value, np.uint8)
        _check(self.L.orb_get_pyramid(self.h, frame, _p(buf), need.value, C.byref(need), _p(off), _p(pit), _p(rw), _p(cl)))
        return [buf[off[l]:off[l] + pit[l] * rw[l]].reshape(rw[l], pit[l])[:, :cl[l]] for l in range(nl)]

    def fast_overflows(self):
        o = np.zeros(self.nlevels, np.int32); s = np.zeros(self.nlevels, np.int32)
        _check(self.L.orb_get_fast_overflows(self.h, _p(o), _p(s)))
        return o, s

    def level_counts(self, frame=0):
        kept = np.zeros(self.nlevels, np.int32)
        cands = np.zeros(self.nlevels, np.int32)
        _check(self.L.orb_get_level_counts(self.h, frame, _p(kept), _p(cands)))
        return kept, cands

    def set_profiling(self, on=True):
        _check(self.L.orb_extractor_set_profiling(self.h, int(on)))

    def stage_ms(self):
        ms = np.zeros(5, np.float32)
        _check(self.L.orb_extractor_get_stage_ms(self.h, _p(ms)))
        return ms

    def profiled_frames(self):
        return self.L.orb_extractor_profiled_frames(self.h)

    @property
    def stream(self):
        return self.L.orb_extractor_stream(self.h)

    def wait_for(self, hip_stream):
        _check(self.L.orb_extractor_wait_for(self.h, C.c_void_p(hip_stream)))


def shard_range(total, world, rank):
    """The C ABI's partition rule (host only)."""
    a, b = C.c_int(0), C.c_int(0)
    lib().orb_shard_range(total, world, rank, C.byref(a), C.byref(b))
    return a.value, b.value


class MultiExtractor:
    """orb_multi_*: one extractor per listed device, frames sharded in contiguous blocks, BRIEF pattern broadcast
    from devices[0] with RCCL."""

    def __init__(self, devices, nfeatures=1000, scale_factor=1.2, nlevels=8, ini_th=20, min_th=7):
        self.L = lib()
        self.h = C.c_void_p()
        self.nlevels = nlevels
        prm = Params(nfeatures, scale_factor, nlevels, ini_th, min_th)
        dev = np.ascontiguousarray(devices, np.int32)
        _check(self.L.orb_multi_create(C.byref(prm), _p(dev), dev.size, C.byref(self.h)))
        self.max_keypoints = self.L.orb_extractor_max_keypoints(self.L.orb_multi_handle(self.h, 0))

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.L.orb_multi_destroy(self.h)
            self.h = C.c_void_p()

    __del__ = close

    def set_pattern(self, pattern):
        pattern = np.ascontiguousarray(pattern, np.int8)
        _check(self.L.orb_multi_set_pattern(self.h, _p(pattern)))

    def extract_batch(self, imgs):
        imgs = np.ascontiguousarray(imgs, dtype=np.uint8)
        f, rows, cols = imgs.shape
        cap = self.max_keypoints
        kps = np.zeros((f, cap), KP_DTYPE)
        desc = np.zeros((f, cap, 32), np.uint8)
        counts = np.zeros(f, np.int32)
        _check(self.L.orb_multi_extract_batch(self.h, _p(imgs), f, rows, cols, imgs.strides[1], imgs.strides[0], _p(kps), _p(desc),
                                              cap, _p(counts)))
        return [(kps[i, :counts[i]].copy(), desc[i, :counts[i]].copy()) for i in range(f)]


class MultiKeyframeDB:
    """A keyframe database sharded by keyframe over `devices` (orb_multi_db_*): host arrays desc (n_kf, cap, 32) u8,
    kps (n_kf, cap) KP_DTYPE, valid (n_kf, cap) u8 or None, counts (n_kf) i32, node_of (n_kf, cap) u16."""

    def __init__(self, devices, desc, kps, valid, counts, node_of, n_nodes):
        self.L = lib()
        desc = np.ascontiguousarray(desc, np.uint8); kps = np.ascontiguousarray(kps)
        counts = np.ascontiguousarray(counts, np.int32); node_of = np.ascontiguousarray(node_of, np.uint16)
        valid = None if valid is None else np.ascontiguousarray(valid, np.uint8)
        self.n_kf, self.cap = desc.shape[0], desc.shape[1]
        devs = np.asarray(devices, np.int32)
        h = C.c_void_p()
        _check(self.L.orb_multi_db_create(_p(devs), len(devs), _p(desc), _p(kps), _p(valid) if valid is not None else None,
                                          _p(counts), _p(node_of), self.n_kf, self.cap, n_nodes, C.byref(h)))
        self.h = h

    @property
    def shards(self):
        return self.L.orb_multi_db_shards(self.h)

    def match(self, q_desc, q_kps, q_node_of, ratio=0.7, check_ori=True):
        q_desc = np.ascontiguousarray(q_desc, np.uint8); q_kps = np.ascontiguousarray(q_kps)
        q_node_of = np.ascontiguousarray(q_node_of, np.uint16)
        match = np.full((self.n_kf, self.cap), -7, np.int32)
        nm = np.full(self.n_kf, -7, np.int32)
        _check(self.L.orb_multi_match_bow_batch(self.h, _p(q_desc), _p(q_kps), len(q_kps), _p(q_node_of), ratio, int(check_ori),
                                                _p(match), _p(nm)))
        return match, nm

    def close(self):
        if self.h:
            self.L.orb_multi_db_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _fv(node_ids, offsets, indices):
    node_ids = np.ascontiguousarray(node_ids, np.uint32)
    offsets = np.ascontiguousarray(offsets, np.int32)
    indices = np.ascontiguousarray(indices, np.int32)
    s = FeatVecC(node_ids.ctypes.data, offsets.ctypes.data, indices.ctypes.data, node_ids.shape[0])
    s._keep = (node_ids, offsets, indices)
    return s


class Matcher:
    """Mirror of the descriptor-matching core of ORB_SLAM2::ORBmatcher (reference include/ORBmatcher.h:43-100)."""

    TH_LOW, TH_HIGH, HISTO_LENGTH = 50, 100, 30

    def __init__(self, nnratio=0.6, check_ori=True, device=0):
        self.L = lib()
        self.h = C.c_void_p()
        self.nnratio = float(nnratio)
        self.check_ori = bool(check_ori)
        _check(self.L.orb_matcher_create(device, C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.L.orb_matcher_destroy(self.h)
            self.h = C.c_void_p()

    __del__ = close

    def sync(self):
        _check(self.L.orb_matcher_sync(self.h))

    @property
    def stream(self):
        return self.L.orb_matcher_stream(self.h)

    def wait_for(self, hip_stream):
        _check(self.L.orb_matcher_wait_for(self.h, C.c_void_p(hip_stream)))

    def search_by_bow(self, desc_kf, angle_kf, valid_kf, fv_kf, desc_f, angle_f, fv_f):
        """fv_* = (node_ids, offsets, indices).  Returns (nmatches, match_f)."""
        desc_kf = np.ascontiguousarray(desc_kf, np.uint8); desc_f = np.ascontiguousarray(desc_f, np.uint8)
        angle_kf = np.ascontiguousarray(angle_kf, np.float32); angle_f = np.ascontiguousarray(angle_f, np.float32)
        valid_kf = np.ascontiguousarray(valid_kf, np.uint8)
        n_kf, n_f = desc_kf.shape[0], desc_f.shape[0]
        out = np.full(max(n_f, 1), -1, np.int32)
        nm = C.c_int(0)
        a, b = _fv(*fv_kf), _fv(*fv_f)
        _check(self.L.orb_match_bow(self.h, _p(desc_kf), _p(angle_kf), _p(valid_kf), n_kf, C.byref(a), _p(desc_f),
                                    _p(angle_f), n_f, C.byref(b), self.nnratio, int(self.check_ori), _p(out), C.byref(nm)))
        return nm.value, out[:n_f]

    def search_by_bow_kk(self, d1, a1, v1, fv1, d2, a2, v2, fv2):
        d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
        a1 = np.ascontiguousarray(a1, np.float32); a2 = np.ascontiguousarray(a2, np.float32)
        v1 = np.ascontiguousarray(v1, np.uint8); v2 = np.ascontiguousarray(v2, np.uint8)
        n1, n2 = d1.shape[0], d2.shape[0]
        out = np.full(max(n1, 1), -1, np.int32)
        nm = C.c_int(0)
        a, b = _fv(*fv1), _fv(*fv2)
        _check(self.L.orb_match_bow_kk(self.h, _p(d1), _p(a1), _p(v1), n1, C.byref(a), _p(d2), _p(a2), _p(v2), n2,
                                       C.byref(b), self.nnratio, int(self.check_ori), _p(out), C.byref(nm)))
        return nm.value, out[:n1]

    def search_for_initialization(self, k1, d1, k2, d2, grid, prev_xy, window=10):
        k1 = np.ascontiguousarray(k1); k2 = np.ascontiguousarray(k2)
        d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
        grid = np.ascontiguousarray(grid, np.float32)
        assert prev_xy.dtype == np.float32 and prev_xy.flags.c_contiguous
        n1, n2 = k1.shape[0], k2.shape[0]
        out = np.full(max(n1, 1), -1, np.int32)
        nm = C.c_int(0)
        _check(self.L.orb_match_init(self.h, _p(k1), _p(d1), n1, _p(k2), _p(d2), n2, _p(grid), _p(prev_xy), window,
                                     self.nnratio, int(self.check_ori), _p(out), C.byref(nm)))
        return nm.value, out[:n1]

    def search_by_projection(self, mode, q, q_desc, q_angle, kps_un, desc, u_right, occupied, grid, max_dist=100):
        """mode 0: SearchByProjection(CurrentFrame, LastFrame, ...); mode 1: SearchByProjection(Frame, MapPoints, ...)."""
        q = np.ascontiguousarray(q, PROJ_DTYPE); q_desc = np.ascontiguousarray(q_desc, np.uint8)
        q_angle = np.ascontiguousarray(q_angle, np.float32)
        kps_un = np.ascontiguousarray(kps_un); desc = np.ascontiguousarray(desc, np.uint8)
        u_right = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
        occupied = np.ascontiguousarray(occupied, np.uint8)
        grid = np.ascontiguousarray(grid, np.float32)
        n = kps_un.shape[0]
        out = np.full(max(n, 1), -1, np.int32)
        nm = C.c_int(0)
        _check(self.L.orb_match_projection(self.h, mode, _p(q), _p(q_desc), _p(q_angle), q.shape[0], _p(kps_un), _p(desc),
                                           _p(u_right), _p(occupied), n, _p(grid), self.nnratio, int(max_dist),
                                           int(self.check_ori), _p(out), C.byref(nm)))
        return nm.value, out[:n]

    def search_by_projection_best(self, q, q_desc, kps_un, desc, u_right, grid, max_dist=50, chi2=False, inv_sigma2=None):
        """Independent best candidate per projected point (search loops of Fuse x2 / SearchBySim3)."""
        q = np.ascontiguousarray(q, PROJ_DTYPE); q_desc = np.ascontiguousarray(q_desc, np.uint8)
        kps_un = np.ascontiguousarray(kps_un); desc = np.ascontiguousarray(desc, np.uint8)
        u_right = None if u_right is None else np.ascontiguousarray(u_right, np.float32)
        inv_sigma2 = None if inv_sigma2 is None else np.ascontiguousarray(inv_sigma2, np.float32)
        grid = np.ascontiguousarray(grid, np.float32)
        nq = q.shape[0]
        best = np.full(max(nq, 1), -1, np.int32)
        dist = np.full(max(nq, 1), 256, np.int32)
        _check(self.L.orb_match_projection_best(self.h, _p(q), _p(q_desc), nq, _p(kps_un), _p(desc), _p(u_right), kps_un.shape[0],
                                                _p(grid), int(max_dist), int(chi2), _p(inv_sigma2),
                                                0 if inv_sigma2 is None else inv_sigma2.shape[0], _p(best), _p(dist)))
        return best[:nq], dist[:nq]

    def search_for_triangulation(self, k1, d1, mp1, ur1, fv1, k2, d2, mp2, ur2, fv2, F12, ex, ey, sf2, sig2, only_stereo=False):
        k1 = np.ascontiguousarray(k1); k2 = np.ascontiguousarray(k2)
        d1 = np.ascontiguousarray(d1, np.uint8); d2 = np.ascontiguousarray(d2, np.uint8)
        mp1 = np.ascontiguousarray(mp1, np.uint8); mp2 = np.ascontiguousarray(mp2, np.uint8)
        ur1 = None if ur1 is None else np.ascontiguousarray(ur1, np.float32)
        ur2 = None if ur2 is None else np.ascontiguousarray(ur2, np.float32)
        F12 = np.ascontiguousarray(F12, np.float32); sf2 = np.ascontiguousarray(sf2, np.float32)
        sig2 = np.ascontiguousarray(sig2, np.float32)
        n1 = k1.shape[0]
        out = np.full(max(n1, 1), -1, np.int32)
        nm = C.c_int(0)
        a, b = _fv(*fv1), _fv(*fv2)
        _check(self.L.orb_match_triangulation(self.h, _p(k1), _p(d1), _p(mp1), _p(ur1), n1, C.byref(a), _p(k2), _p(d2), _p(mp2),
                                              _p(ur2), k2.shape[0], C.byref(b), _p(F12), float(ex), float(ey), _p(sf2), _p(sig2),
                                              sf2.shape[0], int(only_stereo), int(self.check_ori), _p(out), C.byref(nm)))
        return nm.value, out[:n1]

    def distinctive_descriptors(self, desc, offsets):
        """MapPoint::ComputeDistinctiveDescriptors for a batch: desc rows offsets[p]:offsets[p+1] belong to point p."""
        desc = np.ascontiguousarray(desc, np.uint8); offsets = np.ascontiguousarray(offsets, np.int32)
        n = offsets.shape[0] - 1
        out = np.zeros(max(n, 1), np.int32)
        _check(self.L.orb_distinctive_descriptors(self.h, _p(desc), _p(offsets), n, _p(out)))
        return out[:n]

    def bow_assign_device(self, d_desc, d_counts, n_frames, cap, d_centroids, d_node_of):
        _check(self.L.orb_bow_assign_device(self.h, C.c_void_p(d_desc), C.c_void_p(d_counts), n_frames, cap,
                                            C.c_void_p(d_centroids), C.c_void_p(d_node_of)))

    def build_csr_device(self, d_node_of, d_counts, n_frames, cap, n_nodes, d_keys, d_start, d_cnt):
        _check(self.L.orb_bow_build_csr_device(self.h, C.c_void_p(d_node_of), C.c_void_p(d_counts), n_frames, cap, n_nodes,
                                               C.c_void_p(d_keys), C.c_void_p(d_start), C.c_void_p(d_cnt)))

    def build_csr_desc_device(self, d_node_of, d_counts, d_desc, n_frames, cap, n_nodes, d_keys, d_start, d_cnt, d_csr_desc):
        _check(self.L.orb_bow_build_csr_desc_device(self.h, C.c_void_p(d_node_of), C.c_void_p(d_counts), C.c_void_p(d_desc), n_frames,
                                                    cap, n_nodes, C.c_void_p(d_keys), C.c_void_p(d_start), C.c_void_p(d_cnt),
                                                    C.c_void_p(d_csr_desc)))

    @staticmethod
    def _store(store):
        return FeatStoreC(store["desc"], store["kps"], store.get("valid") or None, store["counts"], store["node_of"],
                          store["cap"], store["n_frames"], store.get("n_nodes", 0), store.get("csr_keys") or None,
                          store.get("csr_start") or None, store.get("csr_cnt") or None, store.get("csr_desc") or None)

    def match_bow_query_device(self, store, d_kf_index, n_kf, d_f_index, n_queries, d_match, d_nmatches):
        """queries d_f_index[0..n_queries) against keyframes d_kf_index[0..n_kf): pair q * n_kf + k."""
        s = self._store(store)
        _check(self.L.orb_match_bow_query_device(self.h, C.byref(s), C.c_void_p(d_kf_index), n_kf, C.c_void_p(d_f_index), n_queries,
                                                 self.nnratio, int(self.check_ori), C.c_void_p(d_match), C.c_void_p(d_nmatches)))

    def bow_query_frames_device(self, vocab, store, first_query, n_queries, levelsup, d_kf_index, n_kf, d_f_index, d_match, d_nmatches):
        """ComputeBoW (descent + feature vector, written into the store) of frames [first_query, +n_queries) and their search
        against the keyframe list, one call (orb_bow_query_frames_device)."""
        s = self._store(store)
        _check(self.L.orb_bow_query_frames_device(self.h, vocab.h, C.byref(s), first_query, n_queries, levelsup, C.c_void_p(d_kf_index),
                                                  n_kf, C.c_void_p(d_f_index), self.nnratio, int(self.check_ori), C.c_void_p(d_match),
                                                  C.c_void_p(d_nmatches)))

    def match_bow_batch_device(self, store, d_kf_index, d_f_index, n_pairs, d_match, d_nmatches):
        """store = dict(desc=, kps=, valid=, counts=, node_of=, cap=, n_frames=) of raw device pointers."""
        s = self._store(store)
        _check(self.L.orb_match_bow_batch_device(self.h, C.byref(s), C.c_void_p(d_kf_index), C.c_void_p(d_f_index), n_pairs,
                                                 self.nnratio, int(self.check_ori), C.c_void_p(d_match),
                                                 C.c_void_p(d_nmatches)))


def stereo_match(ex_left, ex_right, k_l, d_l, k_r, d_r, mb, mbf):
    """== Frame::ComputeStereoMatches() on the pyramids the two extractor handles hold on the device."""
    k_l = np.ascontiguousarray(k_l); k_r = np.ascontiguousarray(k_r)
    d_l = np.ascontiguousarray(d_l, np.uint8); d_r = np.ascontiguousarray(d_r, np.uint8)
    n = k_l.shape[0]
    u = np.zeros(max(n, 1), np.float32)
    z = np.zeros(max(n, 1), np.float32)
    _check(lib().orb_stereo_match(ex_left.h, ex_right.h, _p(k_l), _p(d_l), n, _p(k_r), _p(d_r), k_r.shape[0],
                                  float(mb), float(mbf), _p(u), _p(z)))
    return u[:n], z[:n]


def stereo_match_device(ex_left, ex_right, frame_l, frame_r, d_kps_l, d_desc_l, n_l, d_kps_r, d_desc_r, n_r, mb, mbf,
                        d_u_right, d_depth):
    """Device-pointer form: frames frame_l / frame_r of the handles' last batches; asynchronous on the left handle's stream."""
    _check(lib().orb_stereo_match_device(ex_left.h, ex_right.h, frame_l, frame_r, C.c_void_p(d_kps_l), C.c_void_p(d_desc_l), n_l,
                                         C.c_void_p(d_kps_r), C.c_void_p(d_desc_r), n_r, C.c_float(mb), C.c_float(mbf),
                                         C.c_void_p(d_u_right), C.c_void_p(d_depth)))


def stereo_match_batch_device(ex_left, ex_right, first_l, first_r, n_pairs, d_kps_l, d_desc_l, d_counts_l, d_kps_r, d_desc_r,
                              d_counts_r, cap, mb, mbf, d_u_right, d_depth):
    """All pairs of the two handles' last batches in one launch; counts are read on the device."""
    _check(lib().orb_stereo_match_batch_device(ex_left.h, ex_right.h, first_l, first_r, n_pairs, C.c_void_p(d_kps_l),
                                               C.c_void_p(d_desc_l), C.c_void_p(d_counts_l), C.c_void_p(d_kps_r),
                                               C.c_void_p(d_desc_r), C.c_void_p(d_counts_r), cap, C.c_float(mb), C.c_float(mbf),
                                               C.c_void_p(d_u_right), C.c_void_p(d_depth)))


class Vocabulary:
    """A flattened DBoW2 vocabulary tree on the device (see orbhip.synth.synth_vocab_tree for the array layout)."""

    def __init__(self, tree, device=0):
        self.L = lib()
        self.h = C.c_void_p()
        nd = np.ascontiguousarray(tree["node_desc"], np.uint8)
        cb = np.ascontiguousarray(tree["child_begin"], np.int32)
        ch = np.ascontiguousarray(tree["children"], np.int32)
        wi = np.ascontiguousarray(tree["word_id"], np.int32)
        _check(self.L.orb_vocab_create(device, _p(nd), _p(cb), _p(ch), _p(wi), nd.shape[0], int(tree["L"]), C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.L.orb_vocab_destroy(self.h)
            self.h = C.c_void_p()

    __del__ = close

    def level_nodes(self, levelsup=4):
        return self.L.orb_vocab_level_nodes(self.h, levelsup)

    def transform(self, matcher, desc, levelsup=4):
        desc = np.ascontiguousarray(desc, np.uint8)
        n = desc.shape[0]
        word = np.zeros(max(n, 1), np.int32)
        node = np.zeros(max(n, 1), np.int32)
        _check(self.L.orb_bow_transform(matcher.h, self.h, _p(desc), n, levelsup, _p(word), _p(node)))
        return word[:n], node[:n]

    def transform_device(self, matcher, d_desc, d_counts, n_frames, cap, levelsup, d_word_of=0, d_node_id=0, d_node_of=0):
        _check(self.L.orb_bow_transform_device(matcher.h, self.h, C.c_void_p(d_desc), C.c_void_p(d_counts), n_frames, cap,
                                               levelsup, C.c_void_p(d_word_of) if d_word_of else None,
                                               C.c_void_p(d_node_id) if d_node_id else None,
                                               C.c_void_p(d_node_of) if d_node_of else None))
