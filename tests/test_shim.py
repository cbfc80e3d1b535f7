"""The C++ drop-in shims (orb-slam2-chinesenotes_amd/host/: ORBextractor with the reference's class
signature, four ORBmatcher member functions) built against test doubles of the cv:: / SLAM types.
CPU: they compile and link against liborbhip.so.  GPU: driven the way Frame.cc / Tracking.cc drive the
reference classes, results bit-exact vs the CPU oracle."""
import os
import struct
import subprocess

import numpy as np
import pytest

import oracle
from orbhip import capi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SUP = os.path.join(ROOT, "tests", "support")
DRIVER = os.path.join(SUP, "shim_driver")
PKG = os.path.join(ROOT, "orb-slam2-chinesenotes_amd")


def build_driver():
    capi.build_library()
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I" + SUP, "-I" + os.path.join(SUP, "mini_slam"),
           "-I" + os.path.join(PKG, "host"), "-I" + os.path.join(ROOT, "include"),
           os.path.join(SUP, "shim_driver.cpp"), os.path.join(PKG, "host", "ORBextractor.cc"),
           os.path.join(PKG, "host", "ORBmatcherHip.cc"), "-L" + PKG, "-lorbhip", "-Wl,-rpath," + PKG, "-o", DRIVER]
    subprocess.check_call(cmd)
    return DRIVER


def test_shims_compile_and_link():
    assert os.path.exists(build_driver())


TOOL = os.path.join(SUP, "orb_batch_tool")


def build_batch_tool():
    capi.build_library()
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(PKG, "host", "orb_batch_tool.cc"), "-L" + PKG, "-lorbhip", "-Wl,-rpath," + PKG, "-o", TOOL])
    return TOOL


def test_batch_tool_compiles_and_links():
    assert os.path.exists(build_batch_tool())


@pytest.mark.gpu
def test_batch_tool_multi_device_cpp_host(tmp_path):
    """host/orb_batch_tool.cc: the C++ host program of the batched-frames mode (orb_multi_* over the C ABI, pinned
    buffers from orb_host_alloc) on 37 frames with the GPU listed three times; every frame bit-exact vs the oracle."""
    import json
    build_batch_tool()
    n, W, H = 37, 400, 300
    imgs = synth.synth_sequence(700, n, W, H)
    raw = tmp_path / "frames.raw"
    imgs.tofile(raw)
    out = str(tmp_path / "o")
    line = subprocess.check_output([TOOL, str(raw), str(W), str(H), str(n), "600", out, "0,0,0"], text=True)
    info = json.loads(line.strip().splitlines()[-1])
    assert info["frames"] == n and info["devices"] == 3 and info["frames_per_s"] > 0
    counts = np.fromfile(out + ".counts", dtype=np.int32)
    kps = np.fromfile(out + ".kps", dtype=oracle.KP_DTYPE)
    desc = np.fromfile(out + ".desc", dtype=np.uint8).reshape(-1, 32)
    ref = oracle.Extractor(600)
    off = 0
    for i in range(n):
        rk, rd = ref.extract(imgs[i])
        assert counts[i] == len(rk)
        assert kps[off:off + len(rk)].tobytes() == rk.tobytes() and np.array_equal(desc[off:off + len(rk)], rd), i
        off += len(rk)
    assert off == len(kps) == info["keypoints"]


def test_matcher_binding_without_editing_the_reference_source(tmp_path):
    """INTEGRATION.md section 2, the no-edit recipe: an object that defines the four replaced ORBmatcher functions itself
    (standing in for the reference's src/ORBmatcher.cc compiled unchanged) gets those four symbols weakened by
    tools/weaken_matcher_symbols.sh and is linked beside ORBmatcherHip.o: the strong HIP-backed definitions win (checked
    with DescriptorDistance, which is host-only), everything else in the object stays."""
    capi.build_library()
    inc = ["-I" + SUP, "-I" + os.path.join(SUP, "mini_slam"), "-I" + os.path.join(PKG, "host"), "-I" + os.path.join(ROOT, "include")]
    ref_o, hip_o, main_cpp, exe = (str(tmp_path / n) for n in ("ref.o", "hip.o", "main.cpp", "bind"))
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-c"] + inc + [os.path.join(SUP, "unedited_matcher_standin.cpp"), "-o", ref_o])
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-c"] + inc + [os.path.join(PKG, "host", "ORBmatcherHip.cc"), "-o", hip_o])
    open(main_cpp, "w").write(
        '#include <cstdio>\n#include "ORBmatcher.h"\nextern "C" int standin_not_replaced();\n'
        'int main() { cv::Mat a(1, 32, CV_8UC1), b(1, 32, CV_8UC1);\n'
        '  for (int i = 0; i < 32; i++) { a.data[i] = (unsigned char)i; b.data[i] = (unsigned char)(i ^ 0x0f); }\n'
        '  std::printf("%d %d\\n", ORB_SLAM2::ORBmatcher::DescriptorDistance(a, b), standin_not_replaced()); return 0; }\n')
    # without the weakening the link must fail (duplicate definitions) ...
    link = ["g++", "-std=c++17"] + inc + [main_cpp, ref_o, hip_o, "-L" + PKG, "-lorbhip", "-Wl,-rpath," + PKG, "-o", exe]
    assert subprocess.run(link, capture_output=True).returncode != 0
    # ... with it the HIP-backed functions are the ones linked in
    subprocess.check_call([os.path.join(ROOT, "tools", "weaken_matcher_symbols.sh"), ref_o])
    subprocess.check_call(link)
    out = subprocess.check_output([exe], text=True).split()
    assert out == ["128", "777"], out                           # 32 bytes x 4 differing bits, not the stand-in's -12345


@pytest.mark.gpu
def test_extractor_shim_matches_oracle(tmp_path):
    build_driver()
    img = synth.synth_frame(11)
    wide = np.zeros((480, 704), np.uint8)
    wide[:, :640] = img
    raw = tmp_path / "in.raw"
    wide.tofile(raw)
    out = str(tmp_path / "o")
    subprocess.check_call([DRIVER, "extract", str(raw), "640", "480", "704", "1000", out])
    kps = np.fromfile(out + ".kps", dtype=oracle.KP_DTYPE)
    desc = np.fromfile(out + ".desc", dtype=np.uint8).reshape(-1, 32)
    ref = oracle.Extractor()
    rk, rd = ref.extract(img)
    assert kps.tobytes() == rk.tobytes() and np.array_equal(desc, rd)
    pyr = np.fromfile(out + ".pyr", dtype=np.uint8)
    want = np.concatenate([ref.pyramid_level(l).ravel() for l in range(8)])
    assert np.array_equal(pyr, want)                       # mvImagePyramid as Frame::ComputeStereoMatches reads it


@pytest.mark.gpu
def test_matcher_shim_matches_oracle(tmp_path):
    build_driver()
    ref = oracle.Extractor(2000, 1.2, 8, 20, 7)
    base = synth.synth_frame(50, noise=0).astype(np.int16)
    feats = []
    for s in range(2):
        nz = (synth.splitmix64(99 + s, 0, base.size) % np.uint64(13)).astype(np.int16).reshape(base.shape) - 6
        feats.append(ref.extract(np.clip(np.roll(base, 4 * s, axis=1) + nz, 0, 255).astype(np.uint8)))
    (k1, d1), (k2, d2) = feats
    cent = synth.synth_vocabulary()
    fv1, fv2 = oracle.bow_transform(d1, cent), oracle.bow_transform(d2, cent)

    def node_of(fv, n):
        out = np.full(n, -1, np.int32)
        for k, nid in enumerate(fv.node_ids):
            out[fv.indices[fv.offsets[k]:fv.offsets[k + 1]]] = nid
        return out

    # valid: 0 = no MapPoint, 1 = good MapPoint, 2 = MapPoint with isBad()
    rng = np.random.default_rng(4)
    v1 = rng.choice([0, 1, 1, 1, 2], len(k1)).astype(np.uint8)
    v2 = rng.choice([0, 1, 1, 1, 2], len(k2)).astype(np.uint8)
    ratio, ori, window = 0.8, 1, 100
    grid = (0.0, 0.0, 64.0 / 640.0, 48.0 / 480.0)
    blob = struct.pack("<iifii4f", len(k1), len(k2), ratio, ori, window, *grid)
    for k, d, v, fv in ((k1, d1, v1, fv1), (k2, d2, v2, fv2)):
        blob += k.tobytes() + d.tobytes() + v.tobytes() + node_of(fv, len(k)).tobytes()
    scene = tmp_path / "scene.bin"
    scene.write_bytes(blob)
    out = str(tmp_path / "m")
    subprocess.check_call([DRIVER, "match", str(scene), out])
    counts = np.fromfile(out + ".counts", dtype=np.int32)
    na, ma = oracle.search_by_bow(d1, k1["angle"], v1 == 1, fv1, d2, k2["angle"], fv2, ratio, True)
    nb, mb = oracle.search_by_bow_kk(d1, k1["angle"], v1 == 1, fv1, d2, k2["angle"], v2 == 1, fv2, ratio, True)
    prev = np.ascontiguousarray(np.stack([k1["x"], k1["y"]], axis=1), dtype=np.float32)
    nc, mc = oracle.search_for_init(k1, d1, k2, d2, grid, prev, window, ratio, True)
    assert counts.tolist() == [na, nb, nc, oracle.hamming(d1[0], d2[0])]
    assert na > 30 and nb > 20 and nc > 30
    assert np.array_equal(np.fromfile(out + ".bowkf", dtype=np.int32), ma)
    assert np.array_equal(np.fromfile(out + ".bowkk", dtype=np.int32), mb)
    assert np.array_equal(np.fromfile(out + ".init", dtype=np.int32), mc)
    assert np.fromfile(out + ".prev", dtype=np.float32).tobytes() == prev.tobytes()
