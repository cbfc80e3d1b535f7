#!/usr/bin/env python3
"""Pins of the ORACLE's outputs: sha256 of what oracle/ computes for the first frames of every BASELINE.json
configuration (keypoints, descriptors, pyramid, feature vectors, SearchByBoW / SearchForInitialization match arrays,
stereo mvuRight / mvDepth) and of the synthetic inputs themselves.

What this is for: GPU parity tests compare the HIP kernels with the oracle, and oracle and kernels share two headers
(orb_sincos.h, orb_brief_pattern.h) -- a change that moved both together would pass every parity test.  These pins
catch that drift.  What this is NOT: a pin against the reference's OpenCV / DBoW2 (the reference ships no fixtures and
cannot be built here; parity stays "unpinned" in that sense).

usage: python tests/golden/make_oracle_pins.py [--write]     (without --write: print and compare)"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "orb-slam2-chinesenotes_amd", "pyhost"))
import oracle  # noqa: E402
from orbhip import synth  # noqa: E402

PINS = os.path.join(HERE, "oracle_pins.json")
MBF = 386.1448
MB = MBF / 718.856


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def compute():
    pins = {}
    # ---- configs 1 / 2 / 4: 640x480, nFeatures 1000, 8 levels (frames 0 and 1 of the benchmark sequence, frame 0 of the
    # classic generator)
    ref = oracle.Extractor(1000, 1.2, 8, 20, 7)
    f0 = synth.synth_frame(0)
    k, d = ref.extract(f0)
    pins["c1_input_frame0"] = sha(f0)
    pins["c1_keypoints"] = sha(k)
    pins["c1_descriptors"] = sha(d)
    pins["c1_count"] = int(len(k))
    pins["c1_pyramid"] = sha(*[ref.pyramid_level(l) for l in range(8)])
    kept, cands = ref.level_counts()
    pins["c1_level_counts"] = [kept.tolist(), cands.tolist()]
    seq = synth.synth_sequence(0, 2)
    (ka, da), (kb, db) = ref.extract(seq[0]), ref.extract(seq[1])
    pins["c4_input_seq0_1"] = sha(seq)
    pins["c4_keypoints_desc_seq0"] = sha(ka, da)
    pins["c4_keypoints_desc_seq1"] = sha(kb, db)
    cent = synth.synth_vocabulary()
    fva, fvb = oracle.bow_transform(da, cent), oracle.bow_transform(db, cent)
    pins["c4_featvec_seq0"] = sha(fva.node_ids, fva.offsets, fva.indices)
    valid = synth.synth_valid_flags(len(ka), 0)
    nm, m = oracle.search_by_bow(da, ka["angle"], valid, fva, db, kb["angle"], fvb, 0.7, True)
    pins["c4_search_by_bow"] = [int(nm), sha(m)]
    tree = synth.synth_vocab_tree_balanced(10, 4, seed=77)
    w, nid = oracle.vocab_transform(tree, da, 2)
    pins["c4_vocab_transform"] = sha(w, nid)
    grid = (0.0, 0.0, 64.0 / 640.0, 48.0 / 480.0)
    prev = np.ascontiguousarray(np.stack([ka["x"], ka["y"]], axis=1), dtype=np.float32)
    ni, mi = oracle.search_for_init(ka, da, kb, db, grid, prev, 100, 0.9, True)
    pins["c4_search_for_initialization"] = [int(ni), sha(mi), sha(prev)]
    # ---- config 3: 1241x376 stereo pair, nFeatures 2000
    rl, rr = oracle.Extractor(2000), oracle.Extractor(2000)
    left, right = synth.synth_frame(100, 1241, 376), synth.synth_stereo_right(100, 1241, 376)
    kl, dl = rl.extract(left)
    kr, dr = rr.extract(right)
    u, z = oracle.stereo_matches(rl, rr, kl, dl, kr, dr, MB, MBF)
    pins["c3_inputs"] = sha(left, right)
    pins["c3_left"] = sha(kl, dl)
    pins["c3_right"] = sha(kr, dr)
    pins["c3_stereo_uright_depth"] = [int((u >= 0).sum()), sha(u, z)]
    # ---- config 5: 752x480 stream frame
    r5 = oracle.Extractor(1000)
    f5 = synth.synth_sequence(0, 1, 752, 480)[0]
    k5, d5 = r5.extract(f5)
    pins["c5_input"] = sha(f5)
    pins["c5_keypoints_desc"] = [int(len(k5)), sha(k5, d5)]
    return pins


def main():
    pins = compute()
    if "--write" in sys.argv:
        json.dump(pins, open(PINS, "w"), indent=1, sort_keys=True)
        print("wrote", PINS)
        return 0
    old = json.load(open(PINS))
    bad = [k for k in sorted(set(old) | set(pins)) if old.get(k) != pins.get(k)]
    print(json.dumps(pins, indent=1, sort_keys=True))
    print("DIFFERS from the committed pins: %s" % bad if bad else "identical to the committed pins")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
