"""GPU parity of the vocabulary-tree descent (DBoW2 transform, FeatureVector/word part) and of the whole
device-resident chain extract -> transform -> SearchByBoW with a deeper, unbalanced, id-shuffled tree."""
import numpy as np
import pytest

import oracle
from orbhip import capi, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("k,L,levelsup", [(10, 3, 1), (10, 3, 2), (5, 4, 2), (10, 2, 4), (3, 6, 4), (16, 3, 1), (17, 3, 1), (40, 2, 1)])
def test_vocab_transform_matches_oracle(k, L, levelsup):
    tree = synth.synth_vocab_tree(k, L, seed=k * 100 + L)
    rng = np.random.default_rng(L)
    desc = rng.integers(0, 256, (1500, 32), dtype=np.uint8)
    ww, wn = oracle.vocab_transform(tree, desc, levelsup)
    m = capi.Matcher()
    v = capi.Vocabulary(tree)
    gw, gn = v.transform(m, desc, levelsup)
    assert np.array_equal(gw, ww) and np.array_equal(gn, wn)
    assert gw.min() >= 0 and len(set(gw.tolist())) > 20


def test_extract_transform_match_chain_on_device():
    import torch
    tree = synth.synth_vocab_tree(10, 3, seed=77, prune=0.15)
    levelsup = 1                                                      # FeatureVector nodes at depth 2 (~90 nodes)
    F, W, H = 4, 640, 480
    base = synth.synth_frame(60, noise=0).astype(np.int16)
    frames = np.stack([np.clip(base + (synth.splitmix64(5 + i, 0, base.size) % np.uint64(13)).astype(np.int16).reshape(base.shape) - 6,
                               0, 255).astype(np.uint8) for i in range(F)])
    ex, mt, voc = capi.Extractor(), capi.Matcher(0.7, True), capi.Vocabulary(tree)
    K = voc.level_nodes(levelsup)
    cap = ex.max_keypoints
    dev = torch.device("cuda", 0)
    d_imgs = torch.from_numpy(frames).to(dev)
    d_kps = torch.zeros(F * cap * 28, dtype=torch.uint8, device=dev)
    d_desc = torch.zeros(F * cap * 32, dtype=torch.uint8, device=dev)
    d_counts = torch.zeros(F, dtype=torch.int32, device=dev)
    d_node = torch.zeros(F * cap, dtype=torch.int16, device=dev)
    d_nodeid = torch.zeros(F * cap, dtype=torch.int32, device=dev)
    d_word = torch.zeros(F * cap, dtype=torch.int32, device=dev)
    valid_np = np.stack([synth.synth_valid_flags(cap, 40 + i) for i in range(F)])
    d_valid = torch.from_numpy(valid_np).to(dev)
    pairs = [(a, b) for a in range(F) for b in range(F) if a != b]
    kf_idx = torch.tensor([p[0] for p in pairs], dtype=torch.int32, device=dev)
    f_idx = torch.tensor([p[1] for p in pairs], dtype=torch.int32, device=dev)
    d_match = torch.zeros(len(pairs) * cap, dtype=torch.int32, device=dev)
    d_nm = torch.zeros(len(pairs), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ex.extract_batch_device(d_imgs.data_ptr(), F, H, W, W, W * H, d_kps.data_ptr(), d_desc.data_ptr(), cap, d_counts.data_ptr())
    mt.wait_for(ex.stream)
    voc.transform_device(mt, d_desc.data_ptr(), d_counts.data_ptr(), F, cap, levelsup, d_word.data_ptr(), d_nodeid.data_ptr(),
                         d_node.data_ptr())
    store = dict(desc=d_desc.data_ptr(), kps=d_kps.data_ptr(), valid=d_valid.data_ptr(), counts=d_counts.data_ptr(),
                 node_of=d_node.data_ptr(), cap=cap, n_frames=F, n_nodes=K)
    mt.match_bow_batch_device(store, kf_idx.data_ptr(), f_idx.data_ptr(), len(pairs), d_match.data_ptr(), d_nm.data_ptr())
    ex.sync(); mt.sync(); torch.cuda.synchronize()
    counts = d_counts.cpu().numpy()
    kps = d_kps.cpu().numpy().view(capi.KP_DTYPE).reshape(F, cap)
    desc = d_desc.cpu().numpy().reshape(F, cap, 32)
    nodeid = d_nodeid.cpu().numpy().reshape(F, cap)
    word = d_word.cpu().numpy().reshape(F, cap)
    match = d_match.cpu().numpy().reshape(len(pairs), cap)
    nm = d_nm.cpu().numpy()
    feats = []
    for i in range(F):
        n = counts[i]
        ww, wn = oracle.vocab_transform(tree, desc[i, :n], levelsup)
        assert np.array_equal(word[i, :n], ww) and np.array_equal(nodeid[i, :n], wn)
        ids = sorted(set(int(v) for v in wn if v >= 0))                # FeatureVector: ascending node ids, ascending indices
        offs, idx = [0], []
        for nid in ids:
            idx += np.nonzero(wn == nid)[0].tolist()
            offs.append(len(idx))
        feats.append((kps[i, :n], desc[i, :n], oracle.FeatVec(ids, offs, idx)))
    total = 0
    for p, (a, b) in enumerate(pairs):
        (ka, da, fa), (kb, db, fb) = feats[a], feats[b]
        wn_, wm = oracle.search_by_bow(da, ka["angle"], valid_np[a][:len(ka)], fa, db, kb["angle"], fb, 0.7, True)
        assert nm[p] == wn_ and np.array_equal(match[p, :len(kb)], wm), (a, b)
        total += wn_
    assert total > 50 * len(pairs)


def test_vocab_transform_randomized_trees_and_sizes():
    """Tree descent on seeded random trees (branching 2..10, depth 2..6, pruned and unpruned) with descriptor counts of
    0, 1, around the wave width and thousands, every levelsup the tree allows."""
    rng = np.random.default_rng(321)
    m = capi.Matcher()
    for t in range(14):
        k, L = int(rng.integers(2, 11)), int(rng.integers(2, 7))
        if k ** L > 200000:
            L = 4
        tree = synth.synth_vocab_tree(k, L, seed=1000 + t, prune=float(rng.choice([0.0, 0.1, 0.3])))
        v = capi.Vocabulary(tree)
        for n in (0, 1, 63, 64, 65, int(rng.integers(200, 3000))):
            desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
            for levelsup in sorted({0, 1, L - 1, L, int(rng.integers(0, L + 1))}):
                ww, wn = oracle.vocab_transform(tree, desc, levelsup)
                gw, gn = v.transform(m, desc, levelsup)
                assert np.array_equal(gw, ww) and np.array_equal(gn, wn), (t, k, L, n, levelsup)
        v.close()


@pytest.mark.parametrize("k,L", [(10, 4), (12, 3), (16, 3), (7, 5)])
def test_vocab_transform_device_batches_with_ragged_counts(k, L):
    """orb_bow_transform_device over a batch of 20 frames with ragged per-frame counts (0, 1, around the wave width, the full
    capacity), a quarter of the descriptors exact copies of node descriptors (distance ties): word, node id and compact
    node index against the oracle; slots past a frame's count keep 0xFFFF / stay untouched."""
    import torch
    tree = synth.synth_vocab_tree(k, L, seed=4000 + k, prune=0.1)
    levelsup = 1
    F, cap = 20, 1003
    rng = np.random.default_rng(k * 7 + L)
    counts = np.array([0, 1, 3, 63, 64, 65, cap, cap - 1, 500, 777] + [int(x) for x in rng.integers(0, cap + 1, F - 10)], np.int32)
    desc = rng.integers(0, 256, (F, cap, 32), dtype=np.uint8)
    # ties on purpose: a quarter of the descriptors are exact copies of tree nodes' descriptors
    nd = np.asarray(tree["node_desc"], np.uint8).reshape(-1, 32)
    pick = rng.integers(0, nd.shape[0], (F, cap))
    copy = rng.random((F, cap)) < 0.25
    desc[copy] = nd[pick[copy]]
    dev = torch.device("cuda:0")
    dD = torch.from_numpy(desc).to(dev)
    dC = torch.from_numpy(counts).to(dev)
    dW = torch.full((F, cap), -7, dtype=torch.int32, device=dev)
    dN = torch.full((F, cap), -7, dtype=torch.int32, device=dev)
    dO = torch.full((F, cap), 0x1234, dtype=torch.int16, device=dev)
    torch.cuda.synchronize()                                          # (torch's fills are on its own stream)
    m = capi.Matcher()
    v = capi.Vocabulary(tree)
    v.transform_device(m, dD.data_ptr(), dC.data_ptr(), F, cap, levelsup, dW.data_ptr(), dN.data_ptr(), dO.data_ptr())
    m.sync()
    gW, gN, gO = dW.cpu().numpy(), dN.cpu().numpy(), dO.cpu().numpy().view(np.uint16)
    for f in range(F):
        n = int(counts[f])
        ww, wn = oracle.vocab_transform(tree, desc[f, :n], levelsup)
        assert np.array_equal(gW[f, :n], ww) and np.array_equal(gN[f, :n], wn), (f, n)
        assert (gO[f, n:] == 0xFFFF).all() and (gW[f, n:] == -7).all()
        # compact index: ascending node id over the nodes of that level
        if n:
            reached = gN[f, :n] >= 0                                  # (a pruned branch can end above that level: node id -1)
            assert ((gO[f, :n] != 0xFFFF) == reached).all()
            order = np.argsort(gN[f, :n][reached], kind="stable")
            assert (np.diff(gO[f, :n][reached][order].astype(np.int64)) >= 0).all()
    v.close()


def test_compute_bow_and_search_in_one_call_equals_the_three_calls():
    """orb_bow_query_frames_device (Frame::ComputeBoW of the query frames -- descent + feature vector, written into the store --
    and the search against the keyframe list, reference src/Tracking.cc:1471-1492) against the same work as three calls
    (orb_bow_transform_device, orb_bow_build_csr_desc_device, orb_match_bow_query_device): identical store rows and matches."""
    import torch
    tree = synth.synth_vocab_tree(10, 3, seed=91, prune=0.1)
    levelsup = 1
    dev = torch.device("cuda:0")
    n_kf, n_q, cap = 30, 3, 500
    F = n_kf + n_q
    rng = np.random.default_rng(8)
    base = rng.integers(0, 256, (n_q, cap, 32), dtype=np.uint8)
    desc = np.zeros((F, cap, 32), np.uint8)
    counts = np.zeros(F, np.int32)
    for f in range(F):
        src = base[f % n_q] if f < n_kf else base[f - n_kf]
        n = cap if f % 5 else cap - 37
        d = src[:n].copy()
        flip = rng.integers(0, 256, (n, 3))
        for r in range(n):
            for b in flip[r][: int(rng.integers(0, 4))]:
                d[r, b >> 3] ^= np.uint8(1 << (b & 7))
        desc[f, :n], counts[f] = d, n
    kps = np.zeros((F, cap), capi.KP_DTYPE)
    kps["angle"] = rng.uniform(0, 360, (F, cap)).astype(np.float32)
    m, v = capi.Matcher(0.75, True), capi.Vocabulary(tree)
    nn = v.level_nodes(levelsup)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    results = []
    for fused in (False, True):
        d_desc, d_kps, d_counts = t(desc), t(kps.view(np.uint8)), t(counts)
        z = lambda n, dt: torch.zeros(n, dtype=dt, device=dev)
        d_node, d_ck, d_cs, d_cc, d_cd = z(F * cap, torch.int16), z(F * cap, torch.int32), z(F * nn, torch.int16), z(F * nn, torch.int16), z(F * cap * 32, torch.uint8)
        store = dict(desc=d_desc.data_ptr(), kps=d_kps.data_ptr(), valid=0, counts=d_counts.data_ptr(), node_of=d_node.data_ptr(), cap=cap,
                     n_frames=F, n_nodes=nn, csr_keys=d_ck.data_ptr(), csr_start=d_cs.data_ptr(), csr_cnt=d_cc.data_ptr(), csr_desc=d_cd.data_ptr())
        torch.cuda.synchronize()
        # the keyframes' ComputeBoW (once per keyframe in the reference): the plain calls in both runs
        v.transform_device(m, d_desc.data_ptr(), d_counts.data_ptr(), n_kf, cap, levelsup, d_node_of=d_node.data_ptr())
        m.build_csr_desc_device(d_node.data_ptr(), d_counts.data_ptr(), d_desc.data_ptr(), n_kf, cap, nn, d_ck.data_ptr(), d_cs.data_ptr(),
                                d_cc.data_ptr(), d_cd.data_ptr())
        d_kf, d_f = t(np.arange(n_kf, dtype=np.int32)), t(np.arange(n_kf, F, dtype=np.int32))
        d_m, d_n = z(n_q * n_kf * cap, torch.int32), z(n_q * n_kf, torch.int32)
        torch.cuda.synchronize()
        if fused:
            m.bow_query_frames_device(v, store, n_kf, n_q, levelsup, d_kf.data_ptr(), n_kf, d_f.data_ptr(), d_m.data_ptr(), d_n.data_ptr())
        else:
            o = n_kf
            v.transform_device(m, d_desc.data_ptr() + o * cap * 32, d_counts.data_ptr() + o * 4, n_q, cap, levelsup, d_node_of=d_node.data_ptr() + o * cap * 2)
            m.build_csr_desc_device(d_node.data_ptr() + o * cap * 2, d_counts.data_ptr() + o * 4, d_desc.data_ptr() + o * cap * 32, n_q, cap, nn,
                                    d_ck.data_ptr() + o * cap * 4, d_cs.data_ptr() + o * nn * 2, d_cc.data_ptr() + o * nn * 2, d_cd.data_ptr() + o * cap * 32)
            m.match_bow_query_device(store, d_kf.data_ptr(), n_kf, d_f.data_ptr(), n_q, d_m.data_ptr(), d_n.data_ptr())
        m.sync(); torch.cuda.synchronize()
        results.append([x.cpu().numpy() for x in (d_node, d_ck, d_cs, d_cc, d_cd, d_m, d_n)])
    for a, b in zip(*results):
        assert np.array_equal(a, b)
    nm = results[1][6].reshape(n_q, n_kf)
    assert (nm >= 0).all() and nm.max() > 50                          # related frames do match
    v.close()
