"""GPU parity tests of orb_match_bow_query_device (csrc/orb_matcher_query.hip): one query frame against many keyframes --
the candidate loop of reference src/Tracking.cc:1471-1492 over ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...)
(src/ORBmatcher.cc:552-687) -- against the oracle and against the pair kernel on the same feature store."""
import numpy as np
import pytest

import oracle
from orbhip import capi

pytestmark = pytest.mark.gpu


def _flip(rng, d, maxflip):
    d = d.copy()
    for r in range(d.shape[0]):
        for b in rng.integers(0, 256, rng.integers(0, maxflip + 1)):
            d[r, b >> 3] ^= np.uint8(1 << (b & 7))
    return d


def _store(rng, n_kf, n_q, cap, n_nodes, node_weights, reuse, maxflip, p_valid=0.7):
    """A synthetic feature store: n_q query frames (stored behind the keyframes) and n_kf keyframes whose features are
    noisy copies of a pool of the queries' features (so that features of a node compete for the same partner), with node
    ids drawn from `node_weights` (a few heavy nodes -> the 33..64 and > 64 paths); some features in no node at all."""
    F = n_kf + n_q
    desc = np.zeros((F, cap, 32), np.uint8)
    kps = np.zeros((F, cap), capi.KP_DTYPE)
    valid = np.zeros((F, cap), np.uint8)
    node = np.full((F, cap), 0xFFFF, np.uint16)
    counts = np.zeros(F, np.int32)
    p = np.asarray(node_weights, np.float64) / np.sum(node_weights)
    qd, qn = [], []
    for q in range(n_q):
        n = int(rng.integers(cap // 2, cap + 1))
        d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
        nd = rng.choice(n_nodes, n, p=p).astype(np.uint16)
        nd[rng.random(n) < 0.03] = 0xFFFF                          # descent ended above the level: in no feature vector
        f = n_kf + q
        desc[f, :n], node[f, :n], counts[f] = d, nd, n
        kps[f, :n]["angle"] = rng.uniform(0, 360, n).astype(np.float32)
        qd.append(d); qn.append(nd)
    for k in range(n_kf):
        n = int(rng.integers(0, cap + 1)) if k % 7 == 3 else int(rng.integers(cap // 2, cap + 1))
        q = int(rng.integers(0, n_q))
        nq = len(qd[q])
        pool = rng.integers(0, nq, max(1, int(nq * reuse)))
        src = pool[rng.integers(0, len(pool), n)]
        d = _flip(rng, qd[q][src], maxflip)
        nd = qn[q][src].copy()
        other = rng.random(n) < 0.15                               # unrelated features
        d[other] = rng.integers(0, 256, (int(other.sum()), 32), dtype=np.uint8)
        nd[other] = rng.choice(n_nodes, int(other.sum()), p=p).astype(np.uint16)
        desc[k, :n], node[k, :n], counts[k] = d, nd, n
        ang = kps[n_kf + q, src]["angle"] + rng.choice([0.0, 0.0, 90.0], n).astype(np.float32) + rng.uniform(-3, 3, n).astype(np.float32)
        kps[k, :n]["angle"] = (ang % np.float32(360.0)).astype(np.float32)
        valid[k, :n] = rng.random(n) < p_valid
    return desc, kps, valid, node, counts


def _run(mt, desc, kps, valid, node, counts, n_nodes, kf_list, q_list, use_valid=True):
    import torch
    dev = torch.device("cuda", 0)
    F, cap = counts.shape[0], desc.shape[1]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_desc, d_kps, d_valid, d_node, d_counts = t(desc), t(kps.view(np.uint8)), t(valid), t(node.view(np.int16)), t(counts)
    def z(n, dt):
        # torch fills on ITS stream; the library's streams are non-blocking and nothing orders them behind it: a fill still in
        # flight would overwrite what the library writes (seen once as rows of zeros in a result) -- wait for it here
        t = torch.zeros(n, dtype=dt, device=dev)
        torch.cuda.synchronize()
        return t
    d_ck, d_cs, d_cc, d_cd = z(F * cap, torch.int32), z(F * n_nodes, torch.int16), z(F * n_nodes, torch.int16), z(F * cap * 32, torch.uint8)
    mt.build_csr_desc_device(d_node.data_ptr(), d_counts.data_ptr(), d_desc.data_ptr(), F, cap, n_nodes, d_ck.data_ptr(),
                             d_cs.data_ptr(), d_cc.data_ptr(), d_cd.data_ptr())
    store = dict(desc=d_desc.data_ptr(), kps=d_kps.data_ptr(), valid=d_valid.data_ptr() if use_valid else 0,
                 counts=d_counts.data_ptr(), node_of=d_node.data_ptr(), cap=cap, n_frames=F, n_nodes=n_nodes,
                 csr_keys=d_ck.data_ptr(), csr_start=d_cs.data_ptr(), csr_cnt=d_cc.data_ptr(), csr_desc=d_cd.data_ptr())
    nk, nq = len(kf_list), len(q_list)
    d_kf, d_f = t(np.asarray(kf_list, np.int32)), t(np.asarray(q_list, np.int32))
    d_m, d_n = torch.full((nq * nk * cap,), 7, dtype=torch.int32, device=dev), z(nq * nk, torch.int32)
    mt.match_bow_query_device(store, d_kf.data_ptr(), nk, d_f.data_ptr(), nq, d_m.data_ptr(), d_n.data_ptr())
    # the pair kernel on the same store and the same pair list
    d_kf2 = t(np.tile(np.asarray(kf_list, np.int32), nq))
    d_f2 = t(np.repeat(np.asarray(q_list, np.int32), nk))
    d_m2, d_n2 = z(nq * nk * cap, torch.int32), z(nq * nk, torch.int32)
    mt.match_bow_batch_device(store, d_kf2.data_ptr(), d_f2.data_ptr(), nq * nk, d_m2.data_ptr(), d_n2.data_ptr())
    mt.sync(); torch.cuda.synchronize()
    return (d_m.cpu().numpy().reshape(nq, nk, cap), d_n.cpu().numpy().reshape(nq, nk),
            d_m2.cpu().numpy().reshape(nq, nk, cap), d_n2.cpu().numpy().reshape(nq, nk))


def _oracle_pair(desc, kps, valid, node, counts, kf, fq, ratio, ori, use_valid=True):
    na, nb = int(counts[kf]), int(counts[fq])
    nid = lambda f, n: np.where(node[f, :n] == 0xFFFF, -1, node[f, :n].astype(np.int64))
    fva, fvb = oracle.featvec_from_nodes(nid(kf, na)), oracle.featvec_from_nodes(nid(fq, nb))
    va = valid[kf, :na] if use_valid else np.ones(na, np.uint8)
    return oracle.search_by_bow(desc[kf, :na], kps[kf, :na]["angle"], va, fva, desc[fq, :nb], kps[fq, :nb]["angle"], fvb, ratio, ori)


@pytest.mark.parametrize("case", ["small_nodes", "medium_nodes", "big_nodes", "huge_node", "no_valid_array"])
def test_query_against_many_keyframes(case):
    """Every path of k_match_bow_query: nodes of <= 32 query features (one mask word), 33..64 (two), > 64 (the general
    path that reads the flags back from the result row); keyframe counts that do not fill the last group of 64; keyframes
    with 0 features; invalid keyframe / query indices (nmatches = -1, row all -1); ratio / orientation variants.  Every
    pair equals the oracle AND the pair kernel."""
    rng = np.random.default_rng({"small_nodes": 1, "medium_nodes": 2, "big_nodes": 3, "huge_node": 4, "no_valid_array": 5}[case])
    n_nodes = 100
    w = np.ones(n_nodes)
    cap, n_kf, n_q = 600, 150, 3
    if case == "medium_nodes":
        w[:6] = 6.0                                               # a few nodes of 17..40 query features
    elif case == "big_nodes":
        w[:3] = 18.0                                              # nodes of 33..64
        cap = 700
    elif case == "huge_node":
        w[0] = 60.0                                               # one node of > 64 (and > 128) query features
        w[1] = 14.0
        n_kf = 70
    ratio, ori = {"small_nodes": (0.7, True), "medium_nodes": (0.9, True), "big_nodes": (0.75, False), "huge_node": (0.8, True),
                  "no_valid_array": (0.7, True)}[case]
    desc, kps, valid, node, counts = _store(rng, n_kf, n_q, cap, n_nodes, w, reuse=float(rng.choice([0.1, 0.4])), maxflip=12)
    use_valid = case != "no_valid_array"
    F = n_kf + n_q
    kf_list = list(range(n_kf)) + [F + 5, -1, 2]                   # two invalid keyframe indices, one repeated keyframe
    q_list = [n_kf, n_kf + 1, n_kf + 2, F, -3]                      # two invalid queries
    mt = capi.Matcher(ratio, ori)
    m, n, m2, n2 = _run(mt, desc, kps, valid, node, counts, n_nodes, kf_list, q_list, use_valid)
    assert np.array_equal(n, n2), "query form vs pair kernel: counts"
    total = 0
    sizes = set()
    for qi, fq in enumerate(q_list):
        for ki, kf in enumerate(kf_list):
            if not (0 <= fq < F and 0 <= kf < F):
                assert n[qi, ki] == -1 and np.all(m[qi, ki] == -1), (qi, ki)
                continue
            nb = int(counts[fq])
            wn, wm = _oracle_pair(desc, kps, valid, node, counts, kf, fq, ratio, ori, use_valid)
            assert n[qi, ki] == wn and np.array_equal(m[qi, ki, :nb], wm), (case, qi, ki)
            assert np.all(m[qi, ki, nb:] == -1)
            assert np.array_equal(m2[qi, ki, :nb], wm)
            total += wn
        if 0 <= fq < F:
            nd = node[fq, :int(counts[fq])]
            sizes.update(np.bincount(nd[nd != 0xFFFF].astype(np.int64), minlength=n_nodes).tolist())
    assert total > 2000
    if case == "medium_nodes":
        assert any(17 <= s <= 32 for s in sizes)
    if case == "big_nodes":
        assert any(33 <= s <= 64 for s in sizes)
    if case == "huge_node":
        assert any(s > 128 for s in sizes)


def test_query_form_needs_the_sorted_descriptors():
    import torch
    mt = capi.Matcher(0.7, True)
    dev = torch.device("cuda", 0)
    z = torch.zeros(4096, dtype=torch.int32, device=dev)
    store = dict(desc=z.data_ptr(), kps=z.data_ptr(), valid=0, counts=z.data_ptr(), node_of=z.data_ptr(), cap=8, n_frames=2,
                 n_nodes=4, csr_keys=z.data_ptr(), csr_start=z.data_ptr(), csr_cnt=z.data_ptr())
    with pytest.raises(capi.OrbError):
        mt.match_bow_query_device(store, z.data_ptr(), 1, z.data_ptr(), 1, z.data_ptr(), z.data_ptr())
    store["csr_desc"] = z.data_ptr()
    mt.match_bow_query_device(store, z.data_ptr(), 0, z.data_ptr(), 1, z.data_ptr(), z.data_ptr())    # nothing to do
    mt.sync()


def test_one_matcher_reused_with_alternating_query_counts():
    """the ring of group counters.  A launch used to clear only its OWN n_queries entries of the slot a later launch
    takes, so a wide launch (10 queries), seven narrow ones (1 query) and a wide one again found entries 1..9 of its slot as
    the first wide launch left them and skipped keyframe groups (rows never written, no error).  One matcher, 19 calls,
    n_kf well above the workgroups per query; every row of every call equals the pair kernel's, and the wide calls the oracle."""
    import torch
    rng = np.random.default_rng(77)
    n_nodes, cap, n_kf, n_q = 100, 300, 260, 10
    desc, kps, valid, node, counts = _store(rng, n_kf, n_q, cap, n_nodes, np.ones(n_nodes), reuse=0.3, maxflip=10)
    F = n_kf + n_q
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    d_desc, d_kps, d_valid, d_node, d_counts = t(desc), t(kps.view(np.uint8)), t(valid), t(node.view(np.int16)), t(counts)
    d_ck = torch.zeros(F * cap, dtype=torch.int32, device=dev)
    d_cs = torch.zeros(F * n_nodes, dtype=torch.int16, device=dev)
    d_cc = torch.zeros(F * n_nodes, dtype=torch.int16, device=dev)
    d_cd = torch.zeros(F * cap * 32, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    mt = capi.Matcher(0.75, True)
    mt.build_csr_desc_device(d_node.data_ptr(), d_counts.data_ptr(), d_desc.data_ptr(), F, cap, n_nodes, d_ck.data_ptr(),
                             d_cs.data_ptr(), d_cc.data_ptr(), d_cd.data_ptr())
    store = dict(desc=d_desc.data_ptr(), kps=d_kps.data_ptr(), valid=d_valid.data_ptr(), counts=d_counts.data_ptr(),
                 node_of=d_node.data_ptr(), cap=cap, n_frames=F, n_nodes=n_nodes, csr_keys=d_ck.data_ptr(),
                 csr_start=d_cs.data_ptr(), csr_cnt=d_cc.data_ptr(), csr_desc=d_cd.data_ptr())
    kf_list = np.arange(n_kf, dtype=np.int32)
    d_kf = t(kf_list)
    # the reference rows, once, from the pair kernel of a SECOND matcher (its own counters)
    ref = capi.Matcher(0.75, True)
    d_kf2, d_f2 = t(np.tile(kf_list, n_q)), t(np.repeat(np.arange(n_kf, F, dtype=np.int32), n_kf))
    d_m2 = torch.zeros(n_q * n_kf * cap, dtype=torch.int32, device=dev)
    d_n2 = torch.zeros(n_q * n_kf, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    ref.match_bow_batch_device(store, d_kf2.data_ptr(), d_f2.data_ptr(), n_q * n_kf, d_m2.data_ptr(), d_n2.data_ptr())
    ref.sync()
    want_m, want_n = d_m2.cpu().numpy().reshape(n_q, n_kf, cap), d_n2.cpu().numpy().reshape(n_q, n_kf)
    for qi in (0, 9):                                               # ... which the oracle confirms on a sample
        for kf in (0, 57, n_kf - 1):
            wn, wm = _oracle_pair(desc, kps, valid, node, counts, kf, n_kf + qi, 0.75, True)
            assert want_n[qi, kf] == wn and np.array_equal(want_m[qi, kf, :int(counts[n_kf + qi])], wm)
    widths = [10, 1, 1, 1, 1, 1, 1, 1, 10, 1, 10, 3, 1, 1, 1, 1, 1, 1, 10]
    for call, nq in enumerate(widths):
        q_list = np.arange(n_kf, n_kf + nq, dtype=np.int32)
        d_f = t(q_list)
        d_m = torch.full((nq * n_kf * cap,), 7, dtype=torch.int32, device=dev)
        d_n = torch.full((nq * n_kf,), -9, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        mt.match_bow_query_device(store, d_kf.data_ptr(), n_kf, d_f.data_ptr(), nq, d_m.data_ptr(), d_n.data_ptr())
        mt.sync()
        got_m, got_n = d_m.cpu().numpy().reshape(nq, n_kf, cap), d_n.cpu().numpy().reshape(nq, n_kf)
        assert np.array_equal(got_n, want_n[:nq]), (call, nq, np.argwhere(got_n != want_n[:nq])[:5])
        for qi in range(nq):                                        # (the pair kernel writes a row up to the query's feature count)
            nb = int(counts[n_kf + qi])
            assert np.array_equal(got_m[qi, :, :nb], want_m[qi, :, :nb]), (call, nq, qi)
            assert np.all(got_m[qi, :, nb:] == -1), (call, nq, qi)
    assert want_n.sum() > 5000
