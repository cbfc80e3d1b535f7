"""Seeded synthetic inputs for the ORB front-end.

No dataset can be fetched, so every benchmark / parity input is procedurally generated from a
counter-based splitmix64 stream: frame i uses seed 0x0B5EED00 + i.  The generator is plain
numpy (it is input plumbing, not part of the oracle and not part of the measured path).
"""
import numpy as np

SEED0 = 0x0B5EED00
_G = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def splitmix64(seed: int, start: int, n: int) -> np.ndarray:
    """Outputs number start..start+n-1 (0-based) of the splitmix64 stream seeded with `seed`."""
    with np.errstate(over="ignore"):
        idx = np.arange(start + 1, start + n + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + idx * _G
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def synth_frame(index: int, width: int = 640, height: int = 480,
                n_rect: int = 400, n_disc: int = 200, noise: int = 6) -> np.ndarray:
    """Mid-grey canvas, n_rect rectangles then n_disc discs (size 4..120 px, grey 0..255, later
    shapes overwrite earlier ones), then uniform per-pixel noise in [-noise, noise]; u8."""
    seed = SEED0 + index
    nshape = 5 * n_rect + 4 * n_disc
    r = splitmix64(seed, 0, nshape)
    img = np.full((height, width), 128, dtype=np.int16)
    p = 0
    for _ in range(n_rect):
        x0 = int(r[p] % np.uint64(width)); y0 = int(r[p + 1] % np.uint64(height))
        w = 4 + int(r[p + 2] % np.uint64(117)); h = 4 + int(r[p + 3] % np.uint64(117))
        g = int(r[p + 4] % np.uint64(256)); p += 5
        img[y0:min(height, y0 + h), x0:min(width, x0 + w)] = g
    for _ in range(n_disc):
        cx = int(r[p] % np.uint64(width)); cy = int(r[p + 1] % np.uint64(height))
        rad = 2 + int(r[p + 2] % np.uint64(59)); g = int(r[p + 3] % np.uint64(256)); p += 4
        xa, xb = max(0, cx - rad), min(width, cx + rad + 1)
        ya, yb = max(0, cy - rad), min(height, cy + rad + 1)
        yy, xx = np.mgrid[ya:yb, xa:xb]
        m = (xx - cx) ** 2 + (yy - cy) ** 2 <= rad * rad
        img[ya:yb, xa:xb][m] = g
    if noise > 0:
        nz = splitmix64(seed, nshape, width * height) % np.uint64(2 * noise + 1)
        img += nz.astype(np.int16).reshape(height, width) - noise
    return np.clip(img, 0, 255).astype(np.uint8)


def _value_noise(seed: int, stream: int, width: int, height: int, cell: int) -> np.ndarray:
    """Bilinear interpolation (integer, 8 fractional bits of weight) of a seeded random lattice with `cell`-pixel
    spacing: values 0..255 << 8."""
    gw, gh = width // cell + 2, height // cell + 2
    lat = (splitmix64(seed, stream, gw * gh) % np.uint64(256)).astype(np.int64).reshape(gh, gw)
    ys, xs = np.arange(height, dtype=np.int64), np.arange(width, dtype=np.int64)
    y0, x0 = ys // cell, xs // cell
    fy = ((ys - y0 * cell) * 256) // cell
    fx = ((xs - x0 * cell) * 256) // cell
    a = lat[y0][:, x0]; b = lat[y0][:, x0 + 1]; c = lat[y0 + 1][:, x0]; d = lat[y0 + 1][:, x0 + 1]
    top = a * (256 - fx) + b * fx
    bot = c * (256 - fx) + d * fx
    return (top * (256 - fy)[:, None] + bot * fy[:, None]) >> 8          # 0 .. 255 << 8


def synth_natural(index: int, width: int = 640, height: int = 480) -> np.ndarray:
    """Stand-in for camera frames (TUM / KITTI / EuRoC are absent): image statistics instead of drawn shapes.  Integer
    arithmetic only, so the bytes do not depend on the FFT / libm of the machine:
      * 1/f-like texture: value-noise octaves of cell size 2..128 px whose amplitude grows with the cell size
        (amplitude ~ cell^0.75: a power spectrum close to 1/f^2.5, between clouds and man-made scenes);
      * 60 occluding objects (rectangles and ellipses) that shift the local brightness by -50..50 and carry the texture
        on: step edges and corners of moderate contrast, T-junctions where they overlap;
      * two passes of a 3x3 binomial blur (lens / demosaicing), a linear illumination ramp of up to +-30 grey levels
        across the frame, uniform sensor noise in [-2, 2]; clamped to u8."""
    seed = SEED0 + 0x4E415400 + index                                 # 'NAT'
    acc = np.zeros((height, width), np.int64)
    stream = 0
    wsum = 0
    for o, cell in enumerate((2, 4, 8, 16, 32, 64, 128)):
        wgt = (3, 5, 8, 14, 23, 39, 66)[o]                            # ~ cell^0.75
        acc += wgt * (_value_noise(seed, stream, width, height, cell) - (128 << 8))
        stream += (width // cell + 2) * (height // cell + 2)
        wsum += wgt
    img = 120 + (acc * 7) // (4 * (wsum << 8))                        # texture contrast: a standard deviation of about 40 grey levels
    r = splitmix64(seed, 1 << 24, 60 * 6)
    p = 0
    yy, xx = np.mgrid[0:height, 0:width]
    for k in range(60):
        cx = int(r[p] % np.uint64(width)); cy = int(r[p + 1] % np.uint64(height))
        hw = 6 + int(r[p + 2] % np.uint64(90)); hh = 6 + int(r[p + 3] % np.uint64(90))
        dv = int(r[p + 4] % np.uint64(101)) - 50; kind = int(r[p + 5] % np.uint64(2)); p += 6
        ya, yb, xa, xb = max(0, cy - hh), min(height, cy + hh + 1), max(0, cx - hw), min(width, cx + hw + 1)
        if kind == 0:
            img[ya:yb, xa:xb] += dv
        else:
            m = ((xx[ya:yb, xa:xb] - cx) * hh) ** 2 + ((yy[ya:yb, xa:xb] - cy) * hw) ** 2 <= (hw * hh) ** 2
            img[ya:yb, xa:xb][m] += dv
    for _ in range(2):                                                 # 3x3 binomial, edges replicated
        q = np.pad(img, 1, mode="edge")
        h = q[:, :-2] + 2 * q[:, 1:-1] + q[:, 2:]
        img = (h[:-2] + 2 * h[1:-1] + h[2:] + 8) >> 4
    g = splitmix64(seed, 1 << 25, 2)
    gx = int(g[0] % np.uint64(61)) - 30; gy = int(g[1] % np.uint64(61)) - 30
    img = img + (gx * (2 * xx - width)) // (2 * width) + (gy * (2 * yy - height)) // (2 * height)
    nz = splitmix64(seed, 1 << 26, width * height) % np.uint64(5)
    img = img + nz.astype(np.int64).reshape(height, width) - 2
    return np.clip(img, 0, 255).astype(np.uint8)


def synth_natural_batch(first: int, count: int, width: int = 640, height: int = 480) -> np.ndarray:
    return np.stack([synth_natural(first + i, width, height) for i in range(count)])


def synth_natural_stereo_right(left_index: int, width: int = 1241, height: int = 376) -> np.ndarray:
    """Right view of a natural-statistics pair: the left frame shifted left by d(y) = 12 + 8 * floor(y / 94) px with fresh
    sensor noise (the same disparity layout as synth_stereo_right)."""
    base = synth_natural(left_index, width, height).astype(np.int16)
    out = np.empty_like(base)
    for y in range(height):
        d = 12 + 8 * (y // 94)
        out[y, : width - d] = base[y, d:]
        out[y, width - d:] = base[y, width - 1]
    nz = splitmix64(SEED0 + 0x4E415400 + left_index + 0x10000, 0, width * height) % np.uint64(5)
    out += nz.astype(np.int16).reshape(height, width) - 2
    return np.clip(out, 0, 255).astype(np.uint8)


def synth_batch(first: int, count: int, width: int = 640, height: int = 480) -> np.ndarray:
    return np.stack([synth_frame(first + i, width, height) for i in range(count)])


def synth_stereo_right(left_index: int, width: int = 1241, height: int = 376, noise: int = 6) -> np.ndarray:
    """Right image of the C3 pair: the (noise-free) left frame shifted left by the piecewise-constant
    disparity d(y) = 12 + 8*floor(y/94) px, with fresh noise."""
    base = synth_frame(left_index, width, height, noise=0).astype(np.int16)
    out = np.empty_like(base)
    for y in range(height):
        d = 12 + 8 * (y // 94)
        out[y, : width - d] = base[y, d:]
        out[y, width - d:] = base[y, width - 1]
    nz = splitmix64(SEED0 + left_index + 0x10000, 0, width * height) % np.uint64(2 * noise + 1)
    out += nz.astype(np.int16).reshape(height, width) - noise
    return np.clip(out, 0, 255).astype(np.uint8)


def synth_vocabulary(seed: int = 0x0B0C0DE) -> np.ndarray:
    """Stand-in for the absent ORB vocabulary: 10 level-1 + 100 level-2 random 256-bit centroids,
    (110, 32) u8.  Node id of a descriptor = 11 + 10*c1 + c2 (first-minimum Hamming descent)."""
    r = splitmix64(seed, 0, 110 * 4)
    return r.view(np.uint8).reshape(110, 32).copy()


def synth_valid_flags(n: int, seed: int, p_num: int = 7, p_den: int = 10) -> np.ndarray:
    """Bernoulli(p) 'has a good MapPoint' flags for a keyframe's features."""
    r = splitmix64(0x7A11D000 + seed, 0, n) % np.uint64(p_den)
    return (r < np.uint64(p_num)).astype(np.uint8)


def synth_vocab_tree(k: int = 10, L: int = 3, seed: int = 0xB0CAB, prune: float = 0.1, shuffle_ids: bool = True) -> dict:
    """A seeded stand-in for a DBoW2 vocabulary (the ORBvoc file is absent): branching factor k, depth L, random
    256-bit node descriptors; a fraction `prune` of inner nodes is cut into early leaves (trained vocabularies are
    unbalanced) and node ids are permuted so children are NOT consecutive (ids come from file order in DBoW2).
    Returns flat arrays: node_desc (n,32) u8, child_begin (n+1) i32, children i32, word_id (n) i32 (-1 inner), L."""
    rng = np.random.default_rng(seed)
    children_of = [[]]
    depth = [0]
    frontier = [0]
    for lvl in range(L):
        nxt = []
        for v in frontier:
            if v != 0 and rng.random() < prune:
                continue                                            # early leaf
            for _ in range(k):
                children_of.append([])
                depth.append(lvl + 1)
                children_of[v].append(len(children_of) - 1)
                nxt.append(len(children_of) - 1)
        frontier = nxt
    n = len(children_of)
    perm = np.arange(n)
    if shuffle_ids:
        perm[1:] = 1 + rng.permutation(n - 1)                       # root stays node 0
    new_children = [None] * n
    for v in range(n):
        new_children[perm[v]] = [int(perm[c]) for c in children_of[v]]
    child_begin = np.zeros(n + 1, np.int32)
    for v in range(n):
        child_begin[v + 1] = child_begin[v] + len(new_children[v])
    children = np.array([c for v in range(n) for c in new_children[v]], np.int32)
    word_id = np.full(n, -1, np.int32)
    leaves = [v for v in range(n) if not new_children[v]]
    word_id[leaves] = np.arange(len(leaves), dtype=np.int32)
    node_desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    return dict(node_desc=node_desc, child_begin=child_begin, children=children, word_id=word_id, L=L, k=k)


def synth_sequence(first: int, count: int, width: int = 640, height: int = 480, views: int = 8, noise: int = 6,
                   content: str = "shapes") -> np.ndarray:
    """Frames that look like a slowly moving camera: frame i is view v = i % views of scene g = i // views, i.e. the
    noise-free synth_frame(views * g) rolled by (v, 2v) pixels (rows, columns) with fresh noise.  Consecutive frames of
    a scene share almost all corners, so SearchByBoW between frame i and i+1 finds hundreds of matches (unrelated
    scenes give about 7) -- the accept / greedy-taken / rotation-histogram path of the matcher is exercised.
    content = "natural": the scenes are synth_natural frames (which carry their own sensor noise) and a view adds fresh
    uniform noise in [-2, 2] only."""
    if content == "natural":
        noise = min(noise, 2)
    out = np.empty((count, height, width), np.uint8)
    base, base_g = None, -1
    for k in range(count):
        i = first + k
        g, v = divmod(i, views)
        if g != base_g:
            base = synth_natural(views * g, width, height) if content == "natural" else synth_frame(views * g, width, height, noise=0)
            base, base_g = base.astype(np.int16), g
        img = np.roll(base, (v, 2 * v), axis=(0, 1))
        if noise > 0:
            nz = splitmix64(SEED0 + 0x20000 + i, 0, width * height) % np.uint64(2 * noise + 1)
            img = img + (nz.astype(np.int16).reshape(height, width) - noise)
        out[k] = np.clip(img, 0, 255).astype(np.uint8)
    return out


def synth_vocab_tree_balanced(k: int = 10, L: int = 6, seed: int = 0xB0CAB) -> dict:
    """A complete k-ary tree of depth L in the layout of synth_vocab_tree (vectorised: 1.1 M nodes for the DBoW2 shape
    k = 10, L = 6 of ORBvoc, the work Frame::ComputeBoW's transform(..., 4) does per feature: 6 levels x 10 Hamming).
    Node ids are permuted (ids come from file order in DBoW2); leaves are the words."""
    rng = np.random.default_rng(seed)
    n_inner = (k ** L - 1) // (k - 1)
    n = n_inner + k ** L
    perm = np.arange(n, dtype=np.int64)
    perm[1:] = 1 + rng.permutation(n - 1)                           # BFS index -> node id; root stays node 0
    inv = np.empty(n, np.int64)
    inv[perm] = np.arange(n)                                        # node id -> BFS index
    nchild = np.where(inv < n_inner, k, 0)
    child_begin = np.zeros(n + 1, np.int64)
    np.cumsum(nchild, out=child_begin[1:])
    children = np.empty(int(child_begin[-1]), np.int32)
    inner_ids = np.nonzero(inv < n_inner)[0]                        # ascending node id
    bfs = inv[inner_ids]
    kid_bfs = (bfs[:, None] * k + 1 + np.arange(k)[None, :])        # BFS children of BFS node b: k*b+1 .. k*b+k
    children[:] = perm[kid_bfs].reshape(-1).astype(np.int32)
    word_id = np.full(n, -1, np.int32)
    leaves = np.nonzero(inv >= n_inner)[0]
    word_id[leaves] = np.arange(leaves.size, dtype=np.int32)
    node_desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    return dict(node_desc=node_desc, child_begin=child_begin.astype(np.int32), children=children, word_id=word_id, L=L, k=k)


_POP8 = np.array([bin(i).count("1") for i in range(256)], np.uint16)


def _hamming_to(desc: np.ndarray, cent: np.ndarray) -> np.ndarray:
    """(n, 32) u8 x (k, 32) u8 -> (n, k) Hamming distances."""
    return _POP8[desc[:, None, :] ^ cent[None, :, :]].sum(axis=2)


def _k_majority(desc: np.ndarray, k: int, rng, iters: int = 8) -> np.ndarray:
    """k-means in Hamming space with bitwise-majority centroids (what DBoW2's training does for binary descriptors);
    returns (k, 32) u8 centroids.  Empty clusters are re-seeded from the largest one."""
    n = desc.shape[0]
    cent = desc[rng.choice(n, k, replace=n < k)].copy()
    bits = np.unpackbits(desc, axis=1)
    for _ in range(iters):
        lab = _hamming_to(desc, cent).argmin(axis=1)
        for c in range(k):
            m = lab == c
            if not m.any():
                big = np.bincount(lab, minlength=k).argmax()
                cent[c] = desc[rng.choice(np.nonzero(lab == big)[0])]
                continue
            cent[c] = np.packbits(bits[m].mean(axis=0) >= 0.5)
    return cent


def synth_vocab_tree_trained(desc_sample: np.ndarray, k: int = 10, L: int = 6, seed: int = 0xB0CAB) -> dict:
    """A complete k-ary tree of depth L in the layout of synth_vocab_tree whose two top levels are TRAINED on a sample
    of descriptors by hierarchical k-majority clustering, as a DBoW2 vocabulary is (ORBvoc itself is absent): the
    level-2 nodes -- the FeatureVector nodes SearchByBoW iterates over at levelsup = 4 -- then hold a few features of
    a frame each (the regime the reference runs in, "~10 features per node").  A tree of uniformly random node
    descriptors is badly unbalanced instead: real descriptors are not uniform, one node swallows a quarter of a frame.
    Levels 3..L are seeded random refinements of their parent (a shrinking number of flipped bits), so the descent does
    the full 6 x 10 Hamming comparisons per feature.  Node ids are permuted as in synth_vocab_tree_balanced."""
    rng = np.random.default_rng(seed)
    desc_sample = np.ascontiguousarray(desc_sample, np.uint8)
    n_inner = (k ** L - 1) // (k - 1)
    n = n_inner + k ** L
    bfs_desc = np.zeros((n, 32), np.uint8)
    c1 = _k_majority(desc_sample, k, rng)
    bfs_desc[1:1 + k] = c1
    lab1 = _hamming_to(desc_sample, c1).argmin(axis=1)
    for a in range(k):
        sub = desc_sample[lab1 == a]
        if sub.shape[0] < k:
            sub = desc_sample
        b0 = k * (1 + a) + 1                                       # BFS index of the first child of level-1 node a
        bfs_desc[b0:b0 + k] = _k_majority(sub, k, rng)
    flips = [0, 0, 0, 24, 16, 10, 6, 4, 3, 2]
    level_start = [(k ** d - 1) // (k - 1) for d in range(L + 2)]
    for d in range(3, L + 1):
        lo, hi = level_start[d], level_start[d + 1]
        parents = (np.arange(lo, hi) - 1) // k
        child = bfs_desc[parents].copy()
        nb = flips[min(d, len(flips) - 1)]
        pos = rng.integers(0, 256, (hi - lo, nb))
        for j in range(nb):
            child[np.arange(hi - lo), pos[:, j] >> 3] ^= (1 << (pos[:, j] & 7)).astype(np.uint8)
        bfs_desc[lo:hi] = child
    t = synth_vocab_tree_balanced(k, L, seed)                       # the id permutation and the topology
    # synth_vocab_tree_balanced draws perm first from the same seed: recover BFS index of every node id from its children
    cb, ch = t["child_begin"], t["children"]
    inv = np.full(n, -1, np.int64)                                  # node id -> BFS index
    inv[0] = 0
    frontier = np.array([0])
    while frontier.size:
        kids_cnt = cb[frontier + 1] - cb[frontier]
        par = frontier[kids_cnt > 0]
        if par.size == 0:
            break
        kid_ids = ch[(cb[par][:, None] + np.arange(k)[None, :])]
        inv[kid_ids] = inv[par][:, None] * k + 1 + np.arange(k)[None, :]
        frontier = kid_ids.reshape(-1)
    t["node_desc"] = bfs_desc[inv]
    return t
