"""An INDEPENDENT restatement of SURVEY.md Appendix A in numpy / plain Python, written from the appendix's text (A.2 resize,
A.4 FAST-9/16 score + cell loop + cell-local NMS + threshold fallback, A.6 quadtree, A.7 blur), not from oracle/*.cpp.

Purpose: the C++ oracle is the only definition every GPU parity test is measured against, and nothing
of the reference pins it ("parity unpinned": no OpenCV here).  A second, structurally different implementation of the same
specification -- whole-array operations instead of per-pixel loops, sorted ranges instead of std::list surgery -- that
agrees with it bit for bit on fuzzed inputs (tests/test_oracle_spec_fuzz.py) catches transcription slips in either.  It does
NOT make the parity pinned: both implement the same written specification of OpenCV's behaviour.

Test infrastructure only."""
import numpy as np

RING = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1),
        (-2, 2), (-1, 3)]                       # (dx, dy), k = 0..15 (A.4)


# ------------------------------------------------------------------------------------------------ A.2 cv::resize INTER_LINEAR
def _axis(src_len, dst_len, clamp_zeroes_fraction):
    """(index, c0, c1) per output coordinate: fx = (float)((d + 0.5) * scale - 0.5) with scale = 1.0 / ((double)dst / src)."""
    scale = 1.0 / (np.float64(dst_len) / np.float64(src_len))
    d = np.arange(dst_len, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if clamp_zeroes_fraction:                   # x: the fraction is zeroed when the index is clamped
        lo, hi = s < 0, s >= src_len - 1
        f = np.where(lo | hi, np.float32(0), f)
        s = np.where(lo, 0, np.where(hi, src_len - 1, s))
    c0 = np.rint((np.float32(1.0) - f) * np.float32(2048.0)).astype(np.int64)
    c1 = np.rint(f * np.float32(2048.0)).astype(np.int64)
    c0 = np.clip(c0, -32768, 32767)
    c1 = np.clip(c1, -32768, 32767)
    return s, c0, c1


def resize(src, dw, dh):
    src = np.asarray(src, np.uint8)
    sh, sw = src.shape
    sx, a0, a1 = _axis(sw, dw, True)
    sy, b0, b1 = _axis(sh, dh, False)
    S = src.astype(np.int64)
    sx1 = np.minimum(sx + 1, sw - 1)
    H = S[:, sx] * a0[None, :] + S[:, sx1] * a1[None, :]          # horizontal pass of EVERY source row, int
    y0 = np.clip(sy, 0, sh - 1)
    y1 = np.clip(sy + 1, 0, sh - 1)
    t0 = (b0[:, None] * (H[y0] >> 4)) >> 16
    t1 = (b1[:, None] * (H[y1] >> 4)) >> 16
    return ((t0 + t1 + 2) >> 2).astype(np.uint8)


# ------------------------------------------------------------------------------------------------ A.7 GaussianBlur 7x7, sigma 2
TAPS = np.array([18, 34, 49, 55, 49, 34, 18], np.int64)


def blur(src, taps4=None):
    """taps4 = (k0, k1, k2, k3) of another OpenCV version's integer kernel (A.7's closing remark); default: TAPS"""
    T = TAPS if taps4 is None else np.array(list(taps4) + list(taps4[2::-1]), np.int64)
    S = np.pad(np.asarray(src, np.uint8).astype(np.int64), 3, mode="reflect")     # numpy 'reflect' == BORDER_REFLECT_101
    h, w = src.shape
    R = sum(T[i] * S[:, i:i + w] for i in range(7))                                # rows (padded rows included)
    D = sum(T[j] * R[j:j + h, :] for j in range(7))
    return np.clip((D + 32768) >> 16, 0, 255).astype(np.uint8)


# ------------------------------------------------------------------------------------------------ A.4 FAST-9/16 score V
def fast_v(img):
    """V(p) for the interior [3, w-3) x [3, h-3) of `img` (int array of that interior's shape): max over the 16 arcs of 9
    consecutive ring pixels of min(d) and of min(-d), d_k = I(p) - I(ring_k)."""
    I = np.asarray(img).astype(np.int16)
    h, w = I.shape
    c = I[3:h - 3, 3:w - 3]
    d = np.stack([c - I[3 + dy:h - 3 + dy, 3 + dx:w - 3 + dx] for dx, dy in RING])          # (16, h-6, w-6)
    dd = np.concatenate([d, d[:8]])                                                          # cyclic
    best = np.full(c.shape, -32768, np.int16)
    for s in range(16):
        arc = dd[s:s + 9]
        best = np.maximum(best, np.maximum(arc.min(axis=0), (-arc).min(axis=0)))
    return best


def _fast_cell(roi, th):
    """cv::FAST(roi, th, nonmax=true): (x, y, response) in ROI coordinates, ascending y then x."""
    h, w = roi.shape
    if h < 7 or w < 7:
        return np.zeros((0, 3), np.int32)
    V = fast_v(roi).astype(np.int32)
    score = np.where(V > th, V - 1, 0)                          # response of a corner; 0 elsewhere
    P = np.pad(score, 1)                                        # outside the scored interior counts as 0
    nb = np.max(np.stack([P[1 + dy:1 + dy + score.shape[0], 1 + dx:1 + dx + score.shape[1]]
                          for dy in (-1, 0, 1) for dx in (-1, 0, 1) if (dx, dy) != (0, 0)]), axis=0)
    keep = (V > th) & (score > nb)
    ys, xs = np.nonzero(keep)                                   # row-major: ascending y, then x
    return np.stack([xs + 3, ys + 3, score[ys, xs]], axis=1).astype(np.int32)


def cell_candidates(level_img, ini_th, min_th):
    """The cell loop of ComputeKeyPointsOctTree over one pyramid level (A.4): (x, y, response) relative to (16, 16)."""
    I = np.asarray(level_img, np.uint8)
    rows, cols = I.shape
    min_b, max_bx, max_by = 16, cols - 16, rows - 16
    width, height = np.float32(max_bx - min_b), np.float32(max_by - min_b)
    n_cols, n_rows = int(width / np.float32(30)), int(height / np.float32(30))
    if n_cols <= 0 or n_rows <= 0:
        return np.zeros((0, 3), np.int32)
    w_cell, h_cell = int(np.ceil(width / n_cols)), int(np.ceil(height / n_rows))
    out = []
    for i in range(n_rows):
        ini_y = min_b + i * h_cell
        max_y = ini_y + h_cell + 6
        if ini_y >= max_by - 3:
            continue
        max_y = min(max_y, max_by)
        for j in range(n_cols):
            ini_x = min_b + j * w_cell
            max_x = ini_x + w_cell + 6
            if ini_x >= max_bx - 6:
                continue
            max_x = min(max_x, max_bx)
            roi = I[ini_y:max_y, ini_x:max_x]
            k = _fast_cell(roi, ini_th)
            if len(k) == 0:
                k = _fast_cell(roi, min_th)
            if len(k):
                k = k.copy()
                k[:, 0] += j * w_cell
                k[:, 1] += i * h_cell
                out.append(k)
    return np.concatenate(out) if out else np.zeros((0, 3), np.int32)


# ------------------------------------------------------------------------------------------------ A.6 quadtree
def distribute(cands, min_x, max_x, min_y, max_y, N):
    """DistributeOctTree on (x, y, response) candidates (coordinates relative to (min_x, min_y), as the cell loop emits them).
    A node is (ULx, ULy, BRx, BRy, [indices into cands, in input order], creation number); the node list is a Python list
    whose front is index 0.  Ties of the 'largest first' phase: (size, creation number), taken from the back."""
    c = np.asarray(cands, np.int64).reshape(-1, 3)
    if max_y - min_y <= 0 or max_x - min_x <= 0:
        return np.zeros((0, 3), np.int32)
    n_ini = int(np.floor(np.float64(np.float32(max_x - min_x) / np.float32(max_y - min_y)) + 0.5))   # C round() of a positive float
    if n_ini <= 0:
        return np.zeros((0, 3), np.int32)
    hx = np.float32(max_x - min_x) / np.float32(n_ini)
    seq = [0]

    def node(ulx, uly, brx, bry, idx):
        seq[0] += 1
        return [ulx, uly, brx, bry, idx, seq[0]]

    roots = [node(int(hx * np.float32(i)), 0, int(hx * np.float32(i + 1)), max_y - min_y, []) for i in range(n_ini)]
    for k in range(len(c)):
        roots[int(np.float32(c[k, 0]) / hx)][4].append(k)
    nodes = [r for r in roots if r[4]]

    def divide(n):
        ulx, uly, brx, bry, idx, _ = n
        half_x = int(np.ceil(np.float32(brx - ulx) / np.float32(2)))
        half_y = int(np.ceil(np.float32(bry - uly) / np.float32(2)))
        mx, my = ulx + half_x, uly + half_y
        kids = [node(ulx, uly, mx, my, []), node(mx, uly, brx, my, []), node(ulx, my, mx, bry, []), node(mx, my, brx, bry, [])]
        for k in idx:
            x, y = c[k, 0], c[k, 1]
            kids[(0 if x < mx else 1) + (0 if y < my else 2)][4].append(k)
        return [q for q in kids if q[4]]

    finish = False
    while not finish:
        prev = len(nodes)
        expand = 0
        new_multi = []
        work = list(nodes)                                       # walk front -> back over the list as it is NOW
        for n in work:
            if len(n[4]) == 1:
                continue
            kids = divide(n)
            for q in kids:                                       # push_front in order n1..n4
                nodes.insert(0, q)
                if len(q[4]) > 1:
                    expand += 1
                    new_multi.append(q)
            nodes.remove(n)
        if len(nodes) >= N or len(nodes) == prev:
            finish = True
        elif len(nodes) + expand * 3 > N:
            while not finish:
                prev = len(nodes)
                todo = sorted(new_multi, key=lambda q: (len(q[4]), q[5]))
                new_multi = []
                for n in reversed(todo):                         # largest first
                    kids = divide(n)
                    for q in kids:
                        nodes.insert(0, q)
                        if len(q[4]) > 1:
                            new_multi.append(q)
                    nodes.remove(n)
                    if len(nodes) >= N:
                        break
                if len(nodes) >= N or len(nodes) == prev:
                    finish = True
    out = []
    for n in nodes:                                              # front -> back; first maximum of the response wins
        best = n[4][0]
        for k in n[4][1:]:
            if c[k, 2] > c[best, 2]:
                best = k
        out.append(c[best])
    return np.asarray(out, np.int32).reshape(-1, 3)
