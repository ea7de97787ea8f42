"""The arithmetic of csrc/bbme_mfma.hip restated lane by lane in NumPy (no GPU): window and anchor made int8 by flipping the top
bit, the A operand = 16 bytes of one window row per lane, the B operand = the anchor row shifted by the lane's column offset
(Toeplitz), v_mfma_i32_16x16x64_i8 as a 16 x 64 by 64 x 16 product with C at col = lane & 15, row = 4 (lane >> 4) + register,
costs from the signed-square box table, keys (cost << 7 | local index) in ascending scan order, wave minimum.  Must pick the
C oracle's vectors (bbme.py:105-179, pnorm 1) -- at frame edges, under exact ties, for two window sizes.  What this cannot
check is the instruction's real lane layout; tests/test_gpu_mfma.py does that on the device with exact integer data."""
import numpy as np

from helpers import c_oracle


def _i8(x):
    return ((x.astype(np.int64) ^ 0x80) + 128) % 256 - 128          # byte ^ 0x80 read as int8 = byte - 128


def _mfma(A, B, C):
    """A[lane] = 16 bytes of row m = lane & 15, k group lane >> 4; B[lane] = 16 bytes of column n = lane & 15, same k group."""
    Am = np.zeros((16, 64), np.int64)
    Bm = np.zeros((64, 16), np.int64)
    for lane in range(64):
        Am[lane & 15, 16 * (lane >> 4):16 * (lane >> 4) + 16] = A[lane]
        Bm[16 * (lane >> 4):16 * (lane >> 4) + 16, lane & 15] = B[lane]
    D = Am @ Bm
    out = C.copy()
    for lane in range(64):
        for i in range(4):
            out[lane, i] += D[4 * (lane >> 4) + i, lane & 15]
    return out


def _block(prev, cur, r0, c0, sw):
    H, W = prev.shape
    NT = (2 * sw + 16) // 16
    NC = 16 * NT
    win = np.zeros((NC + 15, NC + 16), np.int64)
    for y in range(NC + 15):
        for x in range(NC + 16):
            gy, gx = r0 - sw + y, c0 - sw + x
            win[y, x] = _i8(np.array(cur[gy, gx] if 0 <= gy < H and 0 <= gx < W else 0))
    a = _i8(prev[r0:r0 + 16, c0:c0 + 16])
    image = np.zeros(16 * 48 + 32, np.int64)                         # 16 zero bytes | anchor row | 16 zero bytes
    for r in range(16):
        image[r * 48 + 16:r * 48 + 32] = a[r]
    a2 = int((a * a).sum())
    n, g = np.arange(64) & 15, np.arange(64) >> 4
    bop = np.zeros((2, 4, 64, 16), np.int64)
    for kc in range(2):
        for rg in range(4):
            for lane in range(64):
                off = (4 * rg + g[lane]) * 48 + 16 + 16 * kc - n[lane]
                bop[kc, rg, lane] = image[off:off + 16]
    acc = np.zeros((NT, NT, 64, 4), np.int64)
    for rg in range(4):                                              # the kernel's order: anchor row group outer
        for ty in range(NT):
            for xi in range(NT + 1):
                wop = np.stack([win[4 * (4 * ty + rg) + n[lane] + g[lane], 16 * xi:16 * xi + 16] for lane in range(64)])
                if xi < NT:
                    acc[ty, xi] = _mfma(wop, bop[0, rg], acc[ty, xi])
                if xi >= 1:
                    acc[ty, xi - 1] = _mfma(wop, bop[1, rg], acc[ty, xi - 1])
    lo_r, hi_r = max(0, sw - r0), min(NC - 1, H - 16 - r0 + sw)
    lo_c, hi_c = max(0, sw - c0), min(NC - 1, W - 16 - c0 + sw)

    def table(y, x):                                                 # k_sqbox16<true>: sum (b - 128)^2 over the 16x16 box
        b = cur[y:y + 16, x:x + 16].astype(np.int64) - 128
        return int((b * b).sum())
    keys = np.full(64, 0x7FFFFFFF, np.int64)
    for tx in range(NT):
        for ty in range(NT):
            for i in range(4):
                for lane in range(64):
                    ri, ci = 16 * ty + 4 * g[lane] + i, 16 * tx + n[lane]
                    if lo_c <= ci <= hi_c and lo_r <= ri <= hi_r:
                        cost = table(r0 - sw + ri, c0 - sw + ci) - 2 * int(acc[ty, tx, lane, i])      # + a2 at the end
                        assert -2 ** 24 < cost < 2 ** 24
                        keys[lane] = min(keys[lane], cost * 128 + (tx * NT + ty) * 4 + i)
    best_cost = min((k >> 7) + a2 for k in keys if k != 0x7FFFFFFF)
    idx = []
    for lane in range(64):
        k = int(keys[lane])
        if k != 0x7FFFFFFF and (k >> 7) + a2 == best_cost:
            local = k & 127
            i, t = local & 3, local >> 2
            tx, ty = t // NT, t % NT
            idx.append((16 * tx + n[lane]) * NC + 16 * ty + 4 * g[lane] + i)
    ci, ri = divmod(min(idx), NC)
    return ci - sw, ri - sw


def test_mfma_formulation_picks_the_oracles_vectors():
    co = c_oracle()
    rng = np.random.default_rng(11)
    cases = []
    cur = rng.integers(0, 256, (48, 64), dtype=np.uint8)
    prev = np.roll(cur, (3, -5), (0, 1)).copy()
    prev[::7] = rng.integers(0, 256, prev[::7].shape, dtype=np.uint8)
    cases.append((prev, cur, 16, [(0, 0), (16, 48), (32, 16)]))                       # corners, edges, interior-ish
    cases.append((np.full((40, 48), 7, np.uint8), np.full((40, 48), 7, np.uint8), 8, [(0, 0), (16, 16), (16, 32)]))   # every cost ties
    ext = (rng.integers(0, 2, (2, 33, 47), dtype=np.uint8) * 255).astype(np.uint8)    # 0 / 255: the largest products
    cases.append((ext[0], ext[1], 8, [(0, 0), (16, 16)]))
    for prev, cur, sw, blocks in cases:
        want = co.bbme(prev, cur, 16, sw, 0, 1)
        for r0, c0 in blocks:
            got = _block(prev, cur, r0, c0, sw)
            assert tuple(want[r0 // 16, c0 // 16]) == got, (prev.shape, sw, r0, c0, got, want[r0 // 16, c0 // 16])
