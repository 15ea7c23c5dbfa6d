"""Lane-level CPU emulation of csrc/ffn.h (the fused feed-forward kernel) for ONE workgroup: the packed weight stream
(interdiff_amd/mdm.py: pack_ffn), the LDS images (XOR-swizzled x2 / hid rows, swizzled [rows][16] chunks in a 6-slot ring), the
order in which chunk pairs are issued and consumed, the MFMA fragment <-> lane maps and both epilogues are restated here in numpy,
so that the host-side packing and the kernel's addressing can be checked against a plain matmul WITHOUT a GPU.  The DMA is applied
either at issue time ("early") or at the wait that covers it ("late"): a slot that is overwritten before its last read, or read
before its data is guaranteed, gives a wrong answer in one of the two.  TEST INFRASTRUCTURE (not product code)."""
import numpy as np

D, FF, BM, NSL, NW = 256, 1024, 32, 5, 8
HS, NTILE = 208, 13
W1C, W2C = HS * 16, D * 16
NP1, NP2 = 8, 7
NPAIR = NP1 + NP2
SLICE_FLOATS = 16 * W1C + NTILE * W2C
PSLOT = 2 * W2C


def _gelu(x):
    from math import erf
    return 0.5 * x * (1.0 + np.vectorize(erf)(x / np.sqrt(2.0)))


def pair_off(P):
    return P * 2 * W1C if P < NP1 else 16 * W1C + (P - NP1) * 2 * W2C


def pair_ins(P):
    if P >= NPAIR:
        return 0
    return 2 * W1C // 256 if P < NP1 else (2 * W2C // 256 if P < NPAIR - 1 else W2C // 256)


def emulate_workgroup(x2, pack, b1p, b2, mt, sl, late, bm=BM):
    """bm = 32: ffn_fused_kernel; bm = 16: ffn_fused16_kernel (one row tile, two column tiles per wave, no shared tile); bm = 64:
    ffn_fused64_kernel (four row tiles, wave w = row tile w & 3 x column half w >> 2, TWO ring slots refilled after the barrier)."""
    M = x2.shape[0]
    m0 = mt * bm
    small = bm == 16
    big = bm == 64
    nslot = 2 if big else 3
    nt = NTILE
    stream = pack[sl * SLICE_FLOATS:(sl + 1) * SLICE_FLOATS]
    Xs, ring, Bs = np.zeros(bm * D), np.full(nslot * PSLOT, np.nan), np.zeros(256)
    lane = np.arange(64)
    li, kq = lane & 15, lane >> 4
    key = (4 - (li >> 2)) & 3
    pending = {}

    def issue_pair(P):
        ops = []
        nins = pair_ins(P)
        for wave in range(NW):
            for j in range(4):
                i = wave + 8 * j
                if i < nins:                              # 8j+7 < nins for every wave, or the ragged tail for waves 0, 1
                    assert (8 * j + 7 < nins) or (8 * j < nins and wave < 2)
                    src = pair_off(P) + i * 256
                    ops.append(((P % nslot) * PSLOT + i * 256, stream[src:src + 256].copy()))
        if late and ops:
            pending[P] = ops
        else:
            for dst, data in ops:
                ring[dst:dst + 256] = data

    def land(P):
        for dst, data in pending.pop(P, []):
            ring[dst:dst + 256] = data

    # prologue
    Bs[:256] = b1p[sl * HS:sl * HS + 256]
    for i in range(bm):
        row = x2[min(m0 + i, M - 1)]
        for l in range(64):
            Xs[i * D + l * 4:i * D + l * 4 + 4] = row[(l ^ (i & 15)) * 4:(l ^ (i & 15)) * 4 + 4]
    issue_pair(0)
    issue_pair(1)
    land(0)
    # ---- phase 1
    hid_acc = {}                                      # (wave, j) -> [16 rows][16 cols] accumulator tile in D layout

    def mfma_group(acc, a, b):                        # a [64,4], b [64,4]: four 16x16x4 MFMAs
        for comp in range(4):
            A = np.zeros((16, 4))
            Bm = np.zeros((4, 16))
            A[li, kq] = a[:, comp]
            Bm[kq, li] = b[:, comp]
            acc += A @ Bm
    maps = []
    for wave in range(NW):
        r1 = (wave >> 1) & 1
        c0 = ((7 if wave < 4 else 10) if wave & 1 else (0 if wave < 4 else 4))
        nct = 4 if wave in (0, 2) else 3
        if small:                                     # 13 column tiles of the one row tile: 2,2,2,2,2,1,1,1
            r1, c0, nct = 0, (2 * wave if wave < 5 else 5 + wave), (2 if wave < 5 else 1)
        if big:                                       # row tile w & 3, column tiles 0..6 (w < 4) or 7..12
            r1, c0, nct = wave & 3, (7 if wave >= 4 else 0), (6 if wave >= 4 else 7)
        maps.append((r1, c0, nct))
        for j in range(nct if (small or big) else (4 if wave < 4 else 3)):         # waves 1, 3 hold the odd-chunk half of their neighbour's fourth tile in acc[3]
            hid_acc[(wave, j)] = np.zeros((16, 16))

    def read1(wave, c):
        r1, c0, nct = maps[wave]
        rows = r1 * 16 + li
        a = np.stack([Xs[rows * D + (((kq ^ li) ^ (4 * (c & 3))) << 2) + 64 * (c >> 2) + t] for t in range(4)], axis=1)
        bs = []
        for j in range(nct if (small or big) else 3):
            base = ((c >> 1) % nslot) * PSLOT + (c & 1) * W1C + ((kq ^ key) << 2) + li * 16 + (c0 + j) * 256
            bs.append(np.stack([ring[base + t] for t in range(4)], axis=1))
        if small or big:
            return a, bs
        # the shared column tile 3: owners (waves 0, 2) on even chunks and the last one, helpers (waves 1, 3) on the other odd chunks
        if (wave in (1, 3)) if (c & 1 and c != 15) else (wave in (0, 2)):
            base = ((c >> 1) % nslot) * PSLOT + (c & 1) * W1C + ((kq ^ key) << 2) + li * 16 + 3 * 256
            bs.append(np.stack([ring[base + t] for t in range(4)], axis=1))
        return a, bs
    frag0 = [read1(w, 0) for w in range(NW)]
    for P in range(8):
        if not big:
            issue_pair(P + 2)
        frag1 = [read1(w, 2 * P + 1) for w in range(NW)]
        for w in range(NW):
            a, bs = frag0[w]
            for j, b in enumerate(bs):
                mfma_group(hid_acc[(w, j)], a, b)
        land(P + 1)                                   # the wait + barrier: pair P+1 has landed
        if big:
            issue_pair(P + 2)                         # two slots: pair P's slot is refilled once every wave has read it
        if P + 1 < 8:
            frag0 = [read1(w, 2 * P + 2) for w in range(NW)]
        for w in range(NW):
            a, bs = frag1[w]
            for j, b in enumerate(bs):
                mfma_group(hid_acc[(w, j)], a, b)
    assert small or big or (maps[0][0] == maps[1][0] and maps[2][0] == maps[3][0])            # owner and helper work on the same row tile
    if not (small or big):
        hid_acc[(0, 3)] += hid_acc.pop((1, 3))
        hid_acc[(2, 3)] += hid_acc.pop((3, 3))
    # epilogue 1: gelu(acc + b1) -> Xs (swizzled), D layout: lane (li, kq), reg rr -> row kq*4+rr, col li
    for w in range(NW):
        r1, c0, nct = maps[w]
        for j in range(nct):
            t = hid_acc[(w, j)]
            for rr in range(16):
                for cc in range(16):
                    row, col = r1 * 16 + rr, (c0 + j) * 16 + cc
                    Xs[row * D + (((col >> 2) ^ (row & 15)) << 2) + (col & 3)] = _gelu(t[rr, cc] + Bs[col])
    # ---- phase 2
    nj2 = 2 if small else (8 if big else 4)
    out_acc = {(w, j): np.zeros((16, 16)) for w in range(NW) for j in range(nj2)}

    def read2(wave, q):
        r2, nb = (0, wave * 2) if small else ((wave & 3, (wave >> 2) * 8) if big else (wave & 1, (wave >> 1) * 4))
        rows = r2 * 16 + li
        a = np.stack([Xs[rows * D + (((kq ^ li) ^ (4 * (q & 3))) << 2) + 64 * (q >> 2) + t] for t in range(4)], axis=1)
        bs = []
        for j in range(nj2):
            base = ((NP1 + (q >> 1)) % nslot) * PSLOT + (q & 1) * W2C + ((kq ^ key) << 2) + (nb * 16 + li) * 16 + j * 256
            bs.append(np.stack([ring[base + t] for t in range(4)], axis=1))
        return a, bs
    frag0 = [read2(w, 0) for w in range(NW)]
    for P in range(NP1, NPAIR):
        q = 2 * (P - NP1)
        two = q + 1 < nt
        if not big:
            issue_pair(P + 2)
        if two:
            frag1 = [read2(w, q + 1) for w in range(NW)]
        for w in range(NW):
            a, bs = frag0[w]
            for j, b in enumerate(bs):
                mfma_group(out_acc[(w, j)], a, b)
        land(P + 1)
        if big:
            issue_pair(P + 2)
        if q + 2 < nt:
            frag0 = [read2(w, q + 2) for w in range(NW)]
        if two:
            for w in range(NW):
                a, bs = frag1[w]
                for j, b in enumerate(bs):
                    mfma_group(out_acc[(w, j)], a, b)
    assert not pending
    part = np.zeros((bm, D))
    for w in range(NW):
        r2, nb = (0, w * 2) if small else ((w & 3, (w >> 2) * 8) if big else (w & 1, (w >> 1) * 4))
        for j in range(nj2):
            part[r2 * 16:r2 * 16 + 16, (nb + j) * 16:(nb + j) * 16 + 16] = out_acc[(w, j)]
    rows = np.minimum(m0 + np.arange(bm), M - 1)
    if sl == 0:
        part = part + x2[rows] + b2[None, :]
    return part[:max(0, min(bm, M - m0))]


def emulate_ffn(x2, pack, b1p, b2, late, bm=BM):
    """All workgroups -> parts [NSL][M][256] (float64 arithmetic)."""
    M = x2.shape[0]
    parts = np.zeros((NSL, M, D))
    for mt in range((M + bm - 1) // bm):
        for sl in range(NSL):
            p = emulate_workgroup(x2.astype(np.float64), pack.astype(np.float64), b1p.astype(np.float64), b2.astype(np.float64), mt, sl, late, bm)
            parts[sl, mt * bm:mt * bm + p.shape[0]] = p
    return parts


# ---------------------------------------------------------------------------------------------------------------------------
# csrc/ffn.h ln_linear_kernel (LayerNorm + linear of the QKV projection), one workgroup
LCT, LHS = 10, 160
LW1C = LHS * 16
LPSLOT = 2 * LW1C


def emulate_ln_linear(A_slabs, lnw, lnb, pack, bias, N, late):
    """A_slabs [NP][M][256] -> C [M][N] (float64), all workgroups; the DMA applied at issue or at the covering wait."""
    M = A_slabs.shape[1]
    C = np.zeros((M, N))
    lane = np.arange(64)
    li, kq = lane & 15, lane >> 4
    key = (4 - (li >> 2)) & 3
    x = A_slabs.astype(np.float64).sum(0)
    if lnw is not None:
        mu = x.mean(1, keepdims=True)
        var = ((x - mu) ** 2).mean(1, keepdims=True)
        x = (x - mu) / np.sqrt(var + 1e-5) * lnw + lnb
    tiles = [(w & 1, (w >> 1) * 3 if w < 4 else 6 + ((w - 4) >> 1) * 2, 3 if w < 4 else 2) for w in range(NW)]      # (row tile, first column tile, count)
    assert sorted((r, c0 + j) for r, c0, n in tiles for j in range(n)) == [(r, c) for r in range(2) for c in range(LCT)]
    for mt in range((M + BM - 1) // BM):
        for sl in range((N + LHS - 1) // LHS):
            m0, n0 = mt * BM, sl * LHS
            stream = pack[sl * 16 * LW1C:(sl + 1) * 16 * LW1C].astype(np.float64)
            Xs, ring = np.zeros(BM * D), np.full(3 * LPSLOT, np.nan)
            pending = {}

            def issue_pair(P):
                if P >= 8:
                    return
                ops = []
                for wave in range(NW):
                    for j in range(3):
                        if j < 2 or wave < 4:                            # the third instruction: waves 0..3
                            off = wave * 256 + 2048 * j                  # floats: (wave*1024 + 8192 j) bytes
                            src = P * 2 * LW1C + off
                            ops.append(((P % 3) * LPSLOT + off, stream[src:src + 256].copy()))
                assert len(ops) * 256 == 2 * LW1C
                if late:
                    pending[P] = ops
                else:
                    for dst, data in ops:
                        ring[dst:dst + 256] = data

            def land(P):
                for dst, data in pending.pop(P, []):
                    ring[dst:dst + 256] = data
            issue_pair(0)
            issue_pair(1)
            for row in range(BM):
                src = x[min(m0 + row, M - 1)]
                for l in range(64):
                    p = l ^ (row & 15)
                    Xs[row * D + p * 4:row * D + p * 4 + 4] = src[l * 4:l * 4 + 4]
            land(0)
            acc = {(w, j): np.zeros((16, 16)) for w in range(NW) for j in range(tiles[w][2])}

            def rd(w, c):
                r1, c0, n = tiles[w]
                rows = r1 * 16 + li
                a = np.stack([Xs[rows * D + (((kq ^ li) ^ (4 * (c & 3))) << 2) + 64 * (c >> 2) + t] for t in range(4)], axis=1)
                bs = []
                for j in range(n):
                    base = ((c >> 1) % 3) * LPSLOT + (c & 1) * LW1C + ((kq ^ key) << 2) + li * 16 + (c0 + j) * 256
                    bs.append(np.stack([ring[base + t] for t in range(4)], axis=1))
                return a, bs

            def mma(w, frag):
                a, bs = frag
                for j, b in enumerate(bs):
                    for comp in range(4):
                        Am, Bm = np.zeros((16, 4)), np.zeros((4, 16))
                        Am[li, kq] = a[:, comp]
                        Bm[kq, li] = b[:, comp]
                        acc[(w, j)] += Am @ Bm
            f0 = [rd(w, 0) for w in range(NW)]
            for P in range(8):
                issue_pair(P + 2)
                f1 = [rd(w, 2 * P + 1) for w in range(NW)]
                for w in range(NW):
                    mma(w, f0[w])
                land(P + 1)
                if P + 1 < 8:
                    f0 = [rd(w, 2 * P + 2) for w in range(NW)]
                for w in range(NW):
                    mma(w, f1[w])
            assert not pending
            for w in range(NW):
                r1, c0, n = tiles[w]
                for j in range(n):
                    for rr in range(16):
                        gr = m0 + r1 * 16 + rr
                        for cc in range(16):
                            col = n0 + (c0 + j) * 16 + cc
                            if gr < M and col < N:
                                C[gr, col] = acc[(w, j)][rr, cc] + bias[col]
    return C


# ------------------------------------------------------------------------------------------------------------------------------
# csrc/ffn_h2.h (split-f16 feed-forward kernel), one workgroup, lane by lane: the f16 planes of the packed stream (mdm.py pack_ffn_h2),
# the ring of K-step slots with the DMA applied early / late, the in-place split of the x2 rows, the plane addressing
# (chunk t of row r at position t ^ (r & 15)), the v_mfma_f32_16x16x32_f16 operand <-> lane maps with the WEIGHTS as the A operand,
# the GELU / split / zero-pad phase and the staged output tile.  Products of halves are exact; sums are taken in float64 here, so the
# result equals the split arithmetic's exact value (the kernel's fp32 accumulation differs from it by rounding only).
H2_KS1, H2_KS2 = 8, 7
H2_NPAIR = H2_KS1 + H2_KS2
H2_P1B, H2_P2B, H2_SLOT = 13 * 2 * 1024, 16 * 2 * 1024, 32768
H2_SLICE_BYTES = H2_KS1 * H2_P1B + H2_KS2 * H2_P2B


def h2_step_off(P):
    return P * H2_P1B if P < H2_KS1 else H2_KS1 * H2_P1B + (P - H2_KS1) * H2_P2B


def h2_step_ins(P):
    return 0 if P >= H2_NPAIR else (H2_P1B // 1024 if P < H2_KS1 else H2_P2B // 1024)


def emulate_h2_workgroup(x2, pack_words, b1p, b2, mt, sl, late, bm):
    from interdiff_amd.mdm import split_f16
    tt_n = bm // 16
    S = 2 if bm == 64 else 3
    M = x2.shape[0]
    m0 = mt * bm
    stream = np.frombuffer(np.ascontiguousarray(pack_words).tobytes(), np.uint8)[sl * H2_SLICE_BYTES:(sl + 1) * H2_SLICE_BYTES]
    planes = np.zeros(bm * 1024, np.uint8)                # Xs: row r at r KiB = [hi 512 B | lo' 512 B]
    ring = np.full(S * H2_SLOT, 0x7e, np.uint8)           # 0x7e7e = a NaN half: stale bytes must never be multiplied
    lane = np.arange(64)
    n, g = lane & 15, lane >> 4
    pending = {}

    def issue_step(P):
        nins = h2_step_ins(P)
        ops = []
        for wave in range(NW):
            for j in range(4):
                i = wave + 8 * j
                if i < nins:
                    assert (8 * j + 7 < nins) or (8 * j < nins and wave < 2)
                    src = h2_step_off(P) + 1024 * i
                    ops.append(((P % S) * H2_SLOT + 1024 * i, stream[src:src + 1024].copy()))
        if late and ops:
            pending[P] = ops
        else:
            for dst, data in ops:
                ring[dst:dst + 1024] = data

    def publish(P):                                        # wait_step(P) + barrier: everything up to step P has landed
        for Q in sorted(k for k in pending if k <= P):
            for dst, data in pending.pop(Q):
                ring[dst:dst + 1024] = data

    def halves(buf, byte_addr):                            # [64 lanes][8] float16 at per-lane byte addresses
        idx = byte_addr[:, None] + np.arange(16)[None, :]
        return buf[idx].copy().view(np.float16).astype(np.float64)

    def write_plane_piece(r, chunk, half, hi4, lo4):       # 4 halves (8 bytes) of row r at chunk position chunk ^ (r & 15), second half if `half`
        a = r * 1024 + ((chunk ^ (r & 15)) << 4) + 8 * half
        planes[a:a + 8] = np.frombuffer(hi4.astype(np.float16).tobytes(), np.uint8)
        planes[a + 512:a + 520] = np.frombuffer(lo4.astype(np.float16).tobytes(), np.uint8)

    # prologue: x2 rows as fp32 (row r at r KiB), steps 0 .. S - 2; split in place by the fetching wave
    rows = np.stack([x2[min(m0 + i, M - 1)] for i in range(bm)]).astype(np.float32)
    for P in range(S - 1):
        issue_step(P)
    for r in range(bm):
        hi, lo = split_f16(rows[r], flush=False)
        for l in range(64):
            write_plane_piece(r, l >> 1, l & 1, hi[4 * l:4 * l + 4], lo[4 * l:4 * l + 4])

    def read_x(s):                                         # B operand of token tile t: [tt][plane] -> [64][8]
        out = []
        for t in range(tt_n):
            a = (16 * t + n) * 1024 + (((g ^ n) ^ (4 * s)) << 4)
            out.append((halves(planes, a), halves(planes, a + 512)))
        return out

    def read_w(P, tile):                                   # A operand: (hi, lo) [64][8]
        a = (P % S) * H2_SLOT + (tile * 2) * 1024 + lane * 16
        return halves(ring, a), halves(ring, a + 1024)

    def mfma(acc, a, b):                                   # D[i][nn] += sum_{g, j} A[lane (i, g)][j] * B[lane (nn, g)][j]; acc [16 i][16 nn]
        A = a.reshape(4, 16, 8)                            # [g][i][j]
        B = b.reshape(4, 16, 8)                            # [g][n][j]
        return acc + np.einsum('gij,gnj->in', A, B)

    res = {}                                               # phase-1 results: (wave, a, t) -> [16 hidden][16 token]
    if tt_n == 2:
        # csrc/ffn_h2.h, 32-row tile in its shipped sixteen-wave form (round 6): the computing waves 0..7 own (hidden tile, token tile) UNITS balanced over the SIMDs' wave pairs --
        # tile ta with both token tiles, tile tb with both (waves 0, 1) or with the ONE token tile w & 1 (waves 2..7: tiles 5, 9, 12 are shared by two waves).  (The DMA issue modelled
        # below is the eight-wave form's; the loader waves of the shipped form issue the same pieces into the same slots.)
        TA, TB = [0, 2, 4, 6, 8, 10, 11, 7], [1, 3, 5, 5, 9, 9, 12, 12]
        hid_tiles = {w: [TA[w], TB[w]] for w in range(NW)}
        units = {(w, a): (list(range(tt_n)) if (a == 0 or w < 2) else [w & 1]) for w in range(NW) for a in range(2)}
    else:
        hid_tiles = {w: ([2 * w, 2 * w + 1] if w < 5 else [5 + w]) for w in range(NW)}
        units = {(w, a): list(range(tt_n)) for w in range(NW) for a in range(len(hid_tiles[w]))}
    assert sorted((hid_tiles[w][a], t) for (w, a), ts in units.items() for t in ts) == [(h, t) for h in range(HS // 16) for t in range(tt_n)]      # every unit exactly once
    accs = {(w, a, t): [np.zeros((16, 16)), np.zeros((16, 16))] for (w, a), ts in units.items() for t in ts}
    frag_prev = None
    for P in range(H2_KS1 + 1):
        if P < H2_KS1:
            publish(P)
            issue_step(P + S - 1)
            xs = read_x(P)
            frag = (xs, {w: [read_w(P, h) for h in hid_tiles[w]] for w in range(NW)})
        if frag_prev is not None:
            xs_p, ws_p = frag_prev
            for w in range(NW):
                for a, (wh, wl) in enumerate(ws_p[w]):
                    for t in units[(w, a)]:
                        xh, xl = xs_p[t]
                        m, c = accs[(w, a, t)]
                        accs[(w, a, t)] = [mfma(m, wh, xh), mfma(mfma(c, wh, xl), wl, xh)]
        frag_prev = frag if P < H2_KS1 else None
    publish(H2_KS1)
    issue_step(H2_KS1 + S - 1)
    bias = np.asarray(b1p, np.float64)[sl * HS:sl * HS + HS]
    for w in range(NW):
        for a, h in enumerate(hid_tiles[w]):
            for t in units[(w, a)]:
                m, c = accs[(w, a, t)]
                pre = (m + c / 2048.0) + bias[16 * h:16 * h + 16, None]            # [hidden i][token nn]
                hid = _gelu(pre).astype(np.float32)
                for nn in range(16):
                    for gg in range(4):
                        hi, lo = split_f16(hid[4 * gg:4 * gg + 4, nn], flush=False)
                        write_plane_piece(16 * t + nn, 2 * h + (gg >> 1), gg & 1, hi, lo)
    for tid in range(bm * 4):
        r, pl, ch = tid >> 2, (tid >> 1) & 1, 26 + (tid & 1)
        a = r * 1024 + pl * 512 + ((ch ^ (r & 15)) << 4)
        planes[a:a + 16] = 0
    # phase 2
    acc2 = {(w, a, t): [np.zeros((16, 16)), np.zeros((16, 16))] for w in range(NW) for a in range(2) for t in range(tt_n)}
    frag_prev = None
    for q in range(H2_KS2 + 1):
        if q < H2_KS2:
            if q > 0:
                publish(H2_KS1 + q)
                issue_step(H2_KS1 + q + S - 1)
            xs = read_x(q)
            frag = (xs, {w: [read_w(H2_KS1 + q, 2 * w + a) for a in range(2)] for w in range(NW)})
        if frag_prev is not None:
            xs_p, ws_p = frag_prev
            for w in range(NW):
                for a, (wh, wl) in enumerate(ws_p[w]):
                    for t in range(tt_n):
                        xh, xl = xs_p[t]
                        m, c = acc2[(w, a, t)]
                        acc2[(w, a, t)] = [mfma(m, wh, xh), mfma(mfma(c, wh, xl), wl, xh)]
        frag_prev = frag if q < H2_KS2 else None
    assert not pending, 'DMA issued but never waited for: %r' % sorted(pending)
    out = np.zeros((bm, D))
    for w in range(NW):
        for a in range(2):
            for t in range(tt_n):
                m, c = acc2[(w, a, t)]
                out[16 * t:16 * t + 16, 16 * (2 * w + a):16 * (2 * w + a) + 16] = (m + c / 2048.0).T
    if sl == 0:
        out = out + rows.astype(np.float64) + np.asarray(b2, np.float64)[None, :]
    return out[:max(0, min(bm, M - m0))]


def emulate_ffn_h2(x2, pack_words, b1p, b2, late, bm=32):
    M = x2.shape[0]
    parts = np.zeros((NSL, M, D))
    for mt in range(-(-M // bm)):
        for sl in range(NSL):
            o = emulate_h2_workgroup(x2, pack_words, b1p, b2, mt, sl, late, bm)
            parts[sl, mt * bm:mt * bm + o.shape[0]] = o
    return parts


# ------------------------------------------------------------------------------------------------------------------------------
# csrc/ffn_h2.h ln_linear_h2_kernel (split-f16 QKV projection), one workgroup: rows normalised, scaled by a power of two and split
# into the plane image; 8 K steps [10 tiles][2 planes][64 lanes][8 halves] through a 3-slot ring (waves 0..3 three DMA instructions per
# step, waves 4..7 two); weights as the A operand; outputs x 2^e + bias.
def emulate_ln_linear_h2(slabs, lnw, lnb, pack_words, bias, N, late):
    from interdiff_amd.mdm import split_f16
    NP_, M, _ = slabs.shape
    QSTEP, nsl = 10 * 2 * 1024, -(-N // 160)
    out = np.zeros((M, N))
    stream_all = np.frombuffer(np.ascontiguousarray(pack_words).tobytes(), np.uint8)
    lane = np.arange(64)
    n, g = lane & 15, lane >> 4
    for mt in range(-(-M // 32)):
        for sl in range(nsl):
            stream = stream_all[sl * 8 * QSTEP:(sl + 1) * 8 * QSTEP]
            planes, ring, pending = np.zeros(32 * 1024, np.uint8), np.full(3 * QSTEP, 0x7e, np.uint8), {}
            scale = np.ones(32)

            def issue(P):
                if P >= 8:
                    return
                ops = []
                for wave in range(NW):
                    for j in range(3 if wave < 4 else 2):
                        i = wave + 8 * j
                        ops.append(((P % 3) * QSTEP + 1024 * i, stream[P * QSTEP + 1024 * i:P * QSTEP + 1024 * (i + 1)].copy()))
                assert sorted(d - (P % 3) * QSTEP for d, _ in ops) == [1024 * i for i in range(20)]
                if late:
                    pending[P] = ops
                else:
                    for d, data in ops:
                        ring[d:d + 1024] = data

            def land(P):
                for Q in sorted(k for k in pending if k <= P):
                    for d, data in pending.pop(Q):
                        ring[d:d + 1024] = data
            issue(0)
            issue(1)
            for r in range(32):
                x = slabs[:, min(mt * 32 + r, M - 1)].astype(np.float32)
                row = x[0].copy()
                for k in range(1, NP_):
                    row = row + x[k]
                if lnw is not None:
                    mean = row.astype(np.float64).mean()
                    row = ((row - mean) / np.sqrt(((row - mean) ** 2).mean() + 1e-5) * lnw + lnb).astype(np.float32)
                amax = np.abs(row).max()
                e = int(np.frexp(amax)[1]) if amax > 0 else 0
                scale[r] = 2.0 ** e
                hi, lo = split_f16(row * np.float32(2.0 ** -e), flush=False)
                assert np.abs(hi.astype(np.float32)).max() < 1.0
                for l in range(64):
                    a = r * 1024 + (((l >> 1) ^ (r & 15)) << 4) + 8 * (l & 1)
                    planes[a:a + 8] = np.frombuffer(hi[4 * l:4 * l + 4].tobytes(), np.uint8)
                    planes[a + 512:a + 520] = np.frombuffer(lo[4 * l:4 * l + 4].tobytes(), np.uint8)

            def halves(buf, addr):
                return buf[addr[:, None] + np.arange(16)[None, :]].copy().view(np.float16).astype(np.float64)
            tiles = {w: ([2 * w, 2 * w + 1] if w < 2 else [w + 2]) for w in range(NW)}
            acc = {(w, a, t): [np.zeros((16, 16)), np.zeros((16, 16))] for w in range(NW) for a in range(len(tiles[w])) for t in range(2)}
            prev = None
            for P in range(9):
                if P < 8:
                    land(P)
                    issue(P + 2)
                    xs = [(halves(planes, (16 * t + n) * 1024 + (((g ^ n) ^ (4 * P)) << 4)), halves(planes, (16 * t + n) * 1024 + 512 + (((g ^ n) ^ (4 * P)) << 4))) for t in range(2)]
                    ws = {w: [(halves(ring, (P % 3) * QSTEP + (2 * c) * 1024 + lane * 16), halves(ring, (P % 3) * QSTEP + (2 * c + 1) * 1024 + lane * 16)) for c in tiles[w]] for w in range(NW)}
                    cur = (xs, ws)
                if prev is not None:
                    xs_p, ws_p = prev
                    for w in range(NW):
                        for a, (wh, wl) in enumerate(ws_p[w]):
                            for t in range(2):
                                xh, xl = xs_p[t]
                                m, c = acc[(w, a, t)]
                                mm = lambda acc_, A_, B_: acc_ + np.einsum('gij,gnj->in', A_.reshape(4, 16, 8), B_.reshape(4, 16, 8))
                                acc[(w, a, t)] = [mm(m, wh, xh), mm(mm(c, wh, xl), wl, xh)]
                prev = cur if P < 8 else None
            assert not pending
            for w in range(NW):
                for a, c in enumerate(tiles[w]):
                    for t in range(2):
                        m, cc = acc[(w, a, t)]
                        tile = (m + cc / 2048.0).T * scale[16 * t:16 * t + 16, None]            # [token][col]
                        for rr in range(16):
                            gr = mt * 32 + 16 * t + rr
                            if gr >= M:
                                continue
                            for k in range(16):
                                col = sl * 160 + 16 * c + k
                                if col < N:
                                    out[gr, col] = tile[rr, k] + bias[col]
    return out
