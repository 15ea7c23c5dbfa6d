"""Host side of the MDM denoiser seam: ``model(x, ts, y=...) -> x0``.

Mirrors the reference's call surface (model/diffusion_smpl.py:239-246 ``MDM.forward``; invoked
by the sampler as ``model(x, t, **model_kwargs)``, diffusion/gaussian_diffusion.py:305) on top
of ``interdiff_mdm_forward`` / ``interdiff_mdm_prepare_memory`` (include/interdiff_hip.h).
Weights are packed ONCE from a state_dict that uses the reference's key names.

Weights-only constants folded at pack time (host, once):
  * Qc[n][j][:]  : the 10 learned queries, per-head unit-normalised and /sqrt(64)
                   (sublayers.py:18-35), times LocalAttention's scale 256^-0.5, rotated by the
                   rotary position embedding difference (2 - j) so that
                   logit[t,n,j] = <Qc[n,j], x[t+j-1]>   (SURVEY.md appendix B.2);
  * temb[t][:]   : time_embed(pe[t]) for every diffusion step (layers.py:42-43) -- depends on
                   the weights and t only;
  * W_in = [bodyEmbedding | objEmbedding] ([256][C]), W_out = [bodyFinalLinear ; objFinalLinear].
"""
import ctypes as C
import math
import os
import numpy as np
import torch
from . import _lib

D, FF, HEADS, NQ, MEM, LAYERS = 256, 1024, 4, 10, 10, 8
MEM_MAX = 16                # IDF_MDM_MEM_MAX: longest memory (cond rows) the library folds; MEM = 10 takes the compact fast layout
QAN_LAYERS = (1, 2, 3, 4, 5, 6)
ROTARY_DEFAULT = True
FFN_MATH_DEFAULT = 'split'
ROWBLOCK_MATH_DEFAULT = 'split'


def positional_table(max_len=5000, d=D):
    """PositionalEncoding.pe (layers.py:14-19), computed the same way (torch fp32 on the host)."""
    pe = torch.zeros(max_len, d)
    position = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div = torch.exp(torch.arange(0, d, 2).float() * (-np.log(10000.0) / d))
    pe[:, 0::2] = torch.sin(position * div)
    pe[:, 1::2] = torch.cos(position * div)
    return pe.numpy()


def qan_constants(queries, rotary=ROTARY_DEFAULT, heads=HEADS):
    """queries [NQ, D] -> Qc [NQ, 3, D] float32 (float64 arithmetic on the host)."""
    q = np.asarray(queries, dtype=np.float64)
    n, d = q.shape
    qh = q.reshape(n, heads, d // heads)
    qh = qh / (np.sqrt((qh * qh).sum(-1, keepdims=True)) + 1e-6) / math.sqrt(d // heads)
    q = qh.reshape(n, d) * d ** -0.5
    out = np.empty((n, 3, d))
    if not rotary:
        out[:] = q[:, None, :]
        return out.astype(np.float32)
    inv = 1.0 / (10000.0 ** (np.arange(0, d, 2, dtype=np.float64) / d))
    theta = np.concatenate([inv, inv])
    half = d // 2
    rot = np.concatenate([-q[:, half:], q[:, :half]], axis=1)
    for j in range(3):
        p = 2 - j                      # <R_2 q, R_j k> = <R_{2-j} q, k>
        out[:, j] = q * np.cos(p * theta) + rot * np.sin(p * theta)
    return out.astype(np.float32)


def qan_fragments(qc):
    """Qc [NQ, 3, D] -> the row block's MFMA B-operand fragment order [16 k-groups][3 taps][4 kq][NQ][4] (csrc/denoiser.hip FRAG note):
    lane (kq, li < NQ) holds Qc[li, tap, 16 kg + 4 kq : +4]; a wave's load instruction reads 4 * NQ * 16 contiguous bytes."""
    qc = np.asarray(qc, np.float32)
    nq, _, d = qc.shape
    out = np.empty((d // 16, 3, 4, nq, 4), np.float32)
    for kg in range(d // 16):
        for kq in range(4):
            out[kg, :, kq] = qc[:, :, 16 * kg + 4 * kq:16 * kg + 4 * kq + 4].transpose(1, 0, 2)
    return out


def qan_fragments_h2(qc):
    """Qc [NQ, 3, D] -> the split-f16 row block's B-operand fragments (csrc/denoiser.hip G_H2 note): [4 waves = K quarter][2 K steps][3 taps]
    [2 planes][4 kq][NQ][8 halves], lane (kq, li < NQ) holds plane(Qc[li, tap, 64 w + 32 s + 8 kq : +8]) -- the v_mfma_f32_16x16x32_f16 operand
    of that lane.  Returned as float32 words (two halves each), the same 7680 words as ``qan_fragments``."""
    qc = np.asarray(qc, np.float32)
    nq, _, d = qc.shape
    hi, lo = split_f16(qc)
    out = np.empty((d // 64, 2, 3, 2, 4, nq, 8), np.float16)
    for w in range(d // 64):
        for s in range(2):
            for kq in range(4):
                k0 = 64 * w + 32 * s + 8 * kq
                out[w, s, :, 0, kq] = hi[:, :, k0:k0 + 8].transpose(1, 0, 2)
                out[w, s, :, 1, kq] = lo[:, :, k0:k0 + 8].transpose(1, 0, 2)
    return np.ascontiguousarray(out).view(np.float32).reshape(-1)


def pack_tail_h2(w_out, w_in):
    """W_out [144, 256] (bodyFinalLinear ; objFinalLinear) and W_in [256, 144] (bodyEmbedding | objEmbedding) -> the plane fragments of the
    split-f16 step-tail kernel (csrc/tail_h2.h): for every output tile nt (16 rows of the weight), K step s and plane, lane (li, kq) holds
    plane(W[16 nt + li][32 s + 8 kq : + 8]) (zero past the matrix) -- [tiles][K steps][2 planes][64 lanes][8 halves], as float32 words.
    Returns (out_w_h2 [36864 words], in_w_h2 [40960 words]); weights keep the hi plane's flush rule (split_f16)."""
    def frags(w, ntile, kstep):
        w = np.asarray(w, np.float32)
        wp = np.zeros((16 * ntile, 32 * kstep), np.float32)
        wp[:w.shape[0], :w.shape[1]] = w
        hi, lo = split_f16(wp)
        out = np.empty((ntile, kstep, 2, 4, 16, 8), np.float16)              # [...][kq][li][8]: lane = 16 kq + li
        for s in range(kstep):
            for kq in range(4):
                k0 = 32 * s + 8 * kq
                out[:, s, 0, kq] = hi[:, k0:k0 + 8].reshape(ntile, 16, 8)
                out[:, s, 1, kq] = lo[:, k0:k0 + 8].reshape(ntile, 16, 8)
        return np.ascontiguousarray(out).view(np.float32).reshape(-1)
    w_out, w_in = np.asarray(w_out, np.float32), np.asarray(w_in, np.float32)
    assert w_out.shape == (144, D) and w_in.shape[0] == D and w_in.shape[1] >= 144
    return frags(w_out, 9, 8), frags(w_in[:, :144], 16, 5)


def ln_h2_range_ok(*gamma_beta_pairs):
    """A LayerNorm output is bounded by sqrt(D - 1) max|gamma| + max|beta| (|normalised element| <= sqrt(D - 1) < 16): True when every
    given (gamma, beta) keeps its rows below the f16 limit with the margin of H2_LIMIT -- what the split-f16 row block asks of its A operands."""
    for gam, bet in gamma_beta_pairs:
        gam, bet = np.asarray(gam, np.float64), np.asarray(bet, np.float64)
        if not (np.isfinite(gam).all() and np.isfinite(bet).all() and 16.0 * np.abs(gam).max() + np.abs(bet).max() < H2_LIMIT):
            return False
    return True


def _np(t):
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


FFN_SLICE_H = 208           # hidden units per slice of the fused FFN kernel (13 MFMA column tiles; the last slice takes the rest)
_G4 = np.array([0, 3, 2, 1])


def ffn_slices(n=_lib.FFN_SLICES):
    """(first hidden unit, count) of every slice -- csrc/ffn.h: all slices are FFN_SLICE_H wide, 5 x 208 = 1040 >= 1024, the
    units past 1023 are zero weights / zero bias (gelu(0) = 0 times zero columns: they contribute exactly nothing)."""
    return [(FFN_SLICE_H * s, FFN_SLICE_H) for s in range(n)]


def _swizzle16(block):
    """[rows][16] floats -> the LDS image the kernel reads with conflict-free ds_read_b128: the four 16-byte cells of a row are
    stored at position (cell ^ g4[(row >> 2) & 3]) (an involution, applied again by the reader)."""
    rows = block.shape[0]
    cells = block.reshape(rows, 4, 4)
    pos = np.arange(4)[None, :] ^ _G4[(np.arange(rows) >> 2) & 3][:, None]          # cell stored at position p is cell p ^ key
    return np.take_along_axis(cells, pos[:, :, None], axis=1).reshape(rows, 16)


def pack_ffn(w1, w2):
    """linear1.weight [1024,256], linear2.weight [256,1024] -> the fused FFN kernel's weight stream (csrc/ffn.h): per hidden slice
    the 16 k-group chunks of W1 ([208 rows][16 k]) followed by the 13 k-group chunks of W2 ([256 rows][16 k]), every chunk
    already in its swizzled LDS image, so that a workgroup's LDS-DMA is a linear copy of a contiguous 416-KiB run."""
    w1, w2 = np.asarray(w1, np.float32), np.asarray(w2, np.float32)
    assert w1.shape == (FF, D) and w2.shape == (D, FF)
    hp = FFN_SLICE_H * _lib.FFN_SLICES
    w1p, w2p = np.zeros((hp, D), np.float32), np.zeros((D, hp), np.float32)
    w1p[:FF], w2p[:, :FF] = w1, w2
    out = []
    for h0, hs in ffn_slices():
        for g in range(D // 16):
            out.append(_swizzle16(w1p[h0:h0 + hs, 16 * g:16 * g + 16]).ravel())
        for q in range(hs // 16):
            out.append(_swizzle16(w2p[:, h0 + 16 * q:h0 + 16 * q + 16]).ravel())
    out = np.concatenate(out)
    assert out.size == _lib.FFN_SLICES * (16 * FFN_SLICE_H * 16 + (FFN_SLICE_H // 16) * D * 16)
    return out


H2_KS1, H2_KS2 = D // 32, (FFN_SLICE_H + 31) // 32            # K steps of 32 in the two phases of the split-f16 kernel (csrc/ffn_h2.h): 8, 7
H2_SLICE_FLOATS = (H2_KS1 * (FFN_SLICE_H // 16) * 2 * 1024 + H2_KS2 * (D // 16) * 2 * 1024) // 4       # 110592 floats = 432 KiB per slice
H2_LIMIT = 60000.0          # |value| every operand of the split-f16 kernel must provably stay below (f16 max 65504)


def split_f16(a, flush=True):
    """fp32 array -> (hi, lo') float16 planes with a = hi + lo' / 2048 up to 2^-23 |a| (csrc/ffn_h2.h split1, same roundings):
    hi = f16(a), 0 where |a| < 2^-14 (no subnormal in the hi plane: how the WEIGHT planes are packed); lo' = f16((a - hi) * 2^11) (the residual
    is exact in fp32).  ``flush=False``: the kernels' in-flight split of activations (split4_pk: hi may be subnormal, which gfx950's f16 MFMA honours)."""
    a = np.asarray(a, np.float32)
    with np.errstate(over='ignore'):
        hi = a.astype(np.float16)
    if flush:
        hi[np.abs(a) < np.float32(2.0 ** -14)] = 0
    lo = ((a - hi.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
    return hi, lo


def _h2_fragments(w, k0):
    """w [16 T rows][K] fp32, K step at k0 -> [T tiles][2 planes][64 lanes][8] float16: lane (i = lane & 15, g = lane >> 4) of tile t
    holds plane(w[16 t + i][k0 + 8 g + j]), j = 0..7 (zero past K) -- the v_mfma_f32_16x16x32_f16 operand of that lane, so that a
    wave's ds_read_b128 of a (tile, plane) fragment is 1 KiB contiguous in lane order."""
    rows, K = w.shape
    blk = np.zeros((rows, 32), np.float32)
    kk = min(32, K - k0)
    blk[:, :kk] = w[:, k0:k0 + kk]
    hi, lo = split_f16(blk)
    planes = np.stack([hi, lo], axis=0).reshape(2, rows // 16, 16, 4, 8)          # [plane][tile][i][g][j]
    return np.ascontiguousarray(planes.transpose(1, 0, 3, 2, 4)).reshape(-1)      # [tile][plane][g][i][j] = [tile][plane][lane][j]


def pack_ffn_h2(w1, w2):
    """linear1.weight [1024,256], linear2.weight [256,1024] -> the split-f16 FFN kernel's weight stream (csrc/ffn_h2.h), as float32
    words for the arena: per hidden slice the 8 K steps of W1[slice] ([13 tiles][2 planes][64 lanes][8 halves] = 26 KiB each) followed
    by the 7 K steps of W2[:, slice] ([16 tiles][2 planes][64][8] = 32 KiB each, hidden units past 208 zero)."""
    w1, w2 = np.asarray(w1, np.float32), np.asarray(w2, np.float32)
    assert w1.shape == (FF, D) and w2.shape == (D, FF)
    hp = FFN_SLICE_H * _lib.FFN_SLICES
    w1p, w2p = np.zeros((hp, D), np.float32), np.zeros((D, hp), np.float32)
    w1p[:FF], w2p[:, :FF] = w1, w2
    out = []
    for h0, hs in ffn_slices():
        for s in range(H2_KS1):
            out.append(_h2_fragments(w1p[h0:h0 + hs], 32 * s))
        for q in range(H2_KS2):
            out.append(_h2_fragments(w2p[:, h0:h0 + hs], 32 * q))
    out = np.concatenate(out)
    assert out.dtype == np.float16 and out.size * 2 == _lib.FFN_SLICES * H2_SLICE_FLOATS * 4
    return out.view(np.float32)


def ffn_h2_range_ok(w1, b1, w2, ln_w, ln_b):
    """True when NO operand of this layer's feed-forward block can leave the f16 range, whatever the input: the block's input rows are
    LayerNorm outputs, |x2| <= sqrt(255) max|gamma| + max|beta| < 16 max|gamma| + max|beta|; the hidden activations obey
    |gelu(pre)| <= |pre| <= max_row ||W1_row||_1 max|x2| + max|b1|; and the weights themselves are checked.  (A caller that hands
    interdiff_mdm_ffn arbitrary rows -- the standalone op -- owns this bound itself.)"""
    w1, w2 = np.asarray(w1, np.float64), np.asarray(w2, np.float64)
    xmax = 16.0 * np.abs(np.asarray(ln_w, np.float64)).max() + np.abs(np.asarray(ln_b, np.float64)).max()
    hmax = np.abs(w1).sum(axis=1).max() * xmax + np.abs(np.asarray(b1, np.float64)).max()
    vals = (xmax, hmax, np.abs(w1).max(), np.abs(w2).max())
    return bool(np.all(np.isfinite(vals)) and max(vals) < H2_LIMIT)


QKV_SLICE = 160          # output columns per workgroup of the LayerNorm+linear kernel (csrc/ffn.h LHS)


def pack_linear160(w):
    """[N,256] weight -> the LayerNorm+linear kernel's stream (csrc/ffn.h ln_linear_kernel): per 160-row slice (the last one zero-padded
    past N) the 16 k-group chunks [160 rows][16 k], each in the swizzled LDS image of ``_swizzle16``."""
    w = np.asarray(w, np.float32)
    assert w.shape[1] == D
    ns = -(-w.shape[0] // QKV_SLICE)
    wp = np.zeros((ns * QKV_SLICE, D), np.float32)
    wp[:w.shape[0]] = w
    out = [_swizzle16(wp[n0:n0 + QKV_SLICE, 16 * g:16 * g + 16]).ravel() for n0 in range(0, wp.shape[0], QKV_SLICE) for g in range(D // 16)]
    return np.concatenate(out)


def pack_linear160_h2(w):
    """[N,256] weight -> the split-f16 QKV kernel's stream (csrc/ffn_h2.h ln_linear_h2_kernel), as float32 words: per 160-row slice (the
    last one zero-padded past N) the 8 K steps [10 tiles][2 planes][64 lanes][8 halves] (20 KiB each) of ``_h2_fragments``."""
    w = np.asarray(w, np.float32)
    assert w.shape[1] == D
    ns = -(-w.shape[0] // QKV_SLICE)
    wp = np.zeros((ns * QKV_SLICE, D), np.float32)
    wp[:w.shape[0]] = w
    out = np.concatenate([_h2_fragments(wp[n0:n0 + QKV_SLICE], 32 * s) for n0 in range(0, wp.shape[0], QKV_SLICE) for s in range(D // 32)])
    assert out.size * 2 == ns * QKV_SLICE * D * 4
    return out.view(np.float32)


def qkv_bounds(w, b):
    """in_proj weight [768,256] and bias [768] -> the eight floats the split-f16 QKV kernel finds right behind its weight stream when it writes the self-attention's f16 planes
    (csrc/ffn_h2.h ln_linear_h2_kernel<.., PLANES>): max L1 norm of the q / k / v rows (x 1.001: the kernel's own rounding), max |bias| of the q / k / v rows, two spare.
    The kernel divides every input row by its own power of two (|x'| < 1), so |out_c| <= 2^e_row ||W_c||_1 + |b_c|."""
    w, b = np.asarray(w, np.float64), np.asarray(b, np.float64)
    l1 = np.abs(w).sum(1)
    out = [l1[t * D:(t + 1) * D].max() * 1.001 for t in range(3)] + [np.abs(b[t * D:(t + 1) * D]).max() for t in range(3)] + [0.0, 0.0]
    return np.asarray(out, np.float32)


QKV_BOUND_SLACK_MAX = 16.0      # a (type, head) group may sit at most this factor below the bound all heads of its type are divided by (4 of a plane pair's 22 bits)


def qkv_bounds_usable(w, b, bnd):
    """Whether the per-TYPE bounds of ``qkv_bounds`` are good enough to scale every head's planes by: finite, far from overflow, and no (type, head) group
    whose own bound lies more than QKV_BOUND_SLACK_MAX below its type's (one outlier row or head would push every other head's planes down by that slack and
    cost their small elements range: ADVICE r05).  Such a layer keeps fp32 rows out of the projection and the attention splits them by the measured
    amax per (clip, head) tile, as before round 5."""
    w, b, bnd = np.asarray(w, np.float64), np.asarray(b, np.float64), np.asarray(bnd, np.float64)
    if not np.isfinite(bnd).all() or float(bnd[:3].max()) * 2.0 ** 60 >= 3e38:
        return False
    l1 = np.abs(w).sum(1).reshape(3, HEADS, D // HEADS).max(2) * 1.001 + np.abs(b).reshape(3, HEADS, D // HEADS).max(2)      # [type][head]: the bound a head would get on its own (unit row scale)
    whole = bnd[:3] + bnd[3:6]
    return bool((whole[:, None] <= QKV_BOUND_SLACK_MAX * np.maximum(l1, 1e-30)).all())


def sa_out_fragments(w):
    """out_proj.weight [256 out, 256 in] -> MFMA B-operand fragment order of the attention kernel's out-projection tail
    (csrc/denoiser.hip self_attn_kernel<true>): [head][wave = output column quarter][k-group of 16][column tile][lane = kq*16+li][4],
    element e of lane (kq, li) = w[(wave*4 + tile)*16 + li][head*64 + 16*kgroup + 4*kq + e]."""
    w = np.asarray(w, np.float32)
    assert w.shape == (D, D)
    f = w.reshape(4, 4, 16, HEADS, 4, 4, 4)          # [wave][tile][li][head][kgroup][kq][e]
    return np.ascontiguousarray(f.transpose(3, 0, 4, 1, 5, 2, 6)).ravel()      # [head][wave][kgroup][tile][kq][li][e]


def sa_out_fragments_h2(w):
    """out_proj.weight [256 out, 256 in] -> the split-f16 attention kernel's out-projection fragments (csrc/attn_h2.h): [head][16 output column tiles]
    [2 K steps][2 planes][64 lanes][8 halves]; lane (li, kq) of (head, tile ct, step s) holds plane(w[16 ct + li][64 head + 32 s + 8 kq : + 8]).  As float32 words."""
    w = np.asarray(w, np.float32)
    assert w.shape == (D, D)
    hi, lo = split_f16(w)
    out = np.empty((HEADS, 16, 2, 2, 4, 16, 8), np.float16)              # [...][kq][li][8]
    for h in range(HEADS):
        for s in range(2):
            for kq in range(4):
                k0 = 64 * h + 32 * s + 8 * kq
                out[h, :, s, 0, kq] = hi[:, k0:k0 + 8].reshape(16, 16, 8)
                out[h, :, s, 1, kq] = lo[:, k0:k0 + 8].reshape(16, 16, 8)
    return np.ascontiguousarray(out).view(np.float32).reshape(-1)


def pad_ffn_bias(b1):
    """linear1.bias [1024] -> [5 * 208] zero-padded + one spare KiB (the kernel fetches a slice's bias with one 1-KiB DMA)."""
    out = np.zeros(FFN_SLICE_H * _lib.FFN_SLICES + 256, np.float32)
    out[:FF] = np.asarray(b1, np.float32)
    return out


class _Arena:
    def __init__(self):
        self.parts, self.n = [], 0

    def add(self, a):
        a = np.ascontiguousarray(a, dtype=np.float32).ravel()
        off = self.n
        pad = (-a.size) % 64
        self.parts.append(a)
        if pad:
            self.parts.append(np.zeros(pad, np.float32))
        self.n += a.size + pad
        return off

    def tensor(self, device):
        return torch.from_numpy(np.concatenate(self.parts)).to(device)


def pack_mdm_weights(sd, device, n_steps=1000, max_T=512, rotary=ROTARY_DEFAULT):
    """state_dict (reference key names, tensors or arrays) -> (MdmWeights struct, arena tensor)."""
    g = lambda k: _np(sd[k]).astype(np.float32)
    ar = _Arena()
    w = _lib.MdmWeights()
    win = np.concatenate([g('bodyEmbedding.weight'), g('objEmbedding.weight')], axis=1)      # [256, C]
    w.C = win.shape[1]
    if win.shape[1] % 4:                       # any token width (BASELINE config #1: 106): rows zero-padded to a multiple of 4 floats, the kernel zero-fills x's missing channels
        win = np.concatenate([win, np.zeros((win.shape[0], (-win.shape[1]) % 4), np.float32)], axis=1)
    w.in_w = ar.add(win)                       # [256][(C + 3) & ~3]: the embed GEMM's W[N][K]
    w.in_b = ar.add(g('bodyEmbedding.bias') + g('objEmbedding.bias'))
    w.out_w = ar.add(np.concatenate([g('bodyFinalLinear.weight'), g('objFinalLinear.weight')], axis=0))
    w.out_b = ar.add(np.concatenate([g('bodyFinalLinear.bias'), g('objFinalLinear.bias')]))
    w.out_w_h2 = w.in_w_h2 = 0                 # split-f16 plane fragments of the two token GEMMs (csrc/tail_h2.h): SMPL token width, values inside the f16 range
    wout_full = np.concatenate([g('bodyFinalLinear.weight'), g('objFinalLinear.weight')], axis=0)
    if w.C == 144 and np.abs(wout_full).max() < H2_LIMIT and np.abs(win).max() < H2_LIMIT:
        oh2, ih2 = pack_tail_h2(wout_full, win)
        w.out_w_h2, w.in_w_h2 = ar.add(oh2), ar.add(ih2)
    # the heads GEMM's A operand is LN3 of the LAST layer: its own range proof (the embedding side is row-scaled in the kernel and needs none)
    pl = 'decoder.layers.%d.' % (LAYERS - 1)
    w.tail_h2_ok = 1 if ln_h2_range_ok((g(pl + 'norm3.weight'), g(pl + 'norm3.bias'))) else 0
    pe = g('PositionalEmbedding.pe')[:, 0] if 'PositionalEmbedding.pe' in sd else positional_table()
    if n_steps > pe.shape[0] or max_T > pe.shape[0]:
        raise ValueError('positional table too short')
    # time embedding table (torch fp32 on the host, same op order as TimestepEmbedder)
    t0w, t0b = torch.from_numpy(g('embedTimeStep.time_embed.0.weight')), torch.from_numpy(g('embedTimeStep.time_embed.0.bias'))
    t2w, t2b = torch.from_numpy(g('embedTimeStep.time_embed.2.weight')), torch.from_numpy(g('embedTimeStep.time_embed.2.bias'))
    h = torch.from_numpy(pe[:n_steps].copy()) @ t0w.T + t0b
    temb = torch.nn.functional.silu(h) @ t2w.T + t2b
    w.n_steps = n_steps
    w.temb_table = ar.add(temb.numpy())
    w.max_T = max_T
    w.pe = ar.add(pe[:max_T])
    for l in range(LAYERS):
        p, ly = 'decoder.layers.%d.' % l, w.layer[l]
        ly.is_qan = 1 if (p + 'queries') in sd else 0
        if ly.is_qan:
            qcn = qan_constants(g(p + 'queries'), rotary)
            ly.qc = ar.add(qan_fragments(qcn))
            ly.qc_h2 = ar.add(qan_fragments_h2(qcn)) if np.abs(qcn).max() < H2_LIMIT else 0
            ly.wk = ar.add(g(p + 'wk').reshape(-1))
            # split-f16 row block: its A operands are LN_prev (the previous layer's norm3; none in front of layer 0) and this layer's norm1 outputs
            pp = 'decoder.layers.%d.' % (l - 1)
            ly.rb_h2_ok = 1 if (l > 0 and ln_h2_range_ok((g(pp + 'norm3.weight'), g(pp + 'norm3.bias')), (g(p + 'norm1.weight'), g(p + 'norm1.bias')))) else 0
        else:
            ly.rb_h2_ok = 1 if ln_h2_range_ok((g(p + 'norm1.weight'), g(p + 'norm1.bias'))) else 0
            ly.sa_in_w = ar.add(g(p + 'self_attn.in_proj_weight'))
            ly.sa_in_pack = ar.add(pack_linear160(g(p + 'self_attn.in_proj_weight')))
            if np.abs(g(p + 'self_attn.in_proj_weight')).max() < H2_LIMIT:      # the stream with the output bounds right behind it (include/interdiff_hip.h qkv_bounds_ok)
                bnd = qkv_bounds(g(p + 'self_attn.in_proj_weight'), g(p + 'self_attn.in_proj_bias'))
                ly.sa_in_pack_h2 = ar.add(np.concatenate([pack_linear160_h2(g(p + 'self_attn.in_proj_weight')), bnd]))
                ly.qkv_bounds_ok = 1 if qkv_bounds_usable(g(p + 'self_attn.in_proj_weight'), g(p + 'self_attn.in_proj_bias'), bnd) else 0
            else:
                ly.sa_in_pack_h2, ly.qkv_bounds_ok = 0, 0
            ly.sa_in_b = ar.add(g(p + 'self_attn.in_proj_bias'))
            ly.sa_out_w = ar.add(g(p + 'self_attn.out_proj.weight'))
            ly.sa_out_frag = ar.add(sa_out_fragments(g(p + 'self_attn.out_proj.weight')))
            ly.sa_out_frag_h2 = ar.add(sa_out_fragments_h2(g(p + 'self_attn.out_proj.weight'))) if np.abs(g(p + 'self_attn.out_proj.weight')).max() < H2_LIMIT else 0
            ly.sa_out_b = ar.add(g(p + 'self_attn.out_proj.bias'))
        cw, cb = g(p + 'multihead_attn.in_proj_weight'), g(p + 'multihead_attn.in_proj_bias')
        ly.ca_q_w, ly.ca_q_b = ar.add(cw[:D]), ar.add(cb[:D])
        ly.ca_kv_w, ly.ca_kv_b = ar.add(cw[D:]), ar.add(cb[D:])
        ly.ca_out_w = ar.add(g(p + 'multihead_attn.out_proj.weight'))
        ly.ca_out_b = ar.add(g(p + 'multihead_attn.out_proj.bias'))
        ly.ff1_w, ly.ff1_b = ar.add(g(p + 'linear1.weight')), ar.add(g(p + 'linear1.bias'))
        ly.ff2_w, ly.ff2_b = ar.add(g(p + 'linear2.weight')), ar.add(g(p + 'linear2.bias'))
        ly.ffn_pack = ar.add(pack_ffn(g(p + 'linear1.weight'), g(p + 'linear2.weight')))
        ly.ffn_b1p = ar.add(pad_ffn_bias(g(p + 'linear1.bias')))
        # split-f16 stream (csrc/ffn_h2.h): only when the range proof holds; the block's input is norm2's output (post-norm layer)
        ly.ffn_pack_h2 = (ar.add(pack_ffn_h2(g(p + 'linear1.weight'), g(p + 'linear2.weight')))
                          if ffn_h2_range_ok(g(p + 'linear1.weight'), g(p + 'linear1.bias'), g(p + 'linear2.weight'), g(p + 'norm2.weight'), g(p + 'norm2.bias')) else 0)
        for k in range(3):
            ly.ln_w[k] = ar.add(g(p + 'norm%d.weight' % (k + 1)))
            ly.ln_b[k] = ar.add(g(p + 'norm%d.bias' % (k + 1)))
    # encoder side (MDM._get_embeddings, "next" row): [std, QaN x6, std] without cross-attention
    w.has_encoder = 1 if 'encoder.layers.0.linear1.weight' in sd else 0
    if w.has_encoder:
        for l in range(LAYERS):
            p, ly = 'encoder.layers.%d.' % l, w.enc_layer[l]
            ly.is_qan = 1 if (p + 'queries') in sd else 0
            if ly.is_qan:
                ly.qc = ar.add(qan_fragments(qan_constants(g(p + 'queries'), rotary)))
                ly.wk = ar.add(g(p + 'wk').reshape(-1))
            else:
                ly.sa_in_w = ar.add(g(p + 'self_attn.in_proj_weight'))
                ly.sa_in_pack = ar.add(pack_linear160(g(p + 'self_attn.in_proj_weight')))
                ly.sa_in_pack_h2 = ar.add(pack_linear160_h2(g(p + 'self_attn.in_proj_weight'))) if np.abs(g(p + 'self_attn.in_proj_weight')).max() < H2_LIMIT else 0
                ly.sa_in_b = ar.add(g(p + 'self_attn.in_proj_bias'))
                ly.sa_out_w = ar.add(g(p + 'self_attn.out_proj.weight'))
                ly.sa_out_frag = ar.add(sa_out_fragments(g(p + 'self_attn.out_proj.weight')))
                ly.sa_out_b = ar.add(g(p + 'self_attn.out_proj.bias'))
            ly.ff1_w, ly.ff1_b = ar.add(g(p + 'linear1.weight')), ar.add(g(p + 'linear1.bias'))
            ly.ff2_w, ly.ff2_b = ar.add(g(p + 'linear2.weight')), ar.add(g(p + 'linear2.bias'))
            ly.ffn_pack = ar.add(pack_ffn(g(p + 'linear1.weight'), g(p + 'linear2.weight')))
            ly.ffn_b1p = ar.add(pad_ffn_bias(g(p + 'linear1.bias')))
            ly.ffn_pack_h2 = (ar.add(pack_ffn_h2(g(p + 'linear1.weight'), g(p + 'linear2.weight')))       # encoder layer: the block's input is norm1's output
                              if ffn_h2_range_ok(g(p + 'linear1.weight'), g(p + 'linear1.bias'), g(p + 'linear2.weight'), g(p + 'norm1.weight'), g(p + 'norm1.bias')) else 0)
            for k in range(2):
                ly.ln_w[k] = ar.add(g(p + 'norm%d.weight' % (k + 1)))
                ly.ln_b[k] = ar.add(g(p + 'norm%d.bias' % (k + 1)))
    arena = ar.tensor(device)
    w.arena = arena.data_ptr()
    return w, arena


def pack_pointnet2(sd, device, prefix='pcEmbedding', eps=1e-5):
    """PointNet2Encoder weights (model/layers.py:111-140; pointnet2_ops 3.0.0 naming ``SA_modules.<i>.mlps.<scale>.{0,3,6}`` conv,
    ``{1,4,7}`` BatchNorm) -> (PointNet2 struct, arena).  Eval-mode BatchNorm is folded into the bias-free 1x1 convolutions
    in float64: W' = W * gamma / sqrt(var + eps), b' = beta - mean * gamma / sqrt(var + eps)."""
    f = lambda k: _np(sd[k]).astype(np.float64)
    ar = _Arena()
    pn = _lib.PointNet2()
    for si, dst in ((0, pn.sa1), (1, pn.sa2)):
        for sc in range(2):
            q = '%s.SA_modules.%d.mlps.%d' % (prefix, si, sc)
            for l in range(3):
                W = f('%s.%d.weight' % (q, 3 * l))[:, :, 0, 0]
                bn = '%s.%d' % (q, 3 * l + 1)
                sc_ = f(bn + '.weight') / np.sqrt(f(bn + '.running_var') + eps)
                dst[sc].w[l] = ar.add(W * sc_[:, None])
                dst[sc].b[l] = ar.add(f(bn + '.bias') - f(bn + '.running_mean') * sc_)
                dst[sc].c[l], dst[sc].c[l + 1] = W.shape[1], W.shape[0]
    pn.lin_w, pn.lin_b = ar.add(f(prefix + '.Linear.weight')), ar.add(f(prefix + '.Linear.bias'))
    arena = ar.tensor(device)
    pn.arena = arena.data_ptr()
    return pn, arena


class MDM:
    """Drop-in for the reference denoiser at the sampler seam: ``MDM(state_dict)(x, ts, y={'cond': ...})``."""
    graph_safe = True        # forward() enqueues kernels only (no allocation / sync once warmed up): hipGraph-capturable
    accepts_batch_rows = True        # forward() / forward_step() take ``batch_rows=`` (the sampler's shard / chain plumbing)

    def __init__(self, state_dict, device='cuda', n_steps=1000, rotary=ROTARY_DEFAULT):
        self.lib = _lib.load()
        self.device = torch.device(device)
        self.w, self.arena = pack_mdm_weights(state_dict, self.device, n_steps=n_steps, rotary=rotary)
        self._mem_key, self._mem_cond, self._memctx, self._ws = None, None, None, None
        self._ws_shape, self._ws_pool, self._memctx_pool = None, {}, {}
        self._memlen_of = {}                      # folded-memory buffer address -> (memory length, floats): the length travels with the buffer (_bind_memory)
        self.ffn_rows = 0                           # 0: feed-forward tile by batch size (_pick_ffn_tile); 16 / 32: forced
        # arithmetic of the feed-forward block: 'split' = split-f16 MFMA (csrc/ffn_h2.h: two f16 planes per fp32 operand, three f16 MFMAs per
        # product, fp32 accumulate -- fp32-grade results, layers whose range proof failed at pack time stay exact); 'exact' = fp32 MFMA (csrc/ffn.h)
        self.ffn_math = os.environ.get('INTERDIFF_FFN_MATH', FFN_MATH_DEFAULT)
        if self.ffn_math not in ('split', 'exact'):
            raise ValueError("ffn_math must be 'split' or 'exact'")
        # with ffn_math = 'split': the row block's three contractions as split-f16 products too ('split', layers whose LayerNorm range proof holds) or on
        # the fp32 MFMA ('exact'; INTERDIFF_ROWBLOCK_MATH).  Ignored under ffn_math = 'exact'.
        self.rowblock_math = os.environ.get('INTERDIFF_ROWBLOCK_MATH', ROWBLOCK_MATH_DEFAULT)
        if self.rowblock_math not in ('split', 'exact'):
            raise ValueError("rowblock_math must be 'split' or 'exact'")
        # the split-f16 row block: 8 = the eight-wave kernel (csrc/denoiser.hip rowblock8_kernel, shipped since round 5), 4 = round 4's four-wave kernel (A/B runs)
        self.rowblock_waves = int(os.environ.get('INTERDIFF_ROWBLOCK_WAVES', '8'))
        if self.rowblock_waves not in (4, 8):
            raise ValueError('rowblock_waves must be 4 or 8')
        self.pn = self.pn_arena = None
        if 'pcEmbedding.Linear.weight' in state_dict:
            self.pn, self.pn_arena = pack_pointnet2(state_dict, self.device)

    # -- nn.Module-ish surface the sampler touches (gaussian_diffusion.py:688-689, eval_smpl_short.py:428)
    def parameters(self):
        yield self.arena

    def eval(self):
        return self

    def to(self, device):
        return self

    def _workspace(self, B, T):
        """Workspace for a (B, T) forward.  Buffers are NEVER released or replaced once handed out: their addresses may be
        baked into a captured hipGraph that is replayed long after a call with another shape came by."""
        need = self.lib.interdiff_mdm_workspace_bytes(B, T)
        if self._ws is not None and self._ws.numel() >= need and self._ws_shape == (B, T):
            return self._ws
        ws = self._ws_pool.get((B, T))
        if ws is None or ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=self.device)
            self._ws_pool[(B, T)] = ws
        self._ws, self._ws_shape = ws, (B, T)
        return ws

    def shape_buffer_keys(self):
        """Names of the per-shape buffers the model currently owns (the sampler records which ones a cache entry caused, diffusion.py)."""
        return {('ws',) + k for k in self._ws_pool} | {('memctx', k) for k in self._memctx_pool}

    def forget_shape_buffers(self, keys):
        """Drop the named per-shape buffers (``shape_buffer_keys``): called by the sampler's graph cache for the entries ITS evicted graphs
        had caused -- every graph that baked their addresses in is gone by then.  Buffers in current use are re-created on demand."""
        for k in keys:
            if k[0] == 'ws':
                ws = self._ws_pool.pop(tuple(k[1:]), None)
                if ws is not None and ws is self._ws:
                    self._ws = self._ws_shape = None
            elif k[0] == 'memctx':
                mc = self._memctx_pool.pop(k[1], None)
                if mc is not None and mc is self._memctx:
                    self._memctx = self._mem_key = self._mem_cond = None

    def release_shape_buffers(self):
        """Drop the per-shape workspaces and memory contexts.  EVERY hipGraph that captured a call of this model -- the sampler's
        per-denoiser cache (``_graph_cache``), bench / integrator graphs around ``forward`` -- must have been destroyed first: their
        baked-in addresses die here.  Nothing in the package calls this on its own."""
        self._ws_pool.clear()
        self._memctx_pool.clear()
        self._ws = self._ws_shape = self._memctx = self._mem_key = self._mem_cond = None

    def memctx_floats(self, B, mem_len=None):
        """Floats of a folded-memory buffer for B clips at memory length ``mem_len`` (default: the length of the last ``prepare_memory``, else 10)."""
        return self.lib.interdiff_mdm_memctx_floats_for(B, mem_len or self.mem_len)

    @property
    def mem_len(self):
        """Memory length (rows of ``cond``) the handle is set to: eval_smpl_short.py:376 takes it from the CLI (--past_len); 10 in every BASELINE config."""
        return int(self.w.mem_len) or MEM

    def workspace_bytes(self, B, T):
        return self.lib.interdiff_mdm_workspace_bytes(B, T)

    def prepare_memory(self, cond, into=None):
        """Fold the constant memory ``cond`` [MEM,B,256] into the per-sample cross-attention operands.  ``into``: a caller-owned
        buffer of ``memctx_floats(B)`` floats to fill instead of the model's own (a chain of a split batch, diffusion.py); the
        model's notion of "current memory" is left alone then."""
        if not (1 <= cond.shape[0] <= MEM_MAX) or cond.shape[2] != D:
            raise ValueError('cond must be [1..%d,B,%d]' % (MEM_MAX, D))
        B = cond.shape[1]
        given = cond
        cond = cond.contiguous()
        L = cond.shape[0]
        need = self.lib.interdiff_mdm_memctx_floats_for(B, L)
        if into is not None:
            if into.numel() != need or into.dtype != torch.float32 or not into.is_contiguous():
                raise ValueError('into must be %d contiguous floats' % need)          # (nothing of the handle has changed yet)
            ws = self._workspace(B, 16)
            self.w.mem_len = L                      # read by the C call below; every later forward sets it from the buffer it is handed (_bind_memory)
            _lib.check(self.lib.interdiff_mdm_prepare_memory(C.byref(self.w), _lib.dptr(cond, torch.float32), B,
                                                             _lib.dptr(into), _lib.dptr(ws), ws.numel(), _lib.stream()),
                       'mdm_prepare_memory')
            self._memlen_of[into.data_ptr()] = (L, need)
            return into
        # one buffer per (batch size, memory length), reused for every sample and never released: its address may be baked into a captured hipGraph
        pool_key = B if cond.shape[0] == MEM else (B, cond.shape[0])
        memctx = self._memctx_pool.get(pool_key)
        if memctx is None or memctx.numel() != need:
            memctx = torch.empty(need, dtype=torch.float32, device=self.device)
            self._memctx_pool[pool_key] = memctx
        ws = self._workspace(B, 16)
        self.w.mem_len = L
        _lib.check(self.lib.interdiff_mdm_prepare_memory(C.byref(self.w), _lib.dptr(cond, torch.float32), B,
                                                         _lib.dptr(memctx), _lib.dptr(ws), ws.numel(), _lib.stream()),
                   'mdm_prepare_memory')
        self._memlen_of[memctx.data_ptr()] = (L, need)
        # forward() recognises "same memory as last time" by (address, version, shape) of the tensor it is handed; the tensor itself
        # is kept alive here so that the allocator cannot hand its address to a DIFFERENT cond of the same shape (the outputs of
        # _get_embeddings are written through raw pointers, so their version counter never moves)
        self._mem_key = (given.data_ptr(), given._version, tuple(given.shape))
        self._mem_cond = given
        self._memctx = memctx
        return memctx

    def _bind_memory(self, memctx, B):
        """Set the handle's memory length from the folded buffer a forward is about to read -- the length is a property of the BUFFER (recorded next to it by
        ``prepare_memory``), not of whichever ``prepare_memory`` call came last: a cached model buffer of length 10 stays length 10 after a length-15 fold ``into=`` a
        caller's buffer.  A buffer this model did not fold is identified by its size (the layouts of different lengths differ in size)."""
        rec = self._memlen_of.get(memctx.data_ptr())
        if rec is None or rec[1] != memctx.numel():
            fits = [L for L in range(1, MEM_MAX + 1) if self.lib.interdiff_mdm_memctx_floats_for(B, L) == memctx.numel()]
            if len(fits) != 1:
                raise ValueError('memctx was folded for another batch size or memory length')
            rec = (fits[0], memctx.numel())
        if self.lib.interdiff_mdm_memctx_floats_for(B, rec[0]) != memctx.numel():
            raise ValueError('memctx was folded for another batch size or memory length')
        self.w.mem_len = rec[0]

    def _get_embeddings(self, body_pose, body_trans, obj_angles, obj_trans, obj_points, past_len=10, batch_clips=None):
        """``MDM._get_embeddings`` (model/diffusion_smpl.py:195-223) on tensors instead of the dataset's dict-of-lists:
        body_pose [T,B,66] axis-angle, body_trans [T,B,3], obj_angles [T,B,3] axis-angle, obj_trans [T,B,3],
        obj_points [B,P,3]  ->  (cond [past_len,B,256], gt [T,B,144]).  ``batch_clips``: clips of the WHOLE batch these B are a
        shard of (the encoder's feed-forward tile class follows it, ``_pick_ffn_tile``)."""
        from . import transforms as tr
        if self.pn is None or not self.w.has_encoder:
            raise RuntimeError('this state_dict has no encoder / pcEmbedding weights')
        T, B, _ = body_pose.shape
        pts = obj_points.contiguous().float()
        pc = torch.empty(B, D, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.interdiff_pointnet2_encode(C.byref(self.pn), _lib.dptr(pts), B, pts.shape[1], _lib.dptr(pc), _lib.stream()),
                   'pointnet2_encode')
        body6 = tr.matrix_to_rotation_6d(tr.axis_angle_to_matrix(body_pose.reshape(T, B, -1, 3))).reshape(T, B, -1)
        obj6 = tr.matrix_to_rotation_6d(tr.axis_angle_to_matrix(obj_angles.reshape(T, B, -1, 3))).reshape(T, B, -1)
        gt = torch.cat([body6, body_trans.float(), obj6, obj_trans.float()], dim=2)                       # [T,B,144]
        x_past = gt[:past_len].permute(1, 2, 0).unsqueeze(1).contiguous()                                # [B,1,144,past]
        self._pick_ffn_tile((batch_clips or B) * past_len, B * past_len)
        need = self.lib.interdiff_mdm_encode_workspace_bytes(B, past_len)
        ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        cond = torch.empty(past_len, B, D, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.interdiff_mdm_encode(C.byref(self.w), _lib.dptr(pc), _lib.dptr(x_past), B, past_len, _lib.dptr(cond),
                                                 _lib.dptr(ws), ws.numel(), _lib.stream()), 'mdm_encode')
        return cond, gt

    FFN16_MAX_ROWS, FFN64_MIN_ROWS = 800, 2800      # csrc/ffn.h

    def one_chain_max_rows(self):
        """Up to how many token rows the sampler steps a batch as ONE kernel chain (diffusion.py _graph_loop).  Exact-fp32 feed-forward: 800 (the
        16-row grid's one round; above, two half-batch chains overlap each other's latency: round 3).  Split-f16 feed-forward: 1632 = 51 tiles
        of 32 rows x 5 slices = 255 workgroups, ONE round on 256 CUs -- the kernel owns its CUs (csrc/ffn_h2.h "exclusive CU"), so a second chain can
        no longer slip its small kernels beside it, and up to one round a single chain is ahead (same process, whole samples with correction,
        tools/chains_ab.py: 12 clips 0.2171 vs 0.2252, 16 clips 0.2300 vs 0.2372 ms per step; beyond one round two chains win big: 17 clips 0.2455 vs
        0.3135, 20: 0.2545 vs 0.3310, 24: 0.2875 vs 0.3412, 32: 0.3362 vs 0.3495; re-measured on round 5's final build, where the row block and the
        self-attention own their CUs too: 24 clips 0.290 vs 0.321, 32 clips 0.318 vs 0.323 -- two chains still ahead beyond one round)."""
        if self.ffn_math != 'split':
            return self.FFN16_MAX_ROWS
        if getattr(self, '_n_cu', None) is None:      # one round of 32-row x 5-slice workgroups on THIS device's CUs (256 on MI355X: 1632 rows)
            self._n_cu = torch.cuda.get_device_properties(self.device).multi_processor_count if self.device.type == 'cuda' else 256
        return 32 * (self._n_cu // _lib.FFN_SLICES)


    @classmethod
    def ffn_tile_for_rows(cls, rows):
        """csrc/ffn.h ffn_tile_for_rows, for the rows of a whole batch: 16-row tiles while their grid fits the chip in one round of
        workgroups (<= 800 rows), 64-row tiles from 2800 rows on, 32 in between."""
        return 16 if rows <= cls.FFN16_MAX_ROWS else (32 if rows < cls.FFN64_MIN_ROWS else 64)

    @classmethod
    def ffn_class_for_rows(cls, rows):
        """Rounding class of the feed-forward tile a batch of ``rows`` token rows takes: 32 (the 32-row kernel: one column tile per
        slice is summed in two accumulators) or 16 (the 16- and 64-row kernels, bit-identical to each other)."""
        return 32 if cls.ffn_tile_for_rows(rows) == 32 else 16

    def ffn_graph_key(self, rows):
        """What a captured launch sequence bakes in about the feed-forward block (the sampler's graph cache key, diffusion.py)."""
        return (self.ffn_class_for_rows(rows), self.ffn_math, self.ffn_rows, getattr(self, 'rowblock_math', 'exact'), getattr(self, 'rowblock_waves', 8))

    def _pick_ffn_tile(self, rows, own_rows=None):
        """The fused feed-forward block has 16-, 32- and 64-row kernels (csrc/ffn.h); the 32-row one agrees with the other two to
        rounding, not bit for bit: every launch of one sample must take the same rounding class, whichever way the batch is cut into
        chains (diffusion.py) or shards (dist.py) -- so the class is chosen HERE from ``rows`` = the rows of the WHOLE batch and
        handed down (``tune[IDF_TUNE_FFN]``), not left to the per-launch default.  Inside the 16 / 64 class the tile follows
        ``own_rows`` = the rows of this launch (a rank's 8 clips of a 64-clip batch take the 16-row grid: same bits as the 64-row
        kernel the unsharded batch runs, twice as fast at that size).  ``self.ffn_rows`` (16 / 32 / 64) overrides all of it (A/B runs:
        tools/ffn16_ab.py)."""
        tile = self.ffn_rows
        if not tile:
            tile = self.ffn_tile_for_rows(rows)
            if tile != 32 and own_rows is not None and own_rows != rows:
                tile = 16 if own_rows <= self.FFN16_MAX_ROWS else 64
        self.w.tune[_lib.TUNE['ffn']] = {16: 2, 64: 3}.get(tile, 1)
        self.w.tune[_lib.TUNE['ffn_math']] = (1 if getattr(self, 'rowblock_math', 'exact') == 'split' else 2) if getattr(self, 'ffn_math', 'exact') == 'split' else 0
        if self.w.tune[_lib.TUNE['misc']] in (0, 8):            # (other values of the A/B switch are left to whoever set them)
            self.w.tune[_lib.TUNE['misc']] = 8 if getattr(self, 'rowblock_waves', 8) == 4 else 0

    def arithmetic_report(self, device_verdicts=True):
        """Which arithmetic every contraction of a denoiser forward takes under the current ``ffn_math`` / ``rowblock_math`` selection, layer by
        layer -- the same conditions csrc/denoiser.hip applies (a layer whose f16 range proof failed at pack time stays on the exact fp32-MFMA kernel;
        nothing else reports that).  'split' = split-f16 (two f16 planes per fp32 operand, three f16 MFMAs per product), 'exact' = fp32 MFMA.
        ``device_verdicts`` (on a GPU): the launch-time side of the choice is folded in -- a split-f16 kernel that does not get its CU to itself on this device
        (csrc/common.h idf_exclusive_cu; or sits on the debug deny list) runs as its fp32 counterpart, and the contraction is reported 'exact' with the
        kernel named under ``not_exclusive``: a silent downgrade shows up here and in bench.py's line."""
        split = self.ffn_math == 'split'
        rb_split = split and self.rowblock_math == 'split'
        w = self.w
        failing = []
        if device_verdicts and self.device.type == 'cuda':
            text, _ = _lib.exclusive_cu_report()
            failing = [ln.split('  dev ')[0].strip() for ln in text.splitlines() if ln.strip() and not ln.rstrip().endswith('  exclusive')]
        bad = lambda *tags: any(all(t in name for t in tags) for name in failing)
        tile = {1: '32 rows', 2: '16 rows', 3: '64 rows'}.get(int(w.tune[_lib.TUNE['ffn']]), '')
        layers = []
        for l in range(LAYERS):
            ly = w.layer[l]
            kind = 'QaN' if ly.is_qan else 'std'
            d = dict(layer=l, kind='qan' if ly.is_qan else 'std',
                     ffn='split' if (split and ly.ffn_pack_h2 and not bad('ffn_h2_kernel', tile)) else 'exact',
                     # the eight-wave row block, else round 4's four-wave split kernel, else fp32: 'exact' only when both split forms of this kind are refused
                     rowblock='split' if (rb_split and ly.rb_h2_ok and (not ly.is_qan or (ly.qc_h2 and l > 0)) and not (bad('rowblock8_kernel<' + kind) and bad('rowblock_kernel<' + kind))) else 'exact')
            if not ly.is_qan:
                slabs = '1 slab' if l == 0 else '5 slabs'
                attn_packed = bool(split and ly.sa_out_frag_h2 and w.tune[_lib.TUNE['misc']] != 6)
                planes = bool(attn_packed and ly.sa_in_pack_h2 and ly.qkv_bounds_ok and w.tune[_lib.TUNE['misc']] != 9 and not bad('ln_linear_h2_kernel<' + slabs + ', planes out>') and not bad('self_attn_h2_kernel<planes in>'))
                d['qkv'] = 'split' if (split and ly.sa_in_pack_h2 and (planes or not any(n == 'ln_linear_h2_kernel<%s>' % slabs for n in failing))) else 'exact'
                # csrc/denoiser.hip: the split-f16 attention kernel unless tune misc == 6 (A/B) or its fragments were not packed; clips longer than 192 frames take the fp32 kernel at launch time (not known here)
                d['self_attention'] = 'split' if (attn_packed and (planes or 'self_attn_h2_kernel' not in failing)) else 'exact'
                d['qkv_hands_over_planes'] = planes
            layers.append(d)
        tail = 'split' if (split and w.out_w_h2 and w.in_w_h2 and w.C == 144 and w.tail_h2_ok and not bad('step_tail_h2_kernel')) else 'exact'
        return dict(ffn_math=self.ffn_math, rowblock_math=self.rowblock_math, layers=layers, embedding_and_heads=tail, not_exclusive=failing,
                    all_split=all(d['ffn'] == 'split' and d['rowblock'] == 'split' and d.get('qkv', 'split') == 'split' and d.get('self_attention', 'split') == 'split' for d in layers) and tail == 'split')

    def forward(self, x, timesteps, y=None, out=None, memctx=None, ws=None, batch_rows=None):
        """``memctx`` / ``ws``: caller-owned folded memory and workspace, as in ``forward_step`` (then ``y`` is not consulted).
        ``batch_rows``: B * T of the batch this call is a chain of (default: this call's own)."""
        B, one, Cc, T = x.shape
        self._pick_ffn_tile(batch_rows or B * T, B * T)
        if memctx is None:
            if y is None or 'cond' not in y:
                raise ValueError("model_kwargs['y']['cond'] is required")
            cond = y['cond']
            if self._mem_key != (cond.data_ptr(), cond._version, tuple(cond.shape)):
                self.prepare_memory(cond)
            memctx = self._memctx
        self._bind_memory(memctx, B)
        if one != 1 or Cc != self.w.C:
            raise ValueError('x must be [B,1,%d,T]' % self.w.C)
        x = x.contiguous()
        ts = timesteps.to(torch.int64).contiguous()
        if out is None:
            out = torch.empty_like(x)
        if ws is None:
            ws = self._workspace(B, T)
        _lib.check(self.lib.interdiff_mdm_forward(C.byref(self.w), _lib.dptr(memctx), _lib.dptr(x, torch.float32),
                                                  _lib.dptr(ts, torch.int64), B, T, _lib.dptr(out, torch.float32),
                                                  _lib.dptr(ws), ws.numel(), _lib.stream()), 'mdm_forward')
        return out

    __call__ = forward

    @property
    def supports_forward_step(self):
        return not self.w.layer[0].is_qan             # the step's sampler bookkeeping rides on layer 0's QKV kernel

    @property
    def step_chaining(self):
        """True when ``forward_step`` honours ``embed_ready`` / ``embed_next`` (interdiff_mdm_step_chaining: split arithmetic selected, token width
        144, plane fragments packed); otherwise the flags are ignored (same results, every step runs its own embedding)."""
        self._pick_ffn_tile(1)                    # (writes the arithmetic selection into the handle)
        return bool(self.lib.interdiff_mdm_step_chaining(C.byref(self.w)))

    def forward_step(self, x, timesteps, table, state, gt=None, mask=None, y=None, memctx=None, ws=None, batch_rows=None, embed_ready=False, embed_next=False):
        """``embed_next``: this step's last launch also computes the NEXT plain step's embedding into the workspace (csrc/tail_h2.h); ``embed_ready``: the
        previous call on this x / workspace was made with ``embed_next`` and nothing touched x, the timesteps or the workspace since (same bits either way).
        One plain reverse step with the update applied inside the last GEMM (interdiff_mdm_forward_step): ``x`` [B,1,C,T] and
        the sampler state (``timesteps`` int64 [B], ``state`` int64 [8]) are advanced in place (any T; T % 4 == 0 takes the 16-byte form of the update).  ``memctx`` / ``ws``:
        caller-owned folded memory (``prepare_memory(cond, into=)``) and workspace (``workspace_bytes(B, T)`` bytes) instead of the
        model's -- what lets two chains of one sample run side by side; ``batch_rows`` then names the whole batch's B * T (see
        ``_pick_ffn_tile``)."""
        B, one, Cc, T = x.shape
        self._pick_ffn_tile(batch_rows or B * T, B * T)
        if one != 1 or Cc != self.w.C or not x.is_contiguous():
            raise ValueError('x must be a contiguous [B,1,%d,T]' % self.w.C)
        if memctx is None:
            cond = y['cond']
            if self._mem_key != (cond.data_ptr(), cond._version, tuple(cond.shape)):
                self.prepare_memory(cond)
            memctx = self._memctx
        self._bind_memory(memctx, B)
        if ws is None:
            ws = self._workspace(B, T)
        flags = (_lib.STEP_EMBED_READY if embed_ready else 0) | (_lib.STEP_EMBED_NEXT if embed_next else 0)
        _lib.check(self.lib.interdiff_mdm_forward_step_ex(C.byref(self.w), _lib.dptr(memctx), _lib.dptr(x, torch.float32),
                                                          _lib.dptr(timesteps, torch.int64), B, T, _lib.dptr(gt, allow_none=True),
                                                          _lib.dptr(mask, allow_none=True), _lib.dptr(table), _lib.dptr(state),
                                                          _lib.dptr(ws), ws.numel(), flags, _lib.stream()), 'mdm_forward_step')
        return x


def ffn_parts(model, x2, layer, encoder=False, out=None, batch_rows=None):
    """The fused feed-forward block of one layer on ``model``'s weights: x2 [M,256] -> partial slabs [FFN_SLICES, M, 256] whose
    sum is x2 + linear2(gelu(linear1(x2))) (interdiff_mdm_ffn).  Tile by ``batch_rows`` (default M), see MDM._pick_ffn_tile."""
    lib = _lib.load()
    M = x2.shape[0]
    model._pick_ffn_tile(batch_rows or M, M)
    x2 = x2.contiguous()
    if out is None:
        out = torch.empty(_lib.FFN_SLICES, M, D, dtype=torch.float32, device=x2.device)
    _lib.check(lib.interdiff_mdm_ffn(C.byref(model.w), layer, 1 if encoder else 0, _lib.dptr(x2, torch.float32), M, _lib.dptr(out),
                                     _lib.stream()), 'mdm_ffn')
    return out


def linear(x, weight, bias=None, residual=None, gelu=False, out=None, cfg=0):
    """``epi(x @ weight.T + bias)`` on the denoiser's fp32-MFMA GEMM (x [M,K], weight [N,K]); gelu = erf form."""
    lib = _lib.load()
    M, K = x.shape
    N = weight.shape[0]
    x, weight = x.contiguous(), weight.contiguous()
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=x.device)
    epi = 1 if gelu else (2 if residual is not None else 0)
    if gelu and residual is not None:
        raise ValueError('gelu and residual are separate epilogues')
    _lib.check(lib.interdiff_gemm_f32(_lib.dptr(x, torch.float32), K, _lib.dptr(weight, torch.float32), _lib.dptr(bias, allow_none=True),
                                      _lib.dptr(residual, allow_none=True), _lib.dptr(out), N, M, N, K, epi, cfg, _lib.stream()), 'gemm_f32')
    return out
