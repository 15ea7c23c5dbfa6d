"""Restatement of ``local_attention.LocalAttention`` as InterDiff instantiates it.

The package is NOT under /root/reference and is UNPINNED there
(requirements.txt:28) -> parity for this file is unpinned; this restatement
defines the contract, with the one version-dependent
behaviour (rotary position embedding on q/k) behind the ``rotary`` switch.

Reference ctor call (model/sublayers.py:251-260):
    LocalAttention(dim=d_model, window_size=1, causal=False, look_backward=1,
                   look_forward=1, dropout=p, exact_windowsize=False, autopad=True)
Reference forward call (sublayers.py:350): self_attn(q, k, v, mask=ones(1,T)).
"""
import torch

ROTARY_DEFAULT = True  # local-attention >= 1.5: use_rotary_pos_emb=True and dim given


def _look_around(x, backward, forward, pad_value):
    """x: [b, windows, ...]; concat windows (i-backward .. i+forward) on dim 2."""
    w = x.shape[1]
    pad_shape = list(x.shape)
    pad_shape[1] = backward
    front = torch.full(pad_shape, pad_value, dtype=x.dtype)
    pad_shape[1] = forward
    back = torch.full(pad_shape, pad_value, dtype=x.dtype)
    padded = torch.cat([front, x, back], dim=1)
    pieces = [padded[:, s:s + w] for s in range(backward + forward + 1)]
    return torch.cat(pieces, dim=2)


def _rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], dim=-1)


def rotary_freqs(n_pos, dim, dtype=torch.float32):
    inv = 1.0 / (10000.0 ** (torch.arange(0, dim, 2, dtype=torch.float32) / dim))
    f = torch.outer(torch.arange(n_pos, dtype=torch.float32), inv)
    return torch.cat([f, f], dim=-1).to(dtype)


def local_attention(q, k, v, window_size=1, look_backward=1, look_forward=1,
                    rotary=ROTARY_DEFAULT):
    """q,k,v: [b, n, d] -> [b, n, d].  Non-causal, autopad, all-ones user mask."""
    b, n0, d = q.shape
    pad = (-n0) % window_size
    if pad:
        z = torch.zeros(b, pad, d, dtype=q.dtype)
        q, k, v = (torch.cat([t, z], dim=1) for t in (q, k, v))
    n = q.shape[1]
    w = n // window_size
    scale = d ** -0.5
    bq = q.reshape(b, w, window_size, d) * scale
    bk = _look_around(k.reshape(b, w, window_size, d), look_backward, look_forward, -1.0)
    bv = _look_around(v.reshape(b, w, window_size, d), look_backward, look_forward, -1.0)
    if rotary:
        fr = rotary_freqs(bk.shape[-2], d, q.dtype)        # one row per key slot
        qf = fr[-bq.shape[-2]:]                            # queries take the LAST rows
        bq = bq * qf.cos() + _rotate_half(bq) * qf.sin()
        bk = bk * fr.cos() + _rotate_half(bk) * fr.sin()
    pos = torch.arange(n, dtype=torch.float32).reshape(1, w, window_size)
    kpos = _look_around(pos, look_backward, look_forward, -1.0)   # [1, w, keys]
    valid = (kpos != -1.0) & (kpos < n0)                   # window pad + autopad
    sim = torch.einsum('bwie,bwje->bwij', bq, bk)
    sim = sim.masked_fill(~valid[:, :, None, :], -torch.finfo(sim.dtype).max)
    attn = torch.softmax(sim, dim=-1)
    out = torch.einsum('bwij,bwje->bwie', attn, bv).reshape(b, n, d)
    return out[:, :n0]


class LocalAttention(torch.nn.Module):
    """Shim class with the ctor/forward surface the reference uses (so the
    reference's own sublayers.py can be imported for golden generation)."""

    def __init__(self, dim=None, window_size=1, causal=False, look_backward=1,
                 look_forward=1, dropout=0.0, exact_windowsize=False, autopad=True,
                 **_unused):
        super().__init__()
        assert not causal and not exact_windowsize and autopad
        self.window_size, self.look_backward, self.look_forward = window_size, look_backward, look_forward
        self.rotary = ROTARY_DEFAULT and dim is not None
        self.dropout = torch.nn.Dropout(dropout)

    def forward(self, q, k, v, mask=None, **_unused):
        assert not self.training, "oracle shim is eval-only"
        return local_attention(q, k, v, self.window_size, self.look_backward,
                               self.look_forward, self.rotary)
