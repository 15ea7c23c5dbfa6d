"""MDM denoiser forward (rows A1-A4), restated functionally from a state_dict
whose keys are those of the reference ``MDM`` module.

Follows model/diffusion_smpl.py:226-246 (_decode/forward), :73-120 (decoder =
[std, QaN x6, std] layers, 4 heads, ff 1024, gelu, post-norm),
model/layers.py:9-43 (PositionalEncoding / TimestepEmbedder),
model/sublayers.py:206-375 (TransformerDecoderLayerQaN), torch's
nn.TransformerDecoderLayer / nn.MultiheadAttention (post-norm, eval), and the
LocalAttention restatement in local_attn.py.
"""
import math
import numpy as np
import torch
import torch.nn.functional as F
from .local_attn import local_attention, ROTARY_DEFAULT

N_LAYERS = 8
QAN_LAYERS = (1, 2, 3, 4, 5, 6)
N_HEADS = 4


def positional_table(max_len=5000, d=256):
    """layers.py:14-19."""
    pe = torch.zeros(max_len, d)
    pos = torch.arange(0, max_len, dtype=torch.float).unsqueeze(1)
    div = torch.exp(torch.arange(0, d, 2).float() * (-np.log(10000.0) / d))
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe


def _lin(x, sd, name):
    return x @ sd[name + '.weight'].T + sd[name + '.bias']


def _ln(x, sd, name, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), sd[name + '.weight'], sd[name + '.bias'], eps)


def _mha(xq, xkv, sd, name, heads=N_HEADS):
    """nn.MultiheadAttention (batch_first=False, no masks, eval).  xq [L,B,D], xkv [S,B,D]."""
    W, bvec = sd[name + '.in_proj_weight'], sd[name + '.in_proj_bias']
    D = xq.shape[-1]
    q = xq @ W[:D].T + bvec[:D]
    k = xkv @ W[D:2 * D].T + bvec[D:2 * D]
    v = xkv @ W[2 * D:].T + bvec[2 * D:]
    L, B, _ = q.shape
    S = k.shape[0]
    hd = D // heads
    q = q.reshape(L, B, heads, hd).permute(1, 2, 0, 3)
    k = k.reshape(S, B, heads, hd).permute(1, 2, 0, 3)
    v = v.reshape(S, B, heads, hd).permute(1, 2, 0, 3)
    a = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(hd), dim=-1)
    o = (a @ v).permute(2, 0, 1, 3).reshape(L, B, D)
    return _lin(o, sd, name + '.out_proj')


def _ffn(x, sd, p):
    return _lin(F.gelu(_lin(x, sd, p + '.linear1')), sd, p + '.linear2')


def std_layer(x, mem, sd, p):
    """torch.nn.TransformerDecoderLayer, norm_first=False."""
    x = _ln(x + _mha(x, x, sd, p + '.self_attn'), sd, p + '.norm1')
    x = _ln(x + _mha(x, mem, sd, p + '.multihead_attn'), sd, p + '.norm2')
    x = _ln(x + _ffn(x, sd, p), sd, p + '.norm3')
    return x


def qan_queries(sd, p, heads=N_HEADS):
    """sublayers.py:18-35,295-304: per-head unit-norm (+1e-6) then /sqrt(head_dim)."""
    q = sd[p + '.queries']
    n, D = q.shape
    qh = q.reshape(n, heads, D // heads)
    qh = qh / (torch.sqrt((qh * qh).sum(-1, keepdim=True)) + 1e-6)
    qh = qh / math.sqrt(D // heads)
    return qh.reshape(n, D)


def qan_block(x, sd, p, rotary=ROTARY_DEFAULT):
    """sublayers.py:343-352.  x [T,B,D] -> [T,B,D]."""
    T, B, D = x.shape
    q = qan_queries(sd, p)                                  # [Nq, D]
    nq = q.shape[0]
    qq = q[None, :, None, :].expand(B, nq, T, D).reshape(B * nq, T, D)
    xx = x.permute(1, 0, 2)[:, None].expand(B, nq, T, D).reshape(B * nq, T, D)
    o = local_attention(qq, xx, xx, rotary=rotary).reshape(B, nq, T, D)
    o = torch.einsum('bntd,nk->bktd', o, sd[p + '.wk']).squeeze(1)
    return o.permute(1, 0, 2)


def qan_layer(tgt, mem, sd, p, rotary=ROTARY_DEFAULT):
    """sublayers.py:311-341 (norm_first=False; stochastic_depth p=0 -> identity
    but the tgt + (x - tgt) round trip is kept, :338-339)."""
    x = _ln(tgt + qan_block(tgt, sd, p, rotary), sd, p + '.norm1')
    x = _ln(x + _mha(x, mem, sd, p + '.multihead_attn'), sd, p + '.norm2')
    x = _ln(x + _ffn(x, sd, p), sd, p + '.norm3')
    return tgt + (x - tgt)


def time_embedding(sd, ts, pe):
    """layers.py:42-43: time_embed(pe[ts]) -> [1,B,D]."""
    h = pe[ts]                                              # [B, D]
    h = _lin(F.silu(_lin(h, sd, 'embedTimeStep.time_embed.0')), sd, 'embedTimeStep.time_embed.2')
    return h[None]


def mdm_forward(sd, x, ts, cond, rotary=ROTARY_DEFAULT, n_body=135):
    """x [B,1,C,T], ts int64 [B], cond [M,B,D]  ->  [B,1,C,T]."""
    pe = sd['PositionalEmbedding.pe'][:, 0] if 'PositionalEmbedding.pe' in sd else positional_table()
    pe = pe.to(x.dtype)
    temb = time_embedding(sd, ts, pe)
    xt = x.squeeze(1).permute(2, 0, 1)                      # [T,B,C]
    T = xt.shape[0]
    h = _lin(xt[..., :n_body], sd, 'bodyEmbedding') + _lin(xt[..., n_body:], sd, 'objEmbedding') + temb
    h = h + pe[:T, None]
    for l in range(N_LAYERS):
        p = 'decoder.layers.%d' % l
        h = qan_layer(h, cond, sd, p, rotary) if l in QAN_LAYERS else std_layer(h, cond, sd, p)
    out = torch.cat([_lin(h, sd, 'bodyFinalLinear'), _lin(h, sd, 'objFinalLinear')], dim=-1)
    return out.permute(1, 2, 0).unsqueeze(1).contiguous()


# ---------------------------------------------------------------------------------------------------------------
# Encoder / conditioning ("next" row N1): MDM._get_embeddings (model/diffusion_smpl.py:195-223) with the 8-layer
# encoder [std, QaN x6, std] (:19-70; nn.TransformerEncoderLayer post-norm gelu, sublayers.py:36-205).
# ---------------------------------------------------------------------------------------------------------------
def enc_std_layer(x, sd, p):
    """torch.nn.TransformerEncoderLayer, norm_first=False."""
    x = _ln(x + _mha(x, x, sd, p + '.self_attn'), sd, p + '.norm1')
    return _ln(x + _ffn(x, sd, p), sd, p + '.norm2')


def enc_qan_layer(src, sd, p, rotary=ROTARY_DEFAULT):
    """sublayers.py:140-161 (norm_first=False; stochastic depth identity, round trip kept)."""
    x = _ln(src + qan_block(src, sd, p, rotary), sd, p + '.norm1')
    x = _ln(x + _ffn(x, sd, p), sd, p + '.norm2')
    return src + (x - src)


def get_embeddings(sd, body_pose, body_trans, obj_angles, obj_trans, obj_points, past_len, rotary=ROTARY_DEFAULT):
    """body_pose [T,B,66] axis-angle, body_trans [T,B,3], obj_angles [T,B,3] axis-angle, obj_trans [T,B,3],
    obj_points [B,P,3]  ->  (cond [past_len,B,256], gt [T,B,144])."""
    from . import rotations as R
    from .pointnet2 import pointnet2_encode
    T, B, _ = body_pose.shape
    pc = pointnet2_encode(sd, obj_points)[None]                                       # [1,B,256]
    body6 = R.matrix_to_rotation_6d(R.axis_angle_to_matrix(body_pose.reshape(T, B, -1, 3))).reshape(T, B, -1)
    obj6 = R.matrix_to_rotation_6d(R.axis_angle_to_matrix(obj_angles.reshape(T, B, -1, 3))).reshape(T, B, -1)
    gt = torch.cat([body6, body_trans, obj6, obj_trans], dim=2)
    body, obj = torch.cat([body6, body_trans], dim=2), torch.cat([obj6, obj_trans], dim=2)
    h = _lin(body[:past_len], sd, 'bodyEmbedding') + _lin(obj[:past_len], sd, 'objEmbedding') + pc
    pe = sd['PositionalEmbedding.pe'][:, 0] if 'PositionalEmbedding.pe' in sd else positional_table()
    h = h + pe[:past_len, None].to(h.dtype)
    for l in range(N_LAYERS):
        p = 'encoder.layers.%d' % l
        h = enc_qan_layer(h, sd, p, rotary) if l in QAN_LAYERS else enc_std_layer(h, sd, p)
    return h, gt
