"""ctypes binding of the C-ABI in include/interdiff_hip.h.

The HIP library is the product: if it is missing this module raises -- there is no CPU or
torch fallback anywhere in interdiff_amd.  torch is used for device memory and streams only.
"""
import ctypes as C
import os
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('INTERDIFF_HIP_LIB') or os.path.join(_HERE, 'csrc', 'libinterdiff_hip.so')      # (the override: A/B builds of the SAME library under build_ab/, tools/ only)
ABI_VERSION = 16

vp, i32, i64, f32, u64, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_uint64, C.c_size_t

MDM_LAYERS = 8
FFN_SLICES = 5              # IDF_FFN_SLICES: partial output slabs of the fused feed-forward kernel
STEP_EMBED_READY, STEP_EMBED_NEXT = 1, 2          # flags of interdiff_mdm_forward_step_ex (IDF_STEP_*)
TUNE = dict(embed=0, qkv=1, outproj=2, ffn=3, ffn_math=4, heads=5, contact=6, misc=7)      # indices into MdmWeights.tune (IDF_TUNE_*)


class SmplModel(C.Structure):
    _fields_ = [('V', i32), ('J', i32), ('n_betas', i32), ('KB', i32), ('S', i32),
                ('blend', vp), ('jt', vp), ('js', vp), ('parents', vp), ('skin_idx', vp), ('skin_w', vp)]


class MdmLayer(C.Structure):
    _fields_ = [('is_qan', i64),
                ('sa_in_w', i64), ('sa_in_b', i64), ('sa_out_w', i64), ('sa_out_b', i64),
                ('qc', i64), ('wk', i64),
                ('ca_q_w', i64), ('ca_q_b', i64), ('ca_kv_w', i64), ('ca_kv_b', i64),
                ('ca_out_w', i64), ('ca_out_b', i64),
                ('ff1_w', i64), ('ff1_b', i64), ('ff2_w', i64), ('ff2_b', i64), ('ffn_pack', i64), ('ffn_b1p', i64), ('sa_in_pack', i64), ('sa_out_frag', i64),
                ('ln_w', i64 * 3), ('ln_b', i64 * 3), ('ffn_pack_h2', i64), ('sa_in_pack_h2', i64), ('qc_h2', i64), ('rb_h2_ok', i64), ('sa_out_frag_h2', i64), ('qkv_bounds_ok', i64)]


class MdmWeights(C.Structure):
    _fields_ = [('C', i32), ('n_steps', i32), ('arena', vp),
                ('in_w', i64), ('in_b', i64), ('out_w', i64), ('out_b', i64),
                ('temb_table', i64), ('pe', i64), ('max_T', i32), ('has_encoder', i32),
                ('layer', MdmLayer * MDM_LAYERS), ('enc_layer', MdmLayer * MDM_LAYERS), ('tune', i32 * 8), ('out_w_h2', i64), ('in_w_h2', i64), ('tail_h2_ok', i64), ('mem_len', i32), ('rb_tokens', i32)]


class PnMlp(C.Structure):
    _fields_ = [('w', i64 * 3), ('b', i64 * 3), ('c', i32 * 4)]


class PointNet2(C.Structure):
    _fields_ = [('arena', vp), ('sa1', PnMlp * 2), ('sa2', PnMlp * 2), ('lin_w', i64), ('lin_b', i64)]


class ObjProj(C.Structure):
    _fields_ = [('T', i32), ('past_len', i32), ('P', i32), ('n_pre', i32), ('arena', vp),
                ('dct_pad', i64), ('dct', i64), ('idct', i64), ('hand_bonus', i64),
                ('layer', i64 * 12), ('cin', i32 * 12), ('cout', i32 * 12)]


class CorrectionCtx(C.Structure):
    _fields_ = [('smpl', C.POINTER(SmplModel)), ('objproj', C.POINTER(ObjProj)),
                ('faces', vp), ('adj_ptr', vp), ('adj_face', vp), ('adj_corner', vp), ('markers_idx', vp),
                ('n_markers', i32), ('n_points', i32), ('past_len', i32), ('tune', i32),
                ('vorder', vp), ('faces_scan', vp), ('markers_scan', vp), ('adj_pair_scan', vp), ('adj_pair', vp), ('vrank', vp)]


class OptCtx(C.Structure):
    _fields_ = [('geo', C.POINTER(CorrectionCtx)), ('blendT', vp), ('jv_ptr', vp), ('jv_vtx', vp), ('jv_w', vp), ('K3P', i32), ('_pad', i32)]


OPT_NP, OPT_NLOSS = 483, 6
_OPT_PTRS = ('betas', 'obj_points', 'param', 'init', 'grad', 'm', 'v', 'best', 'pose', 'tr', 'verts', 'vposed', 'verts_gt', 'gv',
             'jtr', 'pts', 'y2x', 'y2x_signed', 'yidx', 'near', 'dvposed', 'dA', 'dfeat', 'gtr', 'lossf', 'loss', 'loss_hist',
             'best_loss', 'flag', 'foot_static', 'foot_cnt', 'ctl', 'smpl_ws')


class OptState(C.Structure):
    _fields_ = [('B', i32), ('T', i32), ('P', i32), ('max_iters', i32)] + [(k, vp) for k in _OPT_PTRS] + [('smpl_ws_bytes', sz), ('porder', vp), ('psort', vp), ('pbox', vp)]


_SIGS = {
    'interdiff_abi_version': (C.c_int, []),
    'interdiff_build_info': (C.c_char_p, []),
    'interdiff_rotation_6d_to_matrix': (C.c_int, [vp, vp, i64, vp]),
    'interdiff_matrix_to_rotation_6d': (C.c_int, [vp, vp, i64, vp]),
    'interdiff_matrix_to_axis_angle': (C.c_int, [vp, vp, i64, vp]),
    'interdiff_axis_angle_to_matrix': (C.c_int, [vp, vp, i64, vp]),
    'interdiff_axis_angle_to_quaternion': (C.c_int, [vp, vp, i64, vp]),
    'interdiff_rotation_6d_to_axis_angle': (C.c_int, [vp, vp, i64, vp]),
    'interdiff_smpl_workspace_bytes': (sz, [C.POINTER(SmplModel), i64]),
    'interdiff_smpl_forward': (C.c_int, [C.POINTER(SmplModel), vp, vp, vp, i64, vp, vp, vp, vp, sz, vp]),
    'interdiff_vertex_normals': (C.c_int, [vp, i64, i32, vp, vp, vp, vp, vp, vp]),
    'interdiff_nn_argmin': (C.c_int, [vp, i32, vp, i32, i64, vp, vp]),
    'interdiff_point2point_signed': (C.c_int, [vp, i32, vp, i32, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    'interdiff_gemm_f32': (C.c_int, [vp, i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    'interdiff_pointnet2_encode': (C.c_int, [C.POINTER(PointNet2), vp, i32, i32, vp, vp]),
    'interdiff_mdm_encode_workspace_bytes': (sz, [i32, i32]),
    'interdiff_mdm_encode': (C.c_int, [C.POINTER(MdmWeights), vp, vp, i32, i32, vp, vp, sz, vp]),
    'interdiff_mdm_ffn': (C.c_int, [C.POINTER(MdmWeights), i32, i32, vp, i32, vp, vp]),
    'interdiff_mdm_memctx_floats': (sz, [i32]),
    'interdiff_mdm_memctx_floats_for': (sz, [i32, i32]),
    'interdiff_mdm_workspace_bytes': (sz, [i32, i32]),
    'interdiff_mdm_prepare_memory': (C.c_int, [C.POINTER(MdmWeights), vp, i32, vp, vp, sz, vp]),
    'interdiff_mdm_forward': (C.c_int, [C.POINTER(MdmWeights), vp, vp, vp, i32, i32, vp, vp, sz, vp]),
    'interdiff_mdm_forward_step': (C.c_int, [C.POINTER(MdmWeights), vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, sz, vp]),
    'interdiff_mdm_forward_step_ex': (C.c_int, [C.POINTER(MdmWeights), vp, vp, vp, i32, i32, vp, vp, vp, vp, vp, sz, i32, vp]),
    'interdiff_mdm_step_chaining': (C.c_int, [C.POINTER(MdmWeights)]),
    'interdiff_inpaint': (C.c_int, [vp, vp, vp, i64, vp]),
    'interdiff_posterior_step': (C.c_int, [vp, vp, vp, i64, f32, f32, f32, u64, u64, vp]),
    'interdiff_randn': (C.c_int, [vp, i64, u64, u64, vp]),
    'interdiff_posterior_step_at': (C.c_int, [vp, vp, vp, i64, f32, f32, f32, u64, u64, u64, vp]),
    'interdiff_randn_at': (C.c_int, [vp, i64, u64, u64, u64, vp]),
    'interdiff_posterior_step_dev': (C.c_int, [vp, vp, vp, vp, i64, vp, vp, vp, i32, vp]),
    'interdiff_sampler_advance': (C.c_int, [vp, vp, i32, vp]),
    'interdiff_objprojector_sample': (C.c_int, [C.POINTER(ObjProj), vp, vp, vp, vp, i32, vp, vp]),
    'interdiff_correction_workspace_bytes': (sz, [C.POINTER(CorrectionCtx), i32, i32]),
    'interdiff_correction': (C.c_int, [C.POINTER(CorrectionCtx), vp, vp, vp, vp, vp, i32, i32, f32,
                                       vp, vp, vp, vp, vp, sz, vp]),
    'interdiff_correction_dev': (C.c_int, [C.POINTER(CorrectionCtx), vp, vp, vp, vp, vp, i32, i32, vp, vp, vp, sz, vp]),
    'interdiff_metrics_workspace_bytes': (sz, [C.POINTER(CorrectionCtx), i32, i32]),
    'interdiff_metrics': (C.c_int, [C.POINTER(CorrectionCtx), vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32,
                                    vp, vp, sz, vp]),
    'interdiff_contact_nn_workspace_bytes': (sz, [C.POINTER(CorrectionCtx), i32, i32]),
    'interdiff_contact_nn': (C.c_int, [C.POINTER(CorrectionCtx), vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, sz, vp]),
    'interdiff_optimize_init': (C.c_int, [C.POINTER(OptCtx), C.POINTER(OptState), vp, vp, vp, vp, i32, vp]),
    'interdiff_optimize_loss_grad': (C.c_int, [C.POINTER(OptCtx), C.POINTER(OptState), vp]),
    'interdiff_optimize_step': (C.c_int, [C.POINTER(OptCtx), C.POINTER(OptState), vp]),
    'interdiff_optimize_finish': (C.c_int, [C.POINTER(OptCtx), C.POINTER(OptState), vp, vp, vp, vp, vp]),
    'interdiff_debug_joint_map_vjp': (C.c_int, [vp, vp, vp, i32]),
    'interdiff_debug_lds_sentinel': (C.c_int, [vp, i32, i32, vp]),
    'interdiff_exclusive_cu_report': (C.c_int, [C.c_char_p, i32]),
    'interdiff_debug_deny_exclusive': (C.c_int, [C.c_char_p]),
    'interdiff_debug_f16_aggressor': (C.c_int, [vp, sz, vp, i32, i32, i32, vp]),
    'interdiff_profile_begin': (C.c_int, [i32]),
    'interdiff_profile_end': (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
}

KERNEL_KINDS = ('embed', 'gemm_qkv', 'self_attn', 'gemm_outproj', 'rowblock_qan', 'rowblock_std', 'ffn_fused', 'reserved7',
                'gemm_heads', 'mem_prep', 'inpaint', 'posterior', 'corr_prepare', 'smpl_pose', 'smpl_blend_skin',
                'corr_contact', 'corr_reduce', 'objproj', 'corr_blend', 'other')

_lib = None


class HipLibraryMissing(RuntimeError):
    pass


def load():
    """Load libinterdiff_hip.so (built by ``python -m interdiff_amd.csrc.build`` or
    ``__graft_entry__.build()``).  Raises HipLibraryMissing -- never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipLibraryMissing('%s not found: build it with `python -m interdiff_amd.csrc.build` '
                                '(there is no CPU/torch fallback for the hot path)' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)             # AttributeError if the .so is stale
        fn.restype, fn.argtypes = res, args
    if lib.interdiff_abi_version() != ABI_VERSION:
        raise HipLibraryMissing('stale %s: abi %d != %d, rebuild' % (LIB_PATH, lib.interdiff_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def exclusive_cu_report():
    """(text table, number of f16-MFMA kernels that do NOT get their CU to themselves on the current device) -- include/interdiff_hip.h
    interdiff_exclusive_cu_report; kernels that fail run as their fp32-MFMA counterparts."""
    buf = C.create_string_buffer(8192)
    bad = load().interdiff_exclusive_cu_report(buf, len(buf))
    if bad < 0:
        check(bad, 'exclusive_cu_report')
    return buf.value.decode(), bad


def debug_deny_exclusive(patterns=''):
    """DEBUG (tests): treat every f16-MFMA kernel whose report name contains one of the comma-separated ``patterns`` as not owning its CU from now on (its launcher
    then takes the fp32 kernel); '' clears the list.  Process-wide -- callers drop their captured graphs (they bake the kernel choice in)."""
    check(load().interdiff_debug_deny_exclusive(patterns.encode() if patterns else None), 'debug_deny_exclusive')


def exported_symbols():
    return sorted(_SIGS)


_ERR = {-22: 'IDF_E_INVAL (bad shape / pointer / unsupported size)', -12: 'IDF_E_NOMEM (workspace too small)',
        -5: 'IDF_E_LAUNCH (kernel launch failed)'}


def check(rc, what=''):
    if rc != 0:
        raise RuntimeError('interdiff_hip %s failed: %s' % (what, _ERR.get(rc, rc)))


def dptr(t, dtype=None, allow_none=False):
    """Device pointer of a contiguous CUDA(HIP) tensor."""
    if t is None:
        if allow_none:
            return vp(None)
        raise ValueError('null tensor')
    if not t.is_cuda:
        raise ValueError('interdiff_amd ops need device tensors (got %s); the HIP path has no CPU fallback' % t.device)
    if not t.is_contiguous():
        raise ValueError('tensor must be contiguous')
    if dtype is not None and t.dtype != dtype:
        raise ValueError('expected %s, got %s' % (dtype, t.dtype))
    return vp(t.data_ptr())


def stream():
    return vp(torch.cuda.current_stream().cuda_stream)
