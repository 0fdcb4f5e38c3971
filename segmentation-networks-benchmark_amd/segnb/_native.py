"""ctypes binding of libsegnb_hip.so (the C ABI declared in include/segnb_hip.h).

The product path has NO fallback: if the shared library is missing or fails to load, importing a
kernel entry point raises.  (Tests may inject an ABI emulator with ``set_backend_for_testing`` to
check host-side plan logic on CPU; nothing in the package ever does.)

Error convention follows the reference's only native-op precedent, ``_check`` in
lib/modules/abn/functions.py:12-15: a non-zero status becomes ``RuntimeError``.
"""
import ctypes
import os

MAX_TAPS = 64
F32, BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('SEGNB_LIB') or os.path.join(os.path.dirname(_HERE), 'csrc', 'libsegnb_hip.so')

c_int, c_float, c_double, c_void_p, c_ll = (ctypes.c_int, ctypes.c_float, ctypes.c_double, ctypes.c_void_p,
                                            ctypes.c_longlong)


class ConvGeom(ctypes.Structure):
    """segnb_conv_geom"""
    _fields_ = [('N', c_int), ('Hi', c_int), ('Wi', c_int), ('Ci', c_int),
                ('Ho', c_int), ('Wo', c_int), ('Co', c_int),
                ('ld_in', c_int), ('ld_out', c_int),
                ('QH', c_int), ('QW', c_int),
                ('in_step', c_int), ('out_step', c_int),
                ('oh0', c_int), ('ow0', c_int),
                ('ntaps', c_int),
                ('dh', c_int * MAX_TAPS), ('dw', c_int * MAX_TAPS)]


class BnReduceEpilogue(ctypes.Structure):
    """segnb_bn_reduce_epilogue"""
    _fields_ = [('y', c_void_p), ('ld_y', c_int), ('coef', c_void_p), ('sums', c_void_p), ('act', c_int),
                ('slope', c_float)]


class BnApplyEpilogue(ctypes.Structure):
    """segnb_bn_apply_epilogue"""
    _fields_ = [('y', c_void_p), ('ld_y', c_int), ('coef', c_void_p), ('sums', c_void_p), ('gamma', c_void_p), ('C', c_int),
                ('count', ctypes.c_double), ('bcoef', c_void_p), ('dgamma', c_void_p), ('dbeta', c_void_p), ('act', c_int),
                ('slope', c_float), ('dx', c_void_p), ('ld_dx', c_int), ('accumulate', c_int)]


class ActEpilogue(ctypes.Structure):
    """segnb_act_epilogue"""
    _fields_ = [('coef', c_void_p), ('act', c_int), ('slope', c_float)]


class OperandTf(ctypes.Structure):
    """segnb_operand_tf"""
    _fields_ = [('kind', c_int), ('y', c_void_p), ('ld_y', c_int), ('coef', c_void_p), ('bcoef', c_void_p),
                ('drop', c_void_p), ('Cp', c_int), ('act', c_int), ('slope', c_float)]


TF_ACT, TF_BNBWD = 1, 2


class UpcatSrc(ctypes.Structure):
    """segnb_upcat_src"""
    _fields_ = [('u', c_void_p), ('Cu', c_int), ('ld_u', c_int)]


class WgradTarget(ctypes.Structure):
    """segnb_wgrad_target"""
    _fields_ = [('gw', c_void_p), ('s_out', c_ll), ('s_in', c_int), ('ci_off', c_int), ('Ci', c_int), ('Co', c_int),
                ('accumulate', c_int), ('ntaps', c_int), ('kpos', c_int * MAX_TAPS)]


class LossSpec(ctypes.Structure):
    """segnb_loss_spec"""
    _fields_ = [('w_bce', c_float), ('w_focal', c_float), ('w_jaccard', c_float), ('w_sjaccard', c_float),
                ('w_dice', c_float), ('smooth', c_float), ('eps', c_float), ('norm', c_float),
                ('focal_mean', c_int), ('bce_sum', c_int), ('focal_gamma', c_float)]


_P = c_void_p
# name -> argtypes (all return int status)
SIGNATURES = {
    'segnb_conv_fprop': [ctypes.POINTER(ConvGeom), c_int, _P, _P, _P, c_int, _P, _P, _P],
    'segnb_conv_fprop_act': [ctypes.POINTER(ConvGeom), c_int, _P, _P, _P, c_int, _P, ctypes.POINTER(ActEpilogue), _P],
    'segnb_conv_fprop_drop': [ctypes.POINTER(ConvGeom), c_int, _P, _P, _P, c_int, _P, _P, c_int, _P, c_int, _P],
    'segnb_conv_fprop_upcat': [ctypes.POINTER(ConvGeom), c_int, _P, ctypes.POINTER(UpcatSrc), _P, _P, c_int, _P, _P, _P],
    'segnb_conv_fprop_upsum': [ctypes.POINTER(ConvGeom), c_int, _P, _P, _P, ctypes.POINTER(UpcatSrc), _P],
    'segnb_conv_wgrad_upcat': [ctypes.POINTER(ConvGeom), c_int, _P, ctypes.POINTER(UpcatSrc), _P, _P, c_int, _P],
    'segnb_conv_fprop_bnreduce': [ctypes.POINTER(ConvGeom), c_int, _P, _P, _P, ctypes.POINTER(BnReduceEpilogue), _P],
    'segnb_conv_fprop_bnsums': [ctypes.POINTER(ConvGeom), c_int, _P, _P, ctypes.POINTER(BnReduceEpilogue), _P],
    'segnb_conv_fprop_bnapply': [ctypes.POINTER(ConvGeom), c_int, _P, _P, ctypes.POINTER(BnApplyEpilogue), _P],
    'segnb_conv_fprop_tf': [ctypes.POINTER(ConvGeom), c_int, _P, ctypes.POINTER(OperandTf), _P, _P, c_int, _P, _P,
                            ctypes.POINTER(BnReduceEpilogue), _P],
    'segnb_conv_wgrad_tf': [ctypes.POINTER(ConvGeom), c_int, _P, ctypes.POINTER(OperandTf), _P, ctypes.POINTER(OperandTf), _P,
                            c_int, _P],
    'segnb_conv_wgrad': [ctypes.POINTER(ConvGeom), c_int, _P, _P, _P, c_int, _P],
    'segnb_wgrad_target_arm': [ctypes.POINTER(WgradTarget)],
    'segnb_sgd_pack_pair_multi': [_P, c_int, c_int, _P, _P, c_float, _P],
    'segnb_sgd_ranges': [_P, _P, _P, c_int, c_ll, c_float, _P],
    'segnb_upconv_fprop': [c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, c_int, c_int, _P, c_int, _P, c_int, _P, _P],
    'segnb_upconv_fprop_acc': [c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, c_int, c_int, _P, c_int, _P, _P],
    'segnb_upconv_fprop_act': [c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, c_int, c_int, _P, c_int, _P, c_int,
                               ctypes.POINTER(ActEpilogue), _P],
    'segnb_conv_wgrad_bnapply': [ctypes.POINTER(ConvGeom), c_int, _P, _P, c_int, _P, c_int, _P, _P, c_int, c_int, c_float, _P, c_int, _P],
    'segnb_pack_weight': [_P, _P, c_int, c_int, c_int, c_int, c_ll, c_ll, ctypes.POINTER(c_int), _P, _P, _P],
    'segnb_unpack_wgrad': [_P, _P, c_int, c_int, c_int, c_ll, c_ll, ctypes.POINTER(c_int), _P, _P, c_int, _P],
    'segnb_pack_weight_multi': [_P, c_int, c_int, _P],
    'segnb_unpack_wgrad_multi': [_P, c_int, c_int, _P],
    'segnb_pack_input_nchw': [_P, c_int, c_int, c_int, c_int, _P, c_int, c_int, c_int, _P],
    'segnb_pack_input_u8': [_P, c_int, c_int, c_int, c_int, c_float, ctypes.POINTER(c_float), ctypes.POINTER(c_float), _P,
                            c_int, c_int, c_int, _P],
    'segnb_conv_fprop_u8': [ctypes.POINTER(ConvGeom), _P, c_int, c_float, ctypes.POINTER(c_float), ctypes.POINTER(c_float),
                            _P, _P, c_int, _P, _P, _P, c_int, _P],
    'segnb_bn_finalize': [_P, c_int, c_int, c_double, _P, _P, c_float, c_float, _P, _P, _P, c_int, _P, _P],
    'segnb_bn_finalize_keep': [_P, c_int, c_int, c_double, _P, _P, c_float, c_float, _P, _P, _P, _P, _P, _P],
    'segnb_bn_act_fwd': [c_int, _P, c_int, c_int, c_int, c_int, c_int, _P, c_int, c_float, _P, _P, c_int, _P,
                         c_int, _P, c_int, _P, c_int, _P],
    'segnb_bn_act_bwd_reduce': [c_int, _P, c_int, c_int, c_int, c_int, c_int, _P, c_int, c_float, _P, _P, c_int,
                                _P, c_int, _P, c_int, _P, c_int, _P, _P, c_int, _P],
    'segnb_bn_act_bwd_reduce_add': [c_int, _P, c_int, c_int, c_int, c_int, c_int, _P, c_int, c_float, _P, _P, c_int, _P, c_int, _P,
                                    c_int, _P, _P, c_int, _P],
    'segnb_bn_fwd_fused': [c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, c_float, c_float, _P, _P, _P, _P,
                           _P, c_int, c_float, _P, _P, c_int, _P, c_int, _P, c_int, _P, c_int, _P],
    'segnb_bn_bwd_apply_fused': [c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, c_int, _P,
                                 _P, c_int, _P, c_int, _P],
    'segnb_bn_fwd_fused_head': [c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, c_float, c_float, _P, _P, _P, _P,
                                _P, c_int, c_float, _P, _P, c_int, _P, _P, c_int, _P, _P],
    'segnb_bias_grad_multi': [_P, c_int, _P],
    'segnb_pack_weight_elem_multi': [_P, c_int, c_int, _P],
    'segnb_pack_weight_pair_multi': [_P, c_int, c_int, _P],
    'segnb_unpack_wgrad_elem_multi': [_P, c_int, c_int, _P],
    'segnb_bn_stats_ld': [c_int, _P, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P],
    'segnb_bn_act_fwd_stats': [c_int, _P, c_int, c_int, c_int, c_int, c_int, _P, c_int, c_float, _P, _P, c_int, _P, c_int, _P],
    'segnb_bn_fwd_fused_ld': [c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, _P, c_float, c_float, _P, _P, _P,
                              _P, _P, c_int, c_float, _P, _P, c_int, _P],
    'segnb_head_bn_bwd': [c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, c_float, _P, _P, c_int, _P, _P, c_int,
                          _P, _P, _P, _P],
    'segnb_head_bn_bwd_apply': [c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, c_int, _P, c_int, c_float,
                                _P, _P, c_int, _P, _P, c_int, _P],
    'segnb_bn_bwd_apply_fused_acc': [c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, c_int, _P,
                                 _P, c_int, _P, c_int, _P],
    'segnb_bn_bwd_apply_fused_direct': [c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, c_int,
                                        _P, c_int, c_float, _P, c_int, _P, c_int, _P],
    'segnb_bn_bwd_apply_fused_direct_acc': [c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, c_int,
                                        _P, c_int, c_float, _P, c_int, _P, c_int, _P],
    'segnb_bn_bwd_apply_fused_src': [c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, c_int, _P,
                                     c_int, c_float, _P, _P, c_int, _P, c_int, _P, c_int, _P, c_int, _P],
    'segnb_abn_scale': [_P, c_float, _P, c_int, _P],
    'segnb_abn_dscale': [_P, _P, _P, c_int, _P],
    'segnb_tiles_gather': [_P, c_int, c_int, c_int, c_int, c_int, _P, c_int, c_int, c_int, _P, _P],
    'segnb_tiles_gather_u8': [_P, c_int, c_int, c_int, c_int, c_int, _P, c_int, c_int, c_int, c_float,
                              ctypes.POINTER(c_float), ctypes.POINTER(c_float), _P, _P],
    'segnb_tiles_merge': [_P, c_int, c_int, _P, c_int, c_int, c_int, c_int, _P, c_int, c_int, c_int, c_int, _P, _P],
    'segnb_add': [c_int, _P, c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, _P],
    'segnb_bn_stats': [c_int, _P, c_int, c_int, c_int, c_int, c_int, _P, _P],
    'segnb_upsample_bilinear2x_fwd': [c_int, _P, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P],
    'segnb_upsample_bilinear2x_bwd': [c_int, _P, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P],
    'segnb_maxpool_fwd': [c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, _P],
    'segnb_maxpool_bwd': [c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, _P],
    'segnb_maxpool_bwd_add': [c_int, _P, c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, _P],
    'segnb_nhwc_to_nchw_f32': [c_int, _P, c_int, c_int, c_int, c_int, c_int, _P, _P],
    'segnb_bn_bwd_finalize': [_P, c_int, c_int, c_double, _P, _P, _P, _P, _P, c_int, _P],
    'segnb_bn_bwd_finalize_clear': [_P, c_int, c_int, c_double, _P, _P, _P, _P, _P, c_int, _P, _P],
    'segnb_bn_bwd_apply_direct': [c_int, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, c_int, c_float, _P, c_int, _P, c_int,
                                  _P, c_int, _P],
    'segnb_bn_bwd_apply': [c_int, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, c_int, _P, c_int, _P, c_int,
                           _P],
    'segnb_head_fwd': [c_int, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, c_int, _P, _P],
    'segnb_head_bwd': [c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, _P, c_int, _P, _P, _P],
    'segnb_head_conv_fwd': [c_int, _P, c_int, c_int, c_int, c_int, c_int, _P, c_int, c_int, c_int, _P, c_int, _P, _P],
    'segnb_head_conv_bwd': [c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, c_int, c_int, c_int, _P, c_int, c_float,
                            _P, c_int, _P, _P, _P, _P],
    'segnb_seg_loss_reduce': [_P, _P, c_ll, c_float, _P, _P],
    'segnb_seg_loss_reduce_finalize': [_P, _P, c_ll, ctypes.POINTER(LossSpec), _P, _P, _P],
    'segnb_seg_loss_map': [_P, _P, c_ll, c_int, c_float, _P, _P],
    'segnb_seg_loss_map_bwd': [_P, _P, c_ll, c_int, c_float, _P, _P, _P],
    'segnb_absmax_f32': [_P, c_ll, _P, _P],
    'segnb_pr_histogram': [_P, _P, c_ll, _P, c_int, _P, _P],
    'segnb_seg_loss_finalize': [_P, ctypes.POINTER(LossSpec), _P, _P],
    'segnb_seg_loss_bwd': [_P, _P, c_ll, _P, _P, ctypes.POINTER(LossSpec), _P, _P, _P],
    'segnb_tune': [ctypes.c_char_p, c_int],
    'segnb_wg_cu_share': [c_int],
    'segnb_plan_begin': [],
    'segnb_plan_end': [ctypes.POINTER(c_void_p), ctypes.POINTER(c_int)],
    'segnb_plan_run': [_P],
    'segnb_plan_destroy': [_P],
    'segnb_stream_fork': [_P, _P],
    'segnb_stream_fork_arm': [_P],
    'segnb_stream_fork_commit': [_P, _P],
    'segnb_event_record': [_P, _P],
    'segnb_stream_join': [_P, _P],
    'segnb_debug_stamps': [_P],
    'segnb_debug_census': [ctypes.c_char_p, c_int],
    'segnb_sgd_step': [_P, _P, c_ll, c_float, _P],
    'segnb_rmsprop_step': [_P, _P, _P, c_ll, c_float, c_float, c_float, _P],
    'segnb_adam_step': [_P, _P, _P, _P, c_ll, c_float, c_float, c_float, c_float, c_int, _P],
}
PLAIN = {'segnb_version': (c_int, []), 'segnb_pack_pair_job_bytes': (c_int, []), 'segnb_pack_pair_job_blocks': (c_int, [c_int, c_int, c_int, c_int]), 'segnb_pack_elem_job_blocks': (c_int, [c_int, c_int, c_int]), 'segnb_bias_grad_job_bytes': (c_int, []), 'segnb_head_fused_ok': (c_int, [c_int, c_int]), 'segnb_head_conv_ok': (c_int, [c_int, c_int, c_int, c_int]), 'segnb_conv_fprop_tf_ok': (c_int, [ctypes.POINTER(ConvGeom), c_int, c_int]), 'segnb_conv_wgrad_tf_ok': (c_int, [ctypes.POINTER(ConvGeom), c_int]), 'segnb_conv_fprop_bnreduce_ok': (c_int, [ctypes.POINTER(ConvGeom), c_int]), 'segnb_conv_fprop_actmask_ok': (c_int, [ctypes.POINTER(ConvGeom), c_int]), 'segnb_conv_fprop_drop_ok': (c_int, [ctypes.POINTER(ConvGeom), c_int]), 'segnb_conv_fprop_bnapply_ok': (c_int, [ctypes.POINTER(ConvGeom), c_int]), 'segnb_conv_fprop_u8_ok': (c_int, [ctypes.POINTER(ConvGeom), c_int]), 'segnb_conv_fprop_upd_ok': (c_int, [ctypes.POINTER(ConvGeom), c_int]), 'segnb_conv_upcat_ok': (c_int, [ctypes.POINTER(ConvGeom), c_int, c_int]), 'segnb_conv_fprop_upsum_ok': (c_int, [ctypes.POINTER(ConvGeom), c_int, c_int]), 'segnb_conv_wgrad_bnapply_ok': (c_int, [ctypes.POINTER(ConvGeom), c_int]), 'segnb_upconv_fprop_acc_ok': (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]), 'segnb_upconv_fprop_ok': (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int]), 'segnb_conv_wgrad_slabs': (c_int, [ctypes.POINTER(ConvGeom), c_int]), 'segnb_pack_job_bytes': (c_int, []), 'segnb_pack_job_blocks': (c_int, [c_int, c_int, c_int, c_ll, c_ll]), 'segnb_device_cus': (c_int, []), 'segnb_last_error': (ctypes.c_char_p, [])}

_lib = None
_test_backend = None


class NativeLibraryMissing(RuntimeError):
    pass


def load():
    """Load libsegnb_hip.so once.  Raises NativeLibraryMissing -- never falls back."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeLibraryMissing(
            'libsegnb_hip.so not found at %s -- build it with __graft_entry__.build() '
            '(segmentation-networks-benchmark_amd/csrc/build.sh); there is no non-HIP path' % LIB_PATH)
    # torch FIRST: PyTorch-ROCm ships its own libamdhip64.so; if this library were the first to pull a HIP runtime into the process it
    # would bind /opt/rocm's copy, torch would then load its own, and every launch made here fails with "no ROCm-capable device is
    # detected" (two runtimes, the device context in the other one) -- seen when __graft_entry__.build() dlopen'ed the library before
    # anything imported torch.  With torch's runtime already loaded the dynamic linker resolves this library's HIP symbols to it.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)         # AttributeError if the ABI and the header drift apart
        fn.argtypes = argtypes
        fn.restype = c_int
    for name, (res, argtypes) in PLAIN.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = res
    _lib = lib
    return lib


def set_backend_for_testing(backend):
    """tests only: route ABI calls to an emulator object exposing the same function names.  Refused unless the
    process was started by the test harness (tests/conftest.py exports SEGNB_TEST_HARNESS=1): nothing in the product
    can switch the HIP library off, by accident or otherwise."""
    global _test_backend
    if backend is not None and os.environ.get('SEGNB_TEST_HARNESS') != '1':
        raise RuntimeError('set_backend_for_testing is test infrastructure (SEGNB_TEST_HARNESS=1 not set); the product '
                           'path runs on libsegnb_hip.so only')
    _test_backend = backend


def call(name, *args):
    """Invoke one ABI entry point; non-zero status -> RuntimeError (functions.py:12-15 convention)."""
    if _test_backend is not None:
        rc = getattr(_test_backend, name)(*args)
        if rc:
            raise RuntimeError('%s failed (emulator) rc=%s' % (name, rc))
        return
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.segnb_last_error()
        raise RuntimeError('HIP error encountered in %s (status %d): %s'
                           % (name, rc, msg.decode() if msg else ''))


def has_test_backend():
    return _test_backend is not None


def plan_record_begin():
    call('segnb_plan_begin')


def plan_record_end():
    """-> (opaque plan handle or None when the recorded sequence is not replayable, number of recorded launches)"""
    h, n = c_void_p(), c_int()
    call('segnb_plan_end', ctypes.byref(h), ctypes.byref(n))
    return (h.value or None), n.value


def plan_record_abort():
    """An exception escaped a region that was being recorded: close the recording on this thread and free what it holds
    (otherwise every later ABI call would be appended to the abandoned list and segnb_plan_run would refuse to run)."""
    try:
        handle, _ = plan_record_end()
    except Exception:
        return
    if handle is not None:
        try:
            call('segnb_plan_destroy', handle)
        except Exception:
            pass


def census_read():
    """{entry point: executions on this thread since the last read} while segnb_tune('call_census', 1) is on (cleared by the
    read); {} on the CPU emulator, which has no launch lists."""
    if _test_backend is not None:
        return {}
    buf = ctypes.create_string_buffer(1 << 16)
    call('segnb_debug_census', buf, len(buf))
    out = {}
    for line in buf.value.decode().splitlines():
        k, _, v = line.rpartition(' ')
        if not k or not v.isdigit():
            continue                       # (a last line cut off by the fixed buffer: skipped, the census still compares)
        out[k] = int(v)
    return out


def query(name, *args):
    """Plain int-returning ABI query (no status convention)."""
    if _test_backend is not None:
        return getattr(_test_backend, name)(*args)
    return getattr(load(), name)(*args)


def ptr(t, offset_elems=0):
    """Device pointer of a torch tensor (+ element offset); None -> NULL."""
    if t is None:
        return None
    return t.data_ptr() + offset_elems * t.element_size()


def float_array(values):
    return (c_float * len(values))(*[float(v) for v in values])


def int_array(values):
    return (c_int * len(values))(*[int(v) for v in values])
