"""ctypes binding of libbidate_hip.so (C ABI: include/bidate_hip.h).

The library is the product: there is no CPU or eager-PyTorch fallback.  If the
shared object is missing or a call fails, a RuntimeError is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('BIDATE_LIB') or os.path.join(_HERE, 'csrc', 'libbidate_hip.so')

BDN_F32, BDN_BF16, BDN_BF16X3, BDN_BF16X2 = 0, 1, 2, 3
IN_PLAIN, IN_BNRELU = 0, 1
WG_SIMPLE, WG_ROLE = 1, 5


def wg_flags(phases=3, kernel=0, blocks=0):
    """BDN_WG_FLAGS of include/bidate_hip.h."""
    return phases | (kernel << 8) | (blocks << 16)

_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t

# name -> (restype, argtypes); mirrors include/bidate_hip.h one to one
SIGNATURES = {
    'bdn_last_error': (C.c_char_p, []),
    'bdn_version': (_i, []),
    'bdn_stream_create': (_i, [_i, C.POINTER(C.c_void_p)]),
    'bdn_stream_destroy': (_i, [_vp]),
    'bdn_event_create': (_i, [C.POINTER(C.c_void_p)]),
    'bdn_event_destroy': (_i, [_vp]),
    'bdn_event_record': (_i, [_vp, _vp]),
    'bdn_stream_wait_event': (_i, [_vp, _vp]),
    'bdn_pack_input': (_i, [_i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'bdn_pack_weights': (_i, [_i, _vp, _vp, _vp, _i, _i, _i, _vp]),
    'bdn_pack_weights_multi': (_i, [_i, _vp, _i, _vp]),
    'bdn_conv3x3': (_i, [_i, _vp, _i, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'bdn_conv3x3_x3src': (_i, [_i, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'bdn_conv3x3_num_mtiles': (_i, [_i, _i, _i, _i, _i]),
    'bdn_conv3x3_num_mtiles_ex': (_i, [_i, _i, _i, _i, _i, _i, _i]),
    'bdn_wgrad_workspace_bytes': (_sz, [_i, _i, _i, _i, _i, _i]),
    'bdn_conv3x3_wgrad': (_i, [_i, _vp, _i, _vp, _i, _vp, _i, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp]),
    'bdn_conv3x3_wgrad_ex': (_i, [_i, _vp, _i, _vp, _i, _vp, _i, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'bdn_conv3x3_wgrad_variant': (_i, [_i, _i, _i, _i, _i, _i, _i, _i, _i, _i]),
    'bdn_wgrad_workspace_bytes_ex': (_sz, [_i, _i, _i, _i, _i, _i, _i, _i, _i, _i]),
    'bdn_conv3x3_wgrad_bnbwd_supported': (_i, [_i, _i, _i, _i, _i, _i, _i]),
    'bdn_conv3x3_wgrad_bnbwd': (_i, [_i, _vp, _i, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp]),
    'bdn_bn_finalize_workspace_bytes': (_sz, [_i, _i, _i]),
    'bdn_bn_finalize': (_i, [_vp, _i, _i, _i, _i, _vp, _vp, _f, _f, _vp, _vp, _vp, _vp, _vp, _vp]),
    'bdn_bn_eval': (_i, [_vp, _vp, _vp, _vp, _f, _i, _i, _vp, _vp]),
    'bdn_bn_eval_fold_multi': (_i, [_vp, _i, _i, _f, _vp]),
    'bdn_conv3x3_eval': (_i, [_i, _vp, _i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'bdn_conv3x3_eval_pair': (_i, [_i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'bdn_conv3x3_eval_cls': (_i, [_i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'bdn_bn_bwd_workspace_bytes': (_sz, [_i, _i, _i, _i, _i, _i]),
    'bdn_bn_bwd': (_i, [_i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    'bdn_conv3d_num_mtiles': (_i, [_i, _i, _i, _i]),
    'bdn_conv3d': (_i, [_i, _vp, _i, _i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'bdn_conv3d_wgrad': (_i, [_i, _vp, _i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'bdn_split_pack': (_i, [_vp, _i, _vp, _i, _i, _vp, _i, _vp, _i, _i, _i, _vp]),
    'bdn_bnrelu': (_i, [_i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp]),
    'bdn_bnrelu_pool': (_i, [_i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp]),
    'bdn_fuse_product': (_i, [_i, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'bdn_product_pool': (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'bdn_product_pool_dates': (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'bdn_upsample2x': (_i, [_i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'bdn_product_pool_split': (_i, [_vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _vp]),
    'bdn_upsample2x_split': (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp]),
    'bdn_upsample2x_bwd': (_i, [_i, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'bdn_upsample2x_bwd_rows': (_i, [_i, _i, _i, _i, _i]),
    'bdn_upsample2x_bwd_bs': (_i, [_i, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'bdn_enc_skip_bwd': (_i, [_i, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'bdn_enc_skip_bwd_rows': (_i, [_i, _i, _i, _i, _i]),
    'bdn_outc_fwd': (_i, [_i, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'bdn_outc_bwd_workspace_bytes': (_sz, [_i, _i, _i, _i, _i, _i]),
    'bdn_outc_bwd': (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'bdn_outc_bn_bwd_apply': (_i, [_i, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'bdn_outc_bwd_rows': (_i, [_i, _i, _i, _i, _i]),
    'bdn_overlap_workspace_bytes': (_sz, [_i, _i, _i, _i, _i]),
    'bdn_tversky': (_i, [_vp, _vp, _f, _f, _f, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'bdn_conv3x3_variant': (C.c_char_p, [_i, _i, _i, _i, _i, _i, _i, _i]),
    'bdn_conv3x3_dgrad_bb_variant': (C.c_char_p, [_i, _i, _i, _i, _i]),
    'bdn_conv3x3_x3src_variant': (C.c_char_p, [_i, _i, _i, _i, _i, _i, _i]),
    'bdn_conv3x3_dgrad_bs': (_i, [_i, _vp, _i, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp]),
    'bdn_conv3x3_dgrad_bb': (_i, [_i, _vp, _i, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'bdn_bn_bwd_scratch_bytes': (_sz, [_i, _i]),
    'bdn_bn_bwd_finalize': (_i, [_vp, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp]),
    'bdn_bn_bwd_apply': (_i, [_i, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    'bdn_bn_bwd_apply_split': (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp]),
    'bdn_overlap_loss': (_i, [_vp, _vp, _f, _f, _f, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'bdn_focal_workspace_bytes': (_sz, []),
    'bdn_focal': (_i, [_vp, _vp, _f, _vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    'bdn_ingest_band': (_i, [_i, _vp, _i, _i, _f, _f, _vp, _i, _i, _vp]),
    'bdn_gather_tiles': (_i, [_i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    'bdn_upload_band': (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'bdn_argmax': (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    'bdn_argmax_stitch': (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    'bdn_sgd_step': (_i, [_vp, _vp, _f, _f, _sz, _vp]),
}

_lib = None


def load():
    """Load the shared library (once).  Raises RuntimeError if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f'fabric_amd: HIP extension {LIB_PATH} not found. Build it with '
            f'`python -c "import __graft_entry__ as g; g.build()"` or `make -C fabric_amd/csrc`. '
            f'There is no CPU fallback.')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def ptr(t):
    """Device pointer of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()


# Per-call timing for bench.py's instrumented step (measurement only): when PROFILE is a list, every entry point that takes a
# stream (always its last argument) is bracketed by an event pair on that stream and (name, PHASE, stream, e0, e1) is appended.
PROFILE = None
PHASE = ''            # 'fwd' / 'bwd': set by the engine, lets the profile tell a data-gradient conv from a forward conv
_ext_streams = {}


def _profiled(lib, name, args):
    import torch
    h = args[-1] or 0
    st = _ext_streams.get(h)
    if st is None:
        st = _ext_streams[h] = torch.cuda.ExternalStream(h) if h else torch.cuda.default_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    rc = getattr(lib, name)(*args)
    e1.record(st)
    PROFILE.append((name, PHASE, h, e0, e1))
    return rc


SKIP = None           # timing diagnostics only (tools/ab_skip.py): callable(name, args) -> True drops the launch (results are then wrong)


def call(name, *args):
    """Invoke an int-returning entry point; raise RuntimeError with the library's message on failure."""
    lib = load()
    if SKIP is not None and SKIP(name, args):
        return
    if PROFILE is not None and SIGNATURES[name][1] and SIGNATURES[name][1][-1] is _vp and not name.startswith(('bdn_stream', 'bdn_event')):
        rc = _profiled(lib, name, args)
    else:
        rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.bdn_last_error().decode(errors='replace')
        raise RuntimeError(f'{name} failed (rc={rc}): {msg}')


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream
