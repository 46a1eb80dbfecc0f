"""ctypes binding of libcae_hip.so (C ABI declared in include/cae_hip.h).

There is no CPU fallback: if the library is missing or fails to load, importing the hot
path raises.  Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``make -C cnn_autoencoder_amd/csrc``.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
import threading

import torch  # noqa: F401  (loads torch's libamdhip64 first so both share one HIP runtime)

_HERE = os.path.dirname(os.path.abspath(__file__))
# CAE_LIB: alternative build of the same library (kernel A/B experiments); never a different implementation
LIB_PATH = os.environ.get('CAE_LIB') or os.path.join(_HERE, 'libcae_hip.so')
_lock = threading.Lock()
_lib = None

c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_size_t = ctypes.c_size_t
c_float = ctypes.c_float

# every symbol include/cae_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    'cae_version': (c_int, []),
    'cae_last_error': (ctypes.c_char_p, []),
    'cae_free': (None, [c_void_p]),
    'cae_model_create': (c_int, [c_int, c_int, c_int, c_int, c_int, ctypes.POINTER(c_void_p)]),
    'cae_model_destroy': (None, [c_void_p]),
    'cae_model_set_layer': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'cae_model_set_layer_act': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'cae_model_set_precision': (c_int, [c_void_p, c_int]),
    'cae_last_range_ticket': (ctypes.c_int64, []),
    'cae_range_check': (c_int, [c_void_p, ctypes.c_int64, c_void_p]),
    'cae_thread_force_fp32': (None, [c_int]),
    'cae_model_effective_precision': (c_int, [c_void_p, c_void_p]),
    'cae_model_set_entropy': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'cae_analysis': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'cae_synthesis': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    'cae_model_set_layer_stage': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                           c_int, c_int]),
    'cae_model_set_color_layer': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'cae_synthesis_multiscale': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p,
                                          c_void_p]),
    'cae_analysis_symbols': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'cae_synthesis_symbols': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    'cae_gdn_forward': (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'cae_quantize': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'cae_quantize_export': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p]),
    'cae_copy_to_host': (c_int, [c_void_p, c_void_p, c_size_t]),
    'cae_dequantize': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'cae_model_set_density': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_float]),
    'cae_model_set_likelihood_form': (c_int, [c_void_p, c_int]),
    'cae_likelihood': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'cae_tile_ssim': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_size_t, c_void_p]),
    'cae_tile_delta_e': (c_int, [c_void_p, c_void_p, c_int, c_size_t, c_void_p, c_void_p, c_size_t, c_void_p]),
    'cae_u8hwc_to_planes': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'cae_avgpool2': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'cae_msssim_level': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'cae_tile_sse': (c_int, [c_void_p, c_void_p, c_int, c_size_t, c_void_p, c_void_p]),
    'cae_model_set_profiling': (c_int, [c_void_p, c_int]),
    'cae_model_get_profile': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int]),
    'cae_t_packed_bytes': (c_size_t, [c_int, c_int, c_int]),
    'cae_t_pack_weights': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'cae_t_from_nchw': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'cae_t_to_nchw': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'cae_t_conv_forward': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                    c_void_p, c_void_p]),
    'cae_t_conv_dgrad_ext': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_int,
                                      c_void_p]),
    'cae_t_deconv_forward': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                      c_void_p, c_void_p]),
    'cae_t_deconv_dgrad': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                    c_void_p]),
    'cae_t_wgrad': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p,
                             c_void_p]),
    'cae_t_gdn_forward': (c_int, [c_void_p, ctypes.c_long, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    'cae_t_gdn_backward': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                    c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'cae_t_conv_forward_act': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                        c_void_p, c_int, c_void_p]),
    'cae_t_deconv_forward_act': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                          c_void_p, c_int, c_void_p]),
    'cae_t_corr_s1': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p,
                               c_int, c_void_p]),
    'cae_t_wgrad_s1': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'cae_t_act_backward': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'cae_t_im2col_s2': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'cae_t_col2im_s2': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'cae_t_pointwise': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int,
                                 c_void_p]),
    'cae_t_wgrad_pointwise': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'cae_t_gdn_saved_elems': (c_size_t, [ctypes.c_long, c_int]),
    'cae_t_gdn_forward_save': (c_int, [c_void_p, ctypes.c_long, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    'cae_t_gdn_backward_fused': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int,
                                          c_void_p, c_void_p, c_void_p, c_void_p]),
    'cae_t_fold_to_bf16': (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'cae_t_bn_moments': (c_int, [c_void_p, c_void_p, c_int, c_int, ctypes.c_long, c_void_p, c_void_p, c_void_p]),
    'cae_t_bn_affine': (c_int, [c_void_p, c_void_p, c_int, c_int, ctypes.c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'cae_t_colsum': (c_int, [c_void_p, ctypes.c_long, c_int, c_void_p, c_void_p]),
    'cae_t_density_params': (c_int, [c_int, c_int]),
    'cae_t_reparam_forward': (c_int, [c_void_p, ctypes.c_long, ctypes.c_float, ctypes.c_float, c_void_p, c_void_p]),
    'cae_t_reparam_backward': (c_int, [c_void_p, c_void_p, ctypes.c_long, ctypes.c_float, c_void_p, c_void_p]),
    'cae_t_density_forward': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, ctypes.c_float, c_void_p,
                                      c_void_p, c_void_p]),
    'cae_t_density_backward': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, ctypes.c_float,
                                       c_void_p, c_void_p, c_void_p]),
    'cae_t_clip_adam': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    'cae_cpu_budget': (c_int, []),
    'cae_coder_lockstep': (c_int, []),
    'cae_coder_threads': (c_int, [c_int, c_int]),
    'cae_pmf_to_quantized_cdf': (c_int, [c_void_p, c_int, c_int, c_void_p]),
    'cae_rans_encode_batch': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int]),
    'cae_rans_encode_packed': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int]),
    'cae_rans_decode_batch': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_int]),
    'cae_door_create': (c_int, [c_void_p, c_void_p, c_int, c_int, ctypes.POINTER(c_void_p)]),
    'cae_door_destroy': (None, [c_void_p]),
    'cae_door_encode': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, ctypes.POINTER(c_void_p),
                                ctypes.POINTER(c_size_t)]),
    'cae_door_decode_shape': (c_int, [c_void_p, c_void_p, c_size_t, ctypes.POINTER(c_int), ctypes.POINTER(c_int),
                                      ctypes.POINTER(c_int)]),
    'cae_door_decode': (c_int, [c_void_p, c_void_p, c_size_t, c_void_p, c_size_t]),
    'cae_door_stats': (c_int, [c_void_p, c_void_p, c_int, c_int]),
    'cae_door_hold': (c_int, [c_void_p, c_int]),
}

CAE_ANALYSIS, CAE_SYNTHESIS = 0, 1
FMT_U8_HWC, FMT_F32_NCHW = 0, 1


class CaeError(RuntimeError):
    pass


def build(verbose: bool = False) -> str:
    """Compile libcae_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ['make', '-j8', '-C', os.path.join(_HERE, 'csrc'), 'all']
    if not verbose:
        cmd.insert(1, '-s')
    subprocess.check_call(cmd)
    return LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise ImportError(
                    f'{LIB_PATH} not found: the HIP extension is not built and there is no CPU fallback. '
                    'Run `make -C cnn_autoencoder_amd/csrc` (needs hipcc).')
            L = ctypes.CDLL(LIB_PATH)
            for name, (res, args) in SYMBOLS.items():
                fn = getattr(L, name)  # AttributeError if the library does not export it
                fn.restype = res
                fn.argtypes = args
            _lib = L
    return _lib


def check(rc: int):
    if rc != 0:
        msg = lib().cae_last_error().decode('utf-8', 'replace')
        if rc == -1:
            raise ValueError(msg)  # the reference / compressai raise ValueError on bad shapes
        raise CaeError(f'libcae_hip error {rc}: {msg}')


def require_gpu() -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError('cnn_autoencoder_amd: no HIP device is visible; the MI355X hot path has no CPU fallback')
    return torch.device('cuda', torch.cuda.current_device())


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


class Handle:
    """Owns one cae_model_t."""

    def __init__(self, c_org: int, c_net: int, c_bn: int, level: int, ks: int):
        self._h = c_void_p()
        check(lib().cae_model_create(c_org, c_net, c_bn, level, ks, ctypes.byref(self._h)))

    @property
    def ptr(self):
        return self._h

    def close(self):
        if self._h:
            lib().cae_model_destroy(self._h)
            self._h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
