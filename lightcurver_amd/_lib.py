"""ctypes binding of liblcmi.so (C ABI: include/lcmi.h).

There is no CPU fallback: importing this module never fails, but ``lib()`` raises
``RuntimeError`` if the shared library has not been built (``python -c 'import
__graft_entry__ as g; g.build()'`` or ``make -C lightcurver_amd/csrc``), and ``Context()`` raises
if no HIP device is visible.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'liblcmi.so')

P_A, P_CX, P_CY, P_DX, P_DY, P_ALPHA, P_H, P_MEAN, P_COUNT = range(9)
PARAM_INDEX = {'a': P_A, 'c_x': P_CX, 'c_y': P_CY, 'dx': P_DX, 'dy': P_DY, 'alpha': P_ALPHA,
               'h': P_H, 'mean': P_MEAN}

fp = C.POINTER(C.c_float)
dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int32)
vp = C.c_void_p
LBFGS_EVAL = C.CFUNCTYPE(C.c_int, C.c_void_p, dp, dp, dp)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p)
IPC_HANDLE_BYTES = 64
RCCL_ID_BYTES = 128


class AdabeliefCfg(C.Structure):
    _fields_ = [('init_learning_rate', C.c_float), ('schedule_learning_rate', C.c_int32),
                ('decay_rate', C.c_float), ('transition_steps', C.c_int32),
                ('b1', C.c_float), ('b2', C.c_float), ('eps', C.c_float), ('eps_root', C.c_float)]


class JointLossCfg(C.Structure):
    _fields_ = [('lam_scales', C.c_float), ('lam_hf', C.c_float), ('lam_positivity', C.c_float),
                ('lam_positivity_ps', C.c_float), ('lam_pts_source', C.c_float),
                ('lam_flux_uniformity', C.c_float), ('n_prior', C.c_int32),
                ('prior_cx_mean', fp), ('prior_cx_sigma', fp), ('prior_cy_mean', fp), ('prior_cy_sigma', fp)]


# every symbol include/lcmi.h declares: name -> (restype, argtypes)
SIGNATURES = {
    'lc_version': (C.c_int, []),
    'lc_ctx_create': (C.c_int, [C.c_int, C.POINTER(vp)]),
    'lc_ctx_destroy': (None, [vp]),
    'lc_last_error': (C.c_char_p, [vp]),
    'lc_copy_bandwidth': (C.c_int, [vp, C.c_int64, C.c_int, C.POINTER(C.c_float)]),
    'lc_ctx_marker': (C.c_int, [vp, C.c_int]),
    'lc_prepare_stamps': (C.c_int, [vp, C.c_int, C.c_int, fp, fp, fp, fp, fp, C.POINTER(C.c_uint8), C.c_float, C.c_float,
                                    C.c_int, fp, fp, fp, C.POINTER(C.c_int32), C.POINTER(C.c_float)]),
    'lc_ctx_stream': (C.c_int, [vp, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    'lc_ctx_synchronize': (C.c_int, [vp]),
    'lc_timer_start': (C.c_int, [vp]),
    'lc_timer_stop': (C.c_int, [vp, fp]),
    'lc_device_info': (C.c_int, [vp, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    'lc_adabelief_defaults': (None, [C.POINTER(AdabeliefCfg)]),
    'lc_psf_supported': (C.c_int, [C.c_int, C.c_int]),
    'lc_psf_batch_create': (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, fp, fp, C.POINTER(vp)]),
    'lc_psf_batch_destroy': (None, [vp]),
    'lc_psf_batch_set_moffat': (C.c_int, [vp, fp]),
    'lc_psf_batch_get_moffat': (C.c_int, [vp, fp]),
    'lc_psf_batch_set_stars': (C.c_int, [vp, fp]),
    'lc_psf_batch_get_stars': (C.c_int, [vp, fp]),
    'lc_psf_batch_set_grid': (C.c_int, [vp, fp]),
    'lc_psf_batch_get_grid': (C.c_int, [vp, fp]),
    'lc_psf_batch_set_regularization': (C.c_int, [vp, fp, C.c_float, C.c_float]),
    'lc_psf_batch_propagate_noise': (C.c_int, [vp]),
    'lc_psf_batch_get_weights': (C.c_int, [vp, fp]),
    'lc_psf_batch_eval': (C.c_int, [vp, fp, fp, fp, fp, fp, fp]),
    'lc_psf_batch_fit_moffat': (C.c_int, [vp, C.c_int, fp]),
    'lc_psf_batch_run_adabelief': (C.c_int, [vp, C.c_int, C.POINTER(AdabeliefCfg)]),
    'lc_psf_batch_iterations_done': (C.c_int, [vp]),
    'lc_psf_batch_split_fallbacks': (C.c_int, [vp, C.POINTER(C.c_int)]),
    'lc_psf_batch_get_loss_history': (C.c_int, [vp, fp, C.c_int]),
    'lc_psf_batch_get_results': (C.c_int, [vp, fp, fp, fp, fp]),
    'lc_psf_batch_set_moffat_q': (C.c_int, [vp, fp]),
    'lc_psf_batch_set_distortion': (C.c_int, [vp, C.c_int, fp, fp]),
    'lc_psf_distortion_forward': (C.c_int, [vp, vp]),
    'lc_psf_distortion_backward': (C.c_int, [vp, vp]),
    'lc_psf_batch_get_ext_grad': (C.c_int, [vp, fp]),
    'lc_psf_batch_step_adabelief': (C.c_int, [vp, C.POINTER(AdabeliefCfg), C.c_int, C.c_int]),
    'lc_psf_distortion_run': (C.c_int, [vp, vp, C.c_int, C.POINTER(AdabeliefCfg)]),
    'lc_batched_lbfgs': (C.c_int, [C.c_int, C.c_int, dp, dp, dp, C.c_int, LBFGS_EVAL, vp, dp, C.POINTER(C.c_int)]),
    'lc_apply_distortion': (C.c_int, [vp, C.c_int, C.c_int, fp, fp, fp, fp]),
    'lc_joint_supported': (C.c_int, [C.c_int, C.c_int]),
    'lc_joint_set_debug_global': (C.c_int, [C.c_int]),
    'lc_joint_create': (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, fp, fp, fp, C.POINTER(vp)]),
    'lc_joint_create_groups': (C.c_int, [vp, C.c_int, ip, C.c_int, C.c_int, C.c_int, fp, fp, fp, C.POINTER(vp)]),
    'lc_joint_get_group_loss_history': (C.c_int, [vp, fp, C.c_int]),
    'lc_joint_destroy': (None, [vp]),
    'lc_joint_set_param': (C.c_int, [vp, C.c_int, fp, C.c_int]),
    'lc_joint_get_param': (C.c_int, [vp, C.c_int, fp, C.c_int]),
    'lc_joint_set_flux_reference': (C.c_int, [vp, fp, C.c_int]),
    'lc_joint_get_flux_reference': (C.c_int, [vp, fp, C.c_int]),
    'lc_joint_set_free': (C.c_int, [vp, ip]),
    'lc_joint_set_loss': (C.c_int, [vp, C.POINTER(JointLossCfg), fp]),
    'lc_joint_propagate_noise': (C.c_int, [vp, fp]),
    'lc_joint_loss_grad': (C.c_int, [vp, fp, C.POINTER(fp)]),
    'lc_joint_model': (C.c_int, [vp, fp, fp]),
    'lc_joint_deconvolved': (C.c_int, [vp, C.c_int, fp, fp]),
    'lc_joint_run_adabelief': (C.c_int, [vp, C.c_int, C.POINTER(AdabeliefCfg)]),
    'lc_joint_run_lbfgs': (C.c_int, [vp, C.c_int, C.POINTER(fp), C.POINTER(fp), fp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'lc_joint_get_loss_history': (C.c_int, [vp, fp, C.c_int]),
    'lc_joint_iterations_done': (C.c_int, [vp]),
    'lc_joint_cluster_info': (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'lc_joint_param_history_begin': (C.c_int, [vp, C.c_int, C.POINTER(C.c_int)]),
    'lc_joint_param_history_rows': (C.c_int, [vp]),
    'lc_joint_param_history_get': (C.c_int, [vp, C.c_int, C.c_int, fp]),
    'lc_joint_param_history_end': (C.c_int, [vp]),
    'lc_joint_fisher_flux_sigma': (C.c_int, [vp, fp]),
    'lc_joint_step_local': (C.c_int, [vp]),
    'lc_joint_shared_buffer_dev': (C.c_int, [vp, C.POINTER(vp), C.POINTER(C.c_int)]),
    'lc_joint_step_update': (C.c_int, [vp, C.POINTER(AdabeliefCfg)]),
    'lc_joint_step_grad': (C.c_int, [vp, fp, C.POINTER(fp)]),
    'lc_joint_shared_get': (C.c_int, [vp, fp, C.c_int]),
    'lc_joint_shared_set': (C.c_int, [vp, fp, C.c_int]),
    'lc_joint_run_sharded': (C.c_int, [vp, C.c_int, C.POINTER(AdabeliefCfg), C.c_void_p, vp]),
    'lc_peer_group_create': (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
    'lc_peer_group_export': (C.c_int, [vp, C.c_void_p, C.c_int]),
    'lc_peer_group_connect': (C.c_int, [vp, C.c_void_p, C.c_int]),
    'lc_peer_allreduce': (C.c_int, [vp, vp, C.c_int, vp]),
    'lc_peer_group_status': (C.c_int, [vp]),
    'lc_peer_group_destroy': (None, [vp]),
    'lc_rccl_available': (C.c_int, []),
    'lc_rccl_unique_id': (C.c_int, [C.c_void_p, C.c_int]),
    'lc_rccl_group_create': (C.c_int, [vp, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
    'lc_rccl_allreduce': (C.c_int, [vp, vp, C.c_int, vp]),
    'lc_rccl_group_info': (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_longlong)]),
    'lc_rccl_group_destroy': (None, [vp]),
}

_lib = None


def _share_hip_runtime_with_torch():
    """A process can drive the GPU through ONE HIP runtime.  The PyTorch-ROCm wheel bundles its own
    libamdhip64 / libhsa-runtime64; if liblcmi.so pulled in the system copies first, a later torch.cuda (RCCL
    collectives of the sharded joint fit) would find no device.  So when torch is installed its runtime is
    loaded first and liblcmi.so binds to it (same soname); without torch the system ROCm runtime is used."""
    import importlib.util
    import sys
    if 'torch' in sys.modules:
        return
    spec = importlib.util.find_spec('torch')
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), 'lib', 'libamdhip64.so')
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """Load liblcmi.so and attach the prototypes; raises if the HIP extension is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'{LIB_PATH} is missing: the HIP extension has not been built '
                '(run __graft_entry__.build() or make -C lightcurver_amd/csrc). There is no CPU fallback.')
        _share_hip_runtime_with_torch()
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def ptr(a):
    return None if a is None else a.ctypes.data_as(fp)


def adabelief_cfg(init_learning_rate=1e-3, schedule_learning_rate=True, decay_rate=0.99,
                  transition_steps=10, b1=0.9, b2=0.999, eps=1e-16, eps_root=1e-16):
    return AdabeliefCfg(float(init_learning_rate), int(bool(schedule_learning_rate)), float(decay_rate),
                        int(transition_steps), float(b1), float(b2), float(eps), float(eps_root))


class LcError(RuntimeError):
    pass


class Context:
    """One lc_ctx (device + stream).  Raises if no GPU is visible: the product path never falls
    back to the CPU."""

    def __init__(self, device=0):
        self._l = lib()
        h = vp()
        rc = self._l.lc_ctx_create(int(device), C.byref(h))
        if rc != 0:
            raise LcError(f'lc_ctx_create failed ({rc}): {self._l.lc_last_error(None).decode()}')
        self.h = h

    def check(self, rc, what=''):
        if rc != 0:
            raise LcError(f'{what} failed ({rc}): {self._l.lc_last_error(self.h).decode()}')

    def synchronize(self):
        self.check(self._l.lc_ctx_synchronize(self.h), 'lc_ctx_synchronize')

    def marker(self, tag):
        """A dispatch with grid size ``tag`` * 64 threads between two synchronisations: cuts a profiler trace into sections."""
        self.check(self._l.lc_ctx_marker(self.h, int(tag)), 'lc_ctx_marker')

    def stream(self):
        """(hipStream_t as int, device ordinal) of this context."""
        p = C.c_void_p()
        d = C.c_int()
        self.check(self._l.lc_ctx_stream(self.h, C.byref(p), C.byref(d)), 'lc_ctx_stream')
        return int(p.value or 0), d.value

    def timer_start(self):
        self.check(self._l.lc_timer_start(self.h), 'lc_timer_start')

    def timer_stop(self):
        ms = C.c_float()
        self.check(self._l.lc_timer_stop(self.h, C.byref(ms)), 'lc_timer_stop')
        return ms.value

    def device_info(self):
        name = C.create_string_buffer(256)
        ncu = C.c_int()
        mem = C.c_int64()
        self.check(self._l.lc_device_info(self.h, name, 256, C.byref(ncu), C.byref(mem)), 'lc_device_info')
        return dict(name=name.value.decode(), n_cu=ncu.value, hbm_bytes=mem.value)

    def close(self):
        if getattr(self, 'h', None):
            self._l.lc_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        # (not while the interpreter shuts down: objects are then torn down in no particular order, and destroying a device
        #  object whose context has already gone is a crash at exit; the process is about to return everything anyway)
        try:
            import sys
            if not sys.is_finalizing():
                self.close()
        except Exception:
            pass


_default_ctx = {}


def default_context(device=0):
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]
