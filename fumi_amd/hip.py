"""ctypes binding of ``include/fumi_hip.h`` (the gfx950 engine, ``fumi_amd/lib/libfumi_hip.so``).

PyTorch is plumbing here: it owns device memory and the HIP stream; every tensor is handed to the
library as a raw device pointer.  There is NO CPU fallback: if the library is missing, or a tensor
is not on a GPU, these functions raise.
"""
import ctypes
import os
import threading
from ctypes import c_int, c_int64, c_float, c_void_p, c_size_t, c_char_p, POINTER

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libfumi_hip.so")

# every symbol include/fumi_hip.h declares (tests/test_abi.py checks the shared object exports them all)
SYMBOLS = [
    "fumi_hip_version", "fumi_hip_strerror", "fumi_hip_last_hip_error",
    "fumi_hip_workspace_create", "fumi_hip_workspace_destroy", "fumi_hip_workspace_bytes", "fumi_hip_read_status",
    "fumi_hip_set_spin_limit", "fumi_hip_set_trace_buffer",
    "fumi_hip_set_profiling", "fumi_hip_set_profiling_every", "fumi_hip_get_profile", "fumi_hip_phase_name",
    "fumi_hip_fumi_step", "fumi_hip_fumi_step_indexed", "fumi_hip_maml_step", "fumi_hip_am3_step",
    "fumi_hip_glove_bag", "fumi_hip_glove_bag_select", "fumi_hip_glove_bag_select_deferred", "fumi_hip_glove_flush", "fumi_hip_class_text_select", "fumi_hip_xpanel_fwd", "fumi_hip_xpanel_bwd",
    "fumi_hip_adam_step", "fumi_hip_adam_step_deferred", "fumi_hip_adam_flush",
    "fumi_hip_linear_fwd", "fumi_hip_linear_bwd_data", "fumi_hip_linear_bwd_weight",
    "fumi_hip_sample_episodes", "fumi_hip_sample_episodes_tm", "fumi_hip_gather_rows", "fumi_hip_publish_scalars",
    "fumi_hip_publish_scalars_deferred", "fumi_hip_publish_flush", "fumi_hip_am3_metrics",
    "fumi_hip_conv4_feature_dim", "fumi_hip_fumi_conv4_step", "fumi_hip_maml_conv4_step", "fumi_hip_conv4_probe", "fumi_hip_conv4_features", "fumi_hip_conv4_set_option",
    "fumi_hip_conv4_encode", "fumi_hip_conv4_encode_bwd", "fumi_hip_am3_step_dx",
    "fumi_hip_resnet12_set_budget", "fumi_hip_fumi_resnet12_step", "fumi_hip_maml_resnet12_step", "fumi_hip_resnet12_features",
    "fumi_hip_rn12_conv", "fumi_hip_rn12_wgrad", "fumi_hip_resnet12_set_option", "fumi_hip_rn12_probe",
    "fumi_hip_conv3x3_fwd", "fumi_hip_conv3x3_bwd_data", "fumi_hip_conv3x3_bwd_weight",
    "fumi_hip_sgd_axpy", "fumi_hip_ce_fwd_bwd", "fumi_hip_proto_reduce", "fumi_hip_clip_step", "fumi_hip_lstm_bidir", "fumi_hip_lstm_tape_floats", "fumi_hip_lstm_bidir_train", "fumi_hip_lstm_bidir_bwd",
    "fumi_hip_want_text_grad",
]

ST_LABEL_RANGE, ST_CLASS_MISSING, ST_SYNC_TIMEOUT = 1, 2, 4
N_PHASES = 18

_lib = None
_lock = threading.Lock()


class FumiHipError(RuntimeError):
    pass

class RowRef:
    """Zero-copy stand-in for an image tensor x[B, rows, D]: x[b, r] = table[idx[b, r]] with ``table`` [n_rows, D] fp32 resident
    on the device (fumi_hip_fumi_step_indexed).  Quacks like the tensor where FUMI.evaluate touches it."""
    __slots__ = ("table", "idx")

    def __init__(self, table, idx):
        if table.dim() != 2 or idx.dim() != 2 or idx.dtype != torch.int64 or table.dtype != torch.float32:
            raise FumiHipError("RowRef: table must be [n_rows, D] fp32 and idx [B, rows] int64")
        self.table, self.idx = table, idx

    shape = property(lambda self: (self.idx.shape[0], self.idx.shape[1], self.table.shape[1]))
    device = property(lambda self: self.table.device)
    is_cuda = property(lambda self: self.table.is_cuda)

    def __getitem__(self, sl):
        return RowRef(self.table, self.idx[sl])

    def to(self, *a, **k):
        return self

    def contiguous(self):
        return RowRef(self.table, self.idx.contiguous())

    def float(self):
        return self

    dtype = torch.float32

    def is_contiguous(self):
        return self.idx.is_contiguous()

    def materialize(self):
        return self.table[self.idx]



def lib():
    """Load the shared object once.  Fails loudly when it has not been built (``python __graft_entry__.py``)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise FumiHipError(
                f"{LIB_PATH} is missing: the MI355X engine has no CPU fallback. Build it with "
                f"`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950).")
        L = ctypes.CDLL(LIB_PATH)
        L.fumi_hip_version.restype = c_int
        L.fumi_hip_strerror.restype = c_char_p
        L.fumi_hip_strerror.argtypes = [c_int]
        L.fumi_hip_last_hip_error.restype = c_char_p
        L.fumi_hip_workspace_create.argtypes = [c_int, c_size_t, POINTER(c_void_p)]
        L.fumi_hip_workspace_destroy.argtypes = [c_void_p]
        L.fumi_hip_workspace_destroy.restype = None
        L.fumi_hip_workspace_bytes.argtypes = [c_void_p]
        L.fumi_hip_workspace_bytes.restype = c_size_t
        L.fumi_hip_read_status.argtypes = [c_void_p, c_void_p, POINTER(c_int)]
        L.fumi_hip_set_spin_limit.argtypes = [c_int]
        L.fumi_hip_set_trace_buffer.argtypes = [c_int, c_void_p]
        L.fumi_hip_set_profiling.argtypes = [c_void_p, c_int]
        L.fumi_hip_set_profiling_every.argtypes = [c_void_p, c_int]
        L.fumi_hip_get_profile.argtypes = [c_void_p, c_int, POINTER(ctypes.c_double), POINTER(c_int)]
        L.fumi_hip_phase_name.argtypes = [c_int]
        L.fumi_hip_phase_name.restype = c_char_p
        PP = POINTER(c_void_p)
        L.fumi_hip_fumi_step.argtypes = (
            [c_void_p, c_void_p] + [c_int] * 6 + [POINTER(c_int), c_int, c_int, c_int, c_float, c_int, c_int, c_float,
                                                 c_float, ctypes.c_uint64]
            + [c_void_p] * 6 + [PP, PP] + [c_void_p] * 6 + [PP, PP])
        L.fumi_hip_fumi_step_indexed.argtypes = (
            [c_void_p, c_void_p] + [c_int] * 6 + [POINTER(c_int), c_int, c_int, c_int, c_float, c_int, c_int, c_float,
                                                 c_float, ctypes.c_uint64]
            + [c_void_p, c_int64] + [c_void_p] * 6 + [PP, PP] + [c_void_p] * 6 + [PP, PP])
        L.fumi_hip_maml_step.argtypes = (
            [c_void_p, c_void_p] + [c_int] * 6 + [POINTER(c_int), c_int, c_float, c_int, c_int, c_float]
            + [c_void_p] * 4 + [PP] + [c_void_p] * 6 + [PP])
        L.fumi_hip_am3_step.argtypes = (
            [c_void_p, c_void_p] + [c_int] * 10 + [c_float, c_float, ctypes.c_uint64] + [c_void_p] * 5 + [PP] + [c_void_p] * 4 + [PP, c_void_p])
        L.fumi_hip_am3_step_dx.argtypes = L.fumi_hip_am3_step.argtypes + [c_void_p, c_void_p]
        L.fumi_hip_am3_metrics.argtypes = [c_void_p, c_void_p, c_int, c_void_p, c_void_p]
        L.fumi_hip_glove_bag.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int64, c_void_p, c_int, c_int,
                                         c_int, c_void_p]
        L.fumi_hip_glove_bag_select.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int64,
                                                c_void_p, c_int, c_int, c_int, c_void_p]
        L.fumi_hip_glove_bag_select_deferred.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int64,
                                                         c_void_p, c_int, c_int, c_int, c_void_p]
        L.fumi_hip_glove_flush.argtypes = [c_void_p, c_void_p]
        L.fumi_hip_class_text_select.argtypes = [c_void_p, c_void_p] + [c_int] * 4 + [c_void_p] * 3
        L.fumi_hip_xpanel_fwd.argtypes = [c_void_p, c_void_p] + [c_int] * 5 + [c_void_p] * 5
        L.fumi_hip_xpanel_bwd.argtypes = [c_void_p, c_void_p] + [c_int] * 5 + [c_void_p] * 3 + [c_float, c_void_p]
        L.fumi_hip_adam_step.argtypes = [c_void_p, c_void_p, c_int, PP, PP, PP, PP, POINTER(ctypes.c_long)] + [c_float] * 5 + [c_int]
        L.fumi_hip_adam_step_deferred.argtypes = [c_void_p, c_int, PP, PP, PP, PP, POINTER(ctypes.c_long)] + [c_float] * 5 + [c_int]
        L.fumi_hip_adam_flush.argtypes = [c_void_p, c_void_p, POINTER(c_int)]
        L.fumi_hip_linear_fwd.argtypes = [c_void_p, c_void_p] + [c_int] * 3 + [c_void_p] * 3 + [c_int, c_void_p]
        L.fumi_hip_linear_bwd_data.argtypes = [c_void_p, c_void_p] + [c_int] * 3 + [c_void_p] * 3
        L.fumi_hip_linear_bwd_weight.argtypes = [c_void_p, c_void_p] + [c_int] * 3 + [c_void_p] * 4
        L.fumi_hip_sample_episodes.argtypes = [c_void_p, c_void_p, ctypes.c_uint64, ctypes.c_uint64] + [c_int] * 5 + [c_void_p] * 5
        L.fumi_hip_sample_episodes_tm.argtypes = ([c_void_p, c_void_p, ctypes.c_uint64, ctypes.c_uint64] + [c_int] * 5 + [c_void_p] * 2
                                                  + [c_int] + [c_void_p] * 4)
        L.fumi_hip_gather_rows.argtypes = [c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p]
        L.fumi_hip_publish_scalars.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_void_p, ctypes.c_uint64]
        L.fumi_hip_publish_scalars_deferred.argtypes = [c_void_p, c_void_p, c_int, c_void_p, ctypes.c_uint64]
        L.fumi_hip_publish_flush.argtypes = [c_void_p, c_void_p]
        L.fumi_hip_conv4_feature_dim.argtypes = [c_int] * 3
        L.fumi_hip_fumi_conv4_step.argtypes = (
            [c_void_p, c_void_p] + [c_int] * 10 + [c_int, c_float, c_int, c_int, c_float]
            + [c_void_p] * 6 + [PP, PP] + [c_void_p] * 6 + [PP, PP])
        L.fumi_hip_maml_conv4_step.argtypes = (
            [c_void_p, c_void_p] + [c_int] * 8 + [c_int, c_float, c_int, c_int, c_float]
            + [c_void_p] * 4 + [PP] + [c_void_p] * 6 + [PP])
        L.fumi_hip_conv4_set_option.argtypes = [c_int, c_int]
        PI = POINTER(c_int)
        L.fumi_hip_resnet12_set_budget.argtypes = [ctypes.c_double]
        L.fumi_hip_fumi_resnet12_step.argtypes = (
            [c_void_p, c_void_p] + [c_int] * 8 + [PI, c_int, c_int] + [c_int, c_float, c_int, c_int, c_float, c_int]
            + [c_void_p] * 6 + [PP, PP] + [c_void_p] * 6 + [PP, PP])
        L.fumi_hip_maml_resnet12_step.argtypes = (
            [c_void_p, c_void_p] + [c_int] * 8 + [PI] + [c_int, c_float, c_int, c_int, c_float, c_int]
            + [c_void_p] * 4 + [PP] + [c_void_p] * 6 + [PP])
        L.fumi_hip_resnet12_features.argtypes = [c_void_p, c_void_p] + [c_int] * 6 + [PI, c_void_p, PP, c_void_p]
        L.fumi_hip_rn12_conv.argtypes = [c_void_p, c_void_p] + [c_int] * 8 + [c_void_p] * 4
        L.fumi_hip_rn12_wgrad.argtypes = [c_void_p, c_void_p] + [c_int] * 7 + [c_void_p] * 3
        L.fumi_hip_resnet12_set_option.argtypes = [c_int, c_int]
        L.fumi_hip_rn12_probe.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_size_t, POINTER(c_size_t),
                                          POINTER(c_int)]
        L.fumi_hip_conv4_features.argtypes = [c_void_p, c_void_p] + [c_int] * 6 + [c_void_p, PP, c_void_p]
        L.fumi_hip_conv4_encode.argtypes = [c_void_p, c_void_p] + [c_int] * 7 + [c_void_p, c_void_p, PP, c_void_p, c_void_p, c_int]
        L.fumi_hip_conv4_encode_bwd.argtypes = [c_void_p, c_void_p] + [c_int] * 7 + [c_void_p] * 4 + [c_float, PP]
        L.fumi_hip_conv4_probe.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_size_t, POINTER(c_size_t)]
        for fn in (L.fumi_hip_conv3x3_fwd, L.fumi_hip_conv3x3_bwd_data, L.fumi_hip_conv3x3_bwd_weight):
            fn.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]
        L.fumi_hip_sgd_axpy.argtypes = [c_void_p, c_void_p, ctypes.c_long, c_void_p, c_float, c_void_p, c_void_p]
        L.fumi_hip_ce_fwd_bwd.argtypes = [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]
        L.fumi_hip_clip_step.argtypes = [c_void_p, c_void_p] + [c_int] * 5 + [c_void_p, c_void_p, PP, c_int, c_void_p, c_void_p, PP]
        L.fumi_hip_lstm_bidir.argtypes = [c_void_p, c_void_p] + [c_int] * 4 + [c_void_p, c_int64, c_void_p, c_int64, PP, c_int, c_void_p]
        L.fumi_hip_lstm_tape_floats.argtypes = [c_int] * 4
        L.fumi_hip_lstm_tape_floats.restype = c_int64
        L.fumi_hip_lstm_bidir_train.argtypes = [c_void_p, c_void_p] + [c_int] * 4 + [c_void_p, c_int64, c_void_p, c_int64, PP, c_int,
                                                c_void_p, c_void_p]
        L.fumi_hip_lstm_bidir_bwd.argtypes = [c_void_p, c_void_p] + [c_int] * 4 + [c_void_p, c_int64, PP, c_int, c_void_p, c_void_p, PP]
        L.fumi_hip_want_text_grad.argtypes = [c_void_p, c_void_p]
        L.fumi_hip_proto_reduce.argtypes = [c_void_p, c_void_p] + [c_int] * 4 + [c_void_p] * 3
        _lib = L
    return _lib


def _check(rc, what):
    if rc != 0:
        L = lib()
        msg = L.fumi_hip_strerror(rc).decode()
        if rc == -3:
            msg += ": " + L.fumi_hip_last_hip_error().decode()
        raise FumiHipError(f"{what} failed: {msg} ({rc})")


def _dev(t):
    if isinstance(t, RowRef):
        t = t.table
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise FumiHipError("the MI355X engine needs GPU tensors (there is no CPU path); got "
                           f"{getattr(t, 'device', type(t))}")
    return t.device


def _f32(t, name):
    _dev(t)
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise FumiHipError(f"{name}: expected a contiguous float32 tensor, got {t.dtype} contiguous={t.is_contiguous()}")
    return c_void_p(t.data_ptr())


def _i64(t, name):
    _dev(t)
    if t.dtype != torch.int64 or not t.is_contiguous():
        raise FumiHipError(f"{name}: expected a contiguous int64 tensor, got {t.dtype}")
    return c_void_p(t.data_ptr())


def _shape(t, shape, name):
    """The kernels index raw pointers with the sizes they are told: a tensor of another shape would be read out of bounds (the
    reference raises a matmul shape error there, e.g. an embedding file whose width differs from --im_emb_dim)."""
    got = tuple(t.shape)
    if got != tuple(shape):
        raise FumiHipError(f"{name}: expected shape {tuple(shape)}, got {got}")


def _mlp_shapes(params, D, name, n_layers):
    d = D
    for i in range(n_layers):
        h = int(params[2 * i].shape[0])
        _shape(params[2 * i], (h, d), f"{name}[{2 * i}] (weight of layer {i})")
        _shape(params[2 * i + 1], (h,), f"{name}[{2 * i + 1}] (bias of layer {i})")
        d = h
    return d


_PARR_CACHE = {}


def _parr(tensors, name):
    """Pointer table of a parameter / gradient list.  The lists are the same tensors step after step, so a validated table is
    kept per (name, data pointers) -- rebuilding it costs more host time than the launch it feeds."""
    key = (name,) + tuple(t.data_ptr() for t in tensors)
    arr = _PARR_CACHE.get(key)
    if arr is not None:
        for t in tensors:                         # an address can be reused by a different tensor: keep the type check
            if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
                arr = None
                break
    if arr is None:
        arr = (c_void_p * len(tensors))()
        for i, t in enumerate(tensors):
            arr[i] = _f32(t, f"{name}[{i}]").value
        if len(_PARR_CACHE) > 256:
            _PARR_CACHE.clear()
        _PARR_CACHE[key] = arr
    return arr


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(device):
    """The caller's current stream on `device` (torch.cuda.current_stream builds a Stream object: ~1.4 us a call, and a step
    asks several times)."""
    if _RAW_STREAM is not None:
        idx = device.index
        return c_void_p(_RAW_STREAM(idx if idx is not None else torch.cuda.current_device()))
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


class Workspace:
    """One per (process, device).  Owns the engine's scratch slab; not re-entrant."""

    _per_device = {}

    def __init__(self, device, bytes_hint=0):
        device = torch.device(device)
        if device.type != "cuda":
            raise FumiHipError("the MI355X engine has no CPU path")
        self.device = torch.device("cuda", device.index if device.index is not None else torch.cuda.current_device())
        h = c_void_p()
        with torch.cuda.device(self.device):
            _check(lib().fumi_hip_workspace_create(self.device.index, bytes_hint, ctypes.byref(h)), "workspace_create")
        self._h = h

    @classmethod
    def get(cls, device, role=None):
        """The device's workspace; ``role`` names a further one (the Conv4 encoder's tape lives in its own slab, so that the
        step it feeds can lay out the main one)."""
        device = torch.device(device)
        idx = device.index if device.index is not None else torch.cuda.current_device()
        key = idx if role is None else (idx, role)
        ws = cls._per_device.get(key)
        if ws is None:
            ws = cls._per_device[key] = Workspace(torch.device("cuda", idx))
        return ws

    @property
    def handle(self):
        return self._h

    def bytes(self):
        return int(lib().fumi_hip_workspace_bytes(self._h))

    def read_status(self):
        """Synchronises the current stream; returns and clears the device status bits."""
        st = c_int(0)
        _check(lib().fumi_hip_read_status(self._h, _stream(self.device), ctypes.byref(st)), "read_status")
        return st.value

    def set_profiling(self, on, phases=None, every=1):
        """HIP-event timing of the library's phases (bench.py); switching it clears the records.  ``phases``: names
        (fumi_hip_phase_name) to time -- an event pair costs stream time (two ~6 us bubbles), so the bench times only the
        kernel its roofline is about and only at every ``every``-th step; None = every phase."""
        _check(lib().fumi_hip_set_profiling_every(self._h, max(1, int(every))), "set_profiling_every")
        mask = 0
        if on:
            if phases is None:
                mask = -1
            else:
                names = {lib().fumi_hip_phase_name(i).decode(): i for i in range(32)}
                for p in phases:
                    mask |= 1 << names[p]
        _check(lib().fumi_hip_set_profiling(self._h, mask), "set_profiling")

    def profile(self):
        """{phase name: (total ms, launches)} since profiling was switched on (synchronises the device)."""
        out = {}
        L = lib()
        for ph in range(N_PHASES):
            ms, n = ctypes.c_double(0), c_int(0)
            _check(L.fumi_hip_get_profile(self._h, ph, ctypes.byref(ms), ctypes.byref(n)), "get_profile")
            if n.value:
                out[L.fumi_hip_phase_name(ph).decode()] = (ms.value, n.value)
        return out

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            lib().fumi_hip_workspace_destroy(self._h)
            self._h = c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def raise_on_status(status):
    if status & ST_SYNC_TIMEOUT:
        raise RuntimeError("a workgroup timed out waiting for the sibling workgroups of its episode "
                           "(FUMI_ST_SYNC_TIMEOUT): the results of that step are invalid")
    if status & ST_CLASS_MISSING:
        raise IndexError("a class has no support sample (the reference raises IndexError at fumi/models/fumi.py:209)")
    if status & ST_LABEL_RANGE:
        raise IndexError("label / token id out of range")


# ----------------------------------------------------------------------------------------------------------------
def fumi_step(ws, x_s, y_s, x_q, y_q, theta, phi, T, alpha, tanh_head, *, cls_text=None, text_s=None,
              need_grad=True, grad_scale=None, g_theta=None, g_phi=None, stats=None, dropout_p=0.0, seed=0):
    """One FuMI meta-step over B episodes (fumi/models/fumi.py:146-192).  Returns a dict of GPU tensors."""
    dev = _dev(x_s)
    B, S, D = x_s.shape
    Qn = x_q.shape[1]
    n_hidden = len(theta) // 2
    hid = [int(theta[2 * i].shape[0]) for i in range(n_hidden)]
    Ht, Dt = int(phi[0].shape[0]), int(phi[0].shape[1])
    N = int((cls_text.shape[1]) if cls_text is not None else 0) or None
    if N is None:
        raise FumiHipError("fumi_step: pass n_way through cls_text=[B,N,Dt] or use fumi_step_select")
    return _fumi_step(ws, dev, B, N, S, Qn, D, hid, Dt, Ht, x_s, y_s, x_q, y_q, theta, phi, T, alpha, tanh_head,
                      cls_text, text_s, need_grad, grad_scale, g_theta, g_phi, stats, dropout_p, seed)


def fumi_step_select(ws, n_way, x_s, y_s, x_q, y_q, text_s, theta, phi, T, alpha, tanh_head, *,
                     need_grad=True, grad_scale=None, g_theta=None, g_phi=None, stats=None, dropout_p=0.0, seed=0):
    """Same, selecting the per-class text rows from text_s [B,S,Dt] on the device (fumi.py:207-210)."""
    dev = _dev(x_s)
    B, S, D = x_s.shape
    Qn = x_q.shape[1]
    n_hidden = len(theta) // 2
    hid = [int(theta[2 * i].shape[0]) for i in range(n_hidden)]
    Ht, Dt = int(phi[0].shape[0]), int(phi[0].shape[1])
    return _fumi_step(ws, dev, B, n_way, S, Qn, D, hid, Dt, Ht, x_s, y_s, x_q, y_q, theta, phi, T, alpha, tanh_head,
                      None, text_s, need_grad, grad_scale, g_theta, g_phi, stats, dropout_p, seed)


# Parameter / gradient LISTS that a model hands over unchanged step after step (the same list objects: fumi_amd/models cache
# them) are validated once: shapes, dtypes, contiguity and the pointer tables are remembered per tuple of list identities
# (the lists are held here, so an id cannot be recycled) and re-checked only through the first tensor's address.  Callers that
# build fresh lists every call take the full checks every call.
_PARAMSETS = {}


def _paramset(key_lists, dims, build):
    key = tuple(id(x) for x in key_lists) + dims
    ent = _PARAMSETS.get(key)
    if ent is not None and all(a is b for a, b in zip(ent[0], key_lists)) and ent[1] == key_lists[0][0].data_ptr():
        return ent[2]
    val = build()
    if len(_PARAMSETS) > 64:
        _PARAMSETS.clear()
    _PARAMSETS[key] = (tuple(key_lists), key_lists[0][0].data_ptr(), val)
    return val


def _fumi_step(ws, dev, B, N, S, Qn, D, hid, Dt, Ht, x_s, y_s, x_q, y_q, theta, phi, T, alpha, tanh_head,
               cls_text, text_s, need_grad, grad_scale, g_theta, g_phi, stats=None, dropout_p=0.0, seed=0):
    L = lib()

    def check_params():
        if len(theta) != 2 * len(hid) or len(phi) != 4 or not hid:
            raise FumiHipError("fumi_step: theta must hold (weight, bias) per hidden layer and phi the 4 hypernetwork tensors")
        H = _mlp_shapes(theta, D, "theta", len(hid))
        _shape(phi[0], (Ht, Dt), "phi[0]"); _shape(phi[1], (Ht,), "phi[1]"); _shape(phi[2], (H + 1, Ht), "phi[2]")
        _shape(phi[3], (H + 1,), "phi[3]")
        if need_grad and g_theta is not None:
            for i, (g, t) in enumerate(zip(list(g_theta) + list(g_phi or []), list(theta) + list(phi))):
                _shape(g, t.shape, f"gradient buffer {i}")
        return (_parr(theta, "theta"), _parr(phi, "phi"),
                _parr(g_theta, "g_theta") if (need_grad and g_theta is not None) else None,
                _parr(g_phi, "g_phi") if (need_grad and g_phi is not None) else None)

    if need_grad and g_theta is not None and g_phi is not None:
        arrs = _paramset((theta, phi, g_theta, g_phi), (D, Dt, Ht, tuple(hid)), check_params)
    else:
        arrs = check_params()
    _shape(x_q, (B, Qn, D), "x_q"); _shape(y_s, (B, S), "y_s"); _shape(y_q, (B, Qn), "y_q")
    if cls_text is not None:
        _shape(cls_text, (B, N, Dt), "cls_text")
    else:
        _shape(text_s, (B, S, Dt), "text_s")
    logits = torch.empty(B, Qn, N, device=dev, dtype=torch.float32)
    preds = torch.empty(B, Qn, device=dev, dtype=torch.int64)
    preds_f = torch.empty(B, Qn, device=dev, dtype=torch.float32)      # the reference's float test_preds, same launch
    loss_b = torch.empty(B, device=dev, dtype=torch.float32)
    acc_b = torch.empty(B, device=dev, dtype=torch.float32)
    if need_grad:
        if g_theta is None:
            g_theta = [torch.empty_like(t) for t in theta]
        if g_phi is None:
            g_phi = [torch.empty_like(t) for t in phi]
    if grad_scale is None:
        grad_scale = 1.0 / B
    hid_arr = (c_int * len(hid))(*hid)
    head = (ws.handle, _stream(dev), B, N, S, Qn, D, len(hid), hid_arr, Dt, Ht, int(T), float(alpha), int(bool(tanh_head)),
            int(bool(need_grad)), float(grad_scale), float(dropout_p), int(seed) & 0xFFFFFFFFFFFFFFFF)
    if isinstance(x_s, RowRef) or isinstance(x_q, RowRef):             # zero-copy episodes: rows stay in the table
        if not (isinstance(x_s, RowRef) and isinstance(x_q, RowRef)) or x_s.table.data_ptr() != x_q.table.data_ptr():
            raise FumiHipError("zero-copy episodes: support and query rows must be RowRefs into the same table")
        fn, rows = L.fumi_hip_fumi_step_indexed, (_f32(x_s.table, "table"), int(x_s.table.shape[0]), _i64(x_s.idx, "idx_s"),
                                                  _i64(y_s, "y_s"), _i64(x_q.idx, "idx_q"), _i64(y_q, "y_q"))
    else:
        fn, rows = L.fumi_hip_fumi_step, (_f32(x_s, "x_s"), _i64(y_s, "y_s"), _f32(x_q, "x_q"), _i64(y_q, "y_q"))
    rc = fn(
        *head, *rows,
        _f32(cls_text, "cls_text") if cls_text is not None else None,
        _f32(text_s, "text_s") if text_s is not None else None,
        arrs[0], arrs[1],
        c_void_p(logits.data_ptr()), c_void_p(preds.data_ptr()), c_void_p(preds_f.data_ptr()), c_void_p(loss_b.data_ptr()),
        c_void_p(acc_b.data_ptr()),
        _f32(stats, "stats") if stats is not None else None,
        (arrs[2] if arrs[2] is not None else _parr(g_theta, "g_theta")) if need_grad else None,
        (arrs[3] if arrs[3] is not None else _parr(g_phi, "g_phi")) if need_grad else None)
    _check(rc, "fumi_hip_fumi_step")
    return dict(logits=logits, preds=preds, preds_f=preds_f, loss_b=loss_b, acc_b=acc_b, g_theta=g_theta, g_phi=g_phi, stats=stats)


def maml_step(ws, x_s, y_s, x_q, y_q, params, T, alpha, first_order=False, *, need_grad=True, grad_scale=None,
              g_params=None, stats=None):
    """One MAML meta-step (fumi/models/maml.py:156-191).  params = hidden (W,b)* then lin_final W [N,H], b [N]."""
    dev = _dev(x_s)
    L = lib()
    B, S, D = x_s.shape
    Qn = x_q.shape[1]
    n_hidden = len(params) // 2 - 1
    if n_hidden < 0 or len(params) != 2 * n_hidden + 2:
        raise FumiHipError("maml_step: params must hold (weight, bias) per hidden layer, then lin_final's")
    hid = [int(params[2 * i].shape[0]) for i in range(n_hidden)]
    N = int(params[-2].shape[0])
    H = _mlp_shapes(params, D, "params", n_hidden)
    _shape(params[-2], (N, H), "params[-2] (lin_final.weight)"); _shape(params[-1], (N,), "params[-1] (lin_final.bias)")
    _shape(x_q, (B, Qn, D), "x_q"); _shape(y_s, (B, S), "y_s"); _shape(y_q, (B, Qn), "y_q")
    if need_grad and g_params is not None:
        for i, (g, t) in enumerate(zip(g_params, params)):
            _shape(g, t.shape, f"g_params[{i}]")
    logits = torch.empty(B, Qn, N, device=dev, dtype=torch.float32)
    preds = torch.empty(B, Qn, device=dev, dtype=torch.int64)
    preds_f = torch.empty(B, Qn, device=dev, dtype=torch.float32)      # the reference's float test_preds, same launch
    loss_b = torch.empty(B, device=dev, dtype=torch.float32)
    acc_b = torch.empty(B, device=dev, dtype=torch.float32)
    if need_grad and g_params is None:
        g_params = [torch.empty_like(t) for t in params]
    if grad_scale is None:
        grad_scale = 1.0 / B
    hid_arr = (c_int * max(1, len(hid)))(*hid)
    rc = L.fumi_hip_maml_step(
        ws.handle, _stream(dev), B, N, S, Qn, D, n_hidden, hid_arr, int(T), float(alpha), int(bool(first_order)),
        int(bool(need_grad)), float(grad_scale),
        _f32(x_s, "x_s"), _i64(y_s, "y_s"), _f32(x_q, "x_q"), _i64(y_q, "y_q"), _parr(params, "params"),
        _f32(logits, "logits"), _i64(preds, "preds"), _f32(preds_f, "preds_f"), _f32(loss_b, "loss_b"), _f32(acc_b, "acc_b"),
        _f32(stats, "stats") if stats is not None else None,
        _parr(g_params, "g_params") if need_grad else None)
    _check(rc, "fumi_hip_maml_step")
    return dict(logits=logits, preds=preds, preds_f=preds_f, loss_b=loss_b, acc_b=acc_b, g_params=g_params, stats=stats)


AM3_KEYS = ["Wi", "bi", "G0", "g0", "G1", "g1", "H0", "h0", "H1", "h1"]


def am3_step(ws, x_s, y_s, x_q, y_q, text_s, w, n_way, lamda_fixed=None, *, need_grad=True, grad_scale=None, g_w=None,
             dropout_p=0.0, seed=0, stats=None, want_dx=False):
    """One AM3 step (fumi/models/am3.py:160-200).  w: list of the 10 tensors in AM3_KEYS order.  ``stats``: optional fp32
    [3 + n_way**2] device tensor receiving [loss, correct count, grad_scale * sum of the episodes' mean lamda, confusion
    counts] (the input of ``am3_metrics``)."""
    dev = _dev(x_s)
    L = lib()
    B, S, D = x_s.shape
    Qn = x_q.shape[1]
    if len(w) != 10:
        raise FumiHipError("am3_step: w must hold the 10 tensors of AM3_KEYS")
    P, Ht, Dt = int(w[0].shape[0]), int(w[2].shape[0]), int(w[2].shape[1])

    def check_params():
        for t, shp, k in zip(w, [(P, D), (P,), (Ht, Dt), (Ht,), (P, Ht), (P,), (Ht, P), (Ht,), (1, Ht), (1,)], AM3_KEYS):
            _shape(t, shp, f"w[{k}]")
        if need_grad and g_w is not None:
            for g, t, k in zip(g_w, w, AM3_KEYS):
                _shape(g, t.shape, f"g_w[{k}]")
        return (_parr(w, "w"), _parr(g_w, "g_w") if (need_grad and g_w is not None) else None)

    arrs = _paramset((w, g_w), (D,), check_params) if (need_grad and g_w is not None) else check_params()
    _shape(x_q, (B, Qn, D), "x_q"); _shape(y_s, (B, S), "y_s"); _shape(y_q, (B, Qn), "y_q"); _shape(text_s, (B, S, Dt), "text_s")
    loss = torch.empty(1, device=dev, dtype=torch.float32)
    correct = torch.empty(1, device=dev, dtype=torch.float32)
    preds = torch.empty(B, Qn, device=dev, dtype=torch.int64)
    lam = torch.empty(B, S, device=dev, dtype=torch.float32)
    if need_grad and g_w is None:
        g_w = [torch.empty_like(t) for t in w]
    lf = -1 if lamda_fixed is None else int(lamda_fixed)
    if grad_scale is None:
        grad_scale = 1.0 / B
    args = [ws.handle, _stream(dev), B, n_way, S, Qn, D, Dt, Ht, P, lf, int(bool(need_grad)), float(grad_scale),
            float(dropout_p), int(seed) & 0xFFFFFFFFFFFFFFFF,
            _f32(x_s, "x_s"), _i64(y_s, "y_s"), _f32(x_q, "x_q"), _i64(y_q, "y_q"), _f32(text_s, "text_s"),
            arrs[0], c_void_p(loss.data_ptr()), c_void_p(preds.data_ptr()), c_void_p(lam.data_ptr()), c_void_p(correct.data_ptr()),
            (arrs[1] if arrs[1] is not None else _parr(g_w, "g_w")) if need_grad else None,
            _f32(stats, "stats") if stats is not None else None]
    dx_s = dx_q = None
    if want_dx and need_grad:               # adjoints of the image rows (an encoder in front of the step continues from them)
        dx_s, dx_q = torch.empty_like(x_s), torch.empty_like(x_q)
        _check(L.fumi_hip_am3_step_dx(*args, _f32(dx_s, "dx_s"), _f32(dx_q, "dx_q")), "fumi_hip_am3_step_dx")
    else:
        _check(L.fumi_hip_am3_step(*args), "fumi_hip_am3_step")
    return dict(loss=loss, preds=preds, lamda_s=lam, correct=correct, grads=g_w, stats=stats, dx_s=dx_s, dx_q=dx_q)


def am3_metrics(ws, n_way, stats):
    """[loss, acc, macro F1, macro precision, macro recall, mean lamda] (fp32 [6] on the device) from an ``am3_step`` stats
    tensor (summed over ranks first when sharded)."""
    dev = _dev(stats)
    out = torch.empty(6, device=dev, dtype=torch.float32)
    _check(lib().fumi_hip_am3_metrics(ws.handle, _stream(dev), int(n_way), _f32(stats, "stats"), _f32(out, "out6")),
           "fumi_hip_am3_metrics")
    return out


def glove_bag(ws, tokens, table, pad_id, mode="mean"):
    """WordEmbedding.forward (fumi/models/common.py:23-41): tokens [..., L] int64 -> [..., E]."""
    if mode not in ("mean", "max"):
        raise NameError(f"{mode} pooling strat not defined")           # common.py:41
    dev = _dev(tokens)
    Lseq = tokens.shape[-1]
    R = tokens.numel() // Lseq
    V, E = table.shape
    out = torch.empty(*tokens.shape[:-1], E, device=dev, dtype=torch.float32)
    rc = lib().fumi_hip_glove_bag(ws.handle, _stream(dev), _i64(tokens, "tokens"), R, Lseq, int(pad_id),
                                  _f32(table, "table"), V, E, 0 if mode == "mean" else 1, _f32(out, "out"))
    _check(rc, "fumi_hip_glove_bag")
    return out


def glove_bag_select(ws, tokens_s, y_s, n_way, table, pad_id, mode="mean", defer=False):
    """[B,N,E] pooled text of the first support row of each class (fumi.py:207-210 + common.py:23-41 in one kernel).
    defer: nothing is launched; the request rides in the first launch of the FuMI step that MUST follow on ``ws`` with the
    returned tensor as its ``cls_text`` (fumi_hip_glove_bag_select_deferred)."""
    if mode not in ("mean", "max"):
        raise NameError(f"{mode} pooling strat not defined")
    dev = _dev(tokens_s)
    B, S, Lseq = tokens_s.shape
    V, E = table.shape
    out = torch.empty(B, n_way, E, device=dev, dtype=torch.float32)
    if defer:
        rc = lib().fumi_hip_glove_bag_select_deferred(ws.handle, _i64(tokens_s, "tokens"), _i64(y_s, "y_s"), B, n_way, S, Lseq,
                                                      int(pad_id), _f32(table, "table"), V, E, 0 if mode == "mean" else 1,
                                                      _f32(out, "out"))
        _check(rc, "fumi_hip_glove_bag_select_deferred")
        ws._glove_keep = (tokens_s, y_s, table)       # (the request holds raw pointers until the step has launched it)
        return out
    rc = lib().fumi_hip_glove_bag_select(ws.handle, _stream(dev), _i64(tokens_s, "tokens"), _i64(y_s, "y_s"), B, n_way, S,
                                         Lseq, int(pad_id), _f32(table, "table"), V, E, 0 if mode == "mean" else 1,
                                         _f32(out, "out"))
    _check(rc, "fumi_hip_glove_bag_select")
    return out


def glove_flush(ws, device):
    """Launches a deferred embedding bag that no step has carried."""
    _check(lib().fumi_hip_glove_flush(ws.handle, _stream(device)), "fumi_hip_glove_flush")


def class_text_select(ws, text_s, y_s, n_way):
    dev = _dev(text_s)
    B, S, Dt = text_s.shape
    out = torch.empty(B, n_way, Dt, device=dev, dtype=torch.float32)
    rc = lib().fumi_hip_class_text_select(ws.handle, _stream(dev), B, n_way, S, Dt, _f32(text_s, "text_s"),
                                          _i64(y_s, "y_s"), _f32(out, "out"))
    _check(rc, "fumi_hip_class_text_select")
    return out


def xpanel_fwd(ws, x_s, x_q, W0):
    """A0 [B,S+Qn,h0], G [B,S+Qn,S] (support rows first)."""
    dev = _dev(x_s)
    B, S, D = x_s.shape
    Qn, h0 = x_q.shape[1], W0.shape[0]
    A0 = torch.empty(B, S + Qn, h0, device=dev, dtype=torch.float32)
    G = torch.empty(B, S + Qn, S, device=dev, dtype=torch.float32)
    _check(lib().fumi_hip_xpanel_fwd(ws.handle, _stream(dev), B, S, Qn, D, h0, _f32(x_s, "x_s"), _f32(x_q, "x_q"),
                                     _f32(W0, "W0"), _f32(A0, "A0"), _f32(G, "G")), "fumi_hip_xpanel_fwd")
    return A0, G


def xpanel_bwd(ws, x_s, x_q, Abar, scale=1.0):
    dev = _dev(x_s)
    B, S, D = x_s.shape
    Qn, h0 = x_q.shape[1], Abar.shape[2]
    gW0 = torch.empty(h0, D, device=dev, dtype=torch.float32)
    _check(lib().fumi_hip_xpanel_bwd(ws.handle, _stream(dev), B, S, Qn, D, h0, _f32(x_s, "x_s"), _f32(x_q, "x_q"),
                                     _f32(Abar, "Abar"), float(scale), _f32(gW0, "gW0")), "fumi_hip_xpanel_bwd")
    return gW0


class AdamArgs:
    """Cached pointer tables of one fused Adam call (rebuilt only when a tensor is reallocated)."""

    def __init__(self, params, grads, exp_avg, exp_avg_sq):
        self.key = tuple(t.data_ptr() for t in params + grads + exp_avg + exp_avg_sq)
        self.n = len(params)
        self.p, self.g = _parr(params, "params"), _parr(grads, "grads")
        self.m, self.v = _parr(exp_avg, "exp_avg"), _parr(exp_avg_sq, "exp_avg_sq")
        self.numel = (ctypes.c_long * self.n)(*[t.numel() for t in params])


def adam_step(ws, args, lr, beta1, beta2, eps, weight_decay, step, device):
    rc = lib().fumi_hip_adam_step(ws.handle, _stream(device), args.n, args.p, args.g, args.m, args.v, args.numel,
                                  float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(step))
    _check(rc, "fumi_hip_adam_step")


def adam_step_deferred(ws, args, lr, beta1, beta2, eps, weight_decay, step):
    """The same update, folded into the last launch of the next training meta-step of ``ws`` (fumi_hip_adam_step_deferred)."""
    rc = lib().fumi_hip_adam_step_deferred(ws.handle, args.n, args.p, args.g, args.m, args.v, args.numel,
                                           float(lr), float(beta1), float(beta2), float(eps), float(weight_decay), int(step))
    _check(rc, "fumi_hip_adam_step_deferred")


def adam_flush(ws, device):
    """Launches a deferred Adam step no meta-step has folded; True when it had to."""
    launched = c_int(0)
    _check(lib().fumi_hip_adam_flush(ws.handle, _stream(device), ctypes.byref(launched)), "fumi_hip_adam_flush")
    return bool(launched.value)


def linear_fwd(ws, x, W, b=None, act=0):
    dev = _dev(x)
    M, K = x.shape
    N = W.shape[0]
    y = torch.empty(M, N, device=dev, dtype=torch.float32)
    rc = lib().fumi_hip_linear_fwd(ws.handle, _stream(dev), M, N, K, _f32(x, "x"), _f32(W, "W"),
                                   _f32(b, "b") if b is not None else None, int(act), _f32(y, "y"))
    _check(rc, "fumi_hip_linear_fwd")
    return y


def linear_bwd_data(ws, dy, W):
    dev = _dev(dy)
    M, N = dy.shape
    K = W.shape[1]
    dx = torch.empty(M, K, device=dev, dtype=torch.float32)
    _check(lib().fumi_hip_linear_bwd_data(ws.handle, _stream(dev), M, N, K, _f32(dy, "dy"), _f32(W, "W"), _f32(dx, "dx")),
           "fumi_hip_linear_bwd_data")
    return dx


def linear_bwd_weight(ws, dy, x):
    dev = _dev(dy)
    M, N = dy.shape
    K = x.shape[1]
    dW = torch.empty(N, K, device=dev, dtype=torch.float32)
    db = torch.empty(N, device=dev, dtype=torch.float32)
    _check(lib().fumi_hip_linear_bwd_weight(ws.handle, _stream(dev), M, N, K, _f32(dy, "dy"), _f32(x, "x"),
                                            _f32(dW, "dW"), _f32(db, "db")), "fumi_hip_linear_bwd_weight")
    return dW, db


def sample_episodes(ws, seed, step, B, N, K, Q, class_ptr, class_items):
    """Episode indices on the device (csrc/sampler.hip): classes [B,N], items_s [B,N,K], items_q [B,N,Q] (int64)."""
    dev = _dev(class_ptr)
    C = int(class_ptr.numel()) - 1
    cls = torch.empty(B, N, device=dev, dtype=torch.int64)
    it_s = torch.empty(B, N, K, device=dev, dtype=torch.int64)
    it_q = torch.empty(B, N, Q, device=dev, dtype=torch.int64)
    _check(lib().fumi_hip_sample_episodes(ws.handle, _stream(dev), int(seed) & 0xFFFFFFFFFFFFFFFF, int(step) & 0xFFFFFFFFFFFFFFFF,
                                          B, N, K, Q, C, _i64(class_ptr, "class_ptr"), _i64(class_items, "class_items"),
                                          _i64(cls, "classes"), _i64(it_s, "items_s"), _i64(it_q, "items_q")),
           "fumi_hip_sample_episodes")
    return cls, it_s, it_q


def sample_episodes_tm(ws, seed, step, B, N, K, Q, class_ptr, class_items, fixed_split=True):
    """Episode indices with torchmeta's task semantics: classes [B,N], labels [B,N] (a permutation of 0..N-1 per task), items_s
    [B,N,K], items_q [B,N,Q]; fixed_split: a class tuple always yields the same support / query members."""
    dev = _dev(class_ptr)
    C = int(class_ptr.numel()) - 1
    cls = torch.empty(B, N, device=dev, dtype=torch.int64)
    lab = torch.empty(B, N, device=dev, dtype=torch.int64)
    it_s = torch.empty(B, N, K, device=dev, dtype=torch.int64)
    it_q = torch.empty(B, N, Q, device=dev, dtype=torch.int64)
    _check(lib().fumi_hip_sample_episodes_tm(ws.handle, _stream(dev), int(seed) & 0xFFFFFFFFFFFFFFFF, int(step) & 0xFFFFFFFFFFFFFFFF,
                                             B, N, K, Q, C, _i64(class_ptr, "class_ptr"), _i64(class_items, "class_items"),
                                             int(bool(fixed_split)), _i64(cls, "classes"), _i64(lab, "labels"), _i64(it_s, "items_s"),
                                             _i64(it_q, "items_q")), "fumi_hip_sample_episodes_tm")
    return cls, lab, it_s, it_q


def gather_rows(ws, table, idx):
    """out[i, :] = table[idx[i], :] for a 2-d fp32 / int64 / int32 table on the device (byte copy of whole rows)."""
    dev = _dev(table)
    if table.dim() != 2 or not table.is_contiguous() or table.device != idx.device or idx.dtype != torch.int64:
        raise FumiHipError("gather_rows: table must be a contiguous 2-d device tensor and idx an int64 tensor on the same device")
    row_bytes = table.shape[1] * table.element_size()
    if row_bytes % 4:
        raise FumiHipError("gather_rows: row size must be a multiple of 4 bytes")
    idx = idx.contiguous()
    out = torch.empty(idx.numel(), table.shape[1], device=dev, dtype=table.dtype)
    _check(lib().fumi_hip_gather_rows(ws.handle, _stream(dev), ctypes.c_void_p(table.data_ptr()), table.shape[0], row_bytes,
                                      _i64(idx, "idx"), idx.numel(), ctypes.c_void_p(out.data_ptr())), "fumi_hip_gather_rows")
    return out


def publish_scalars(ws, src, n, host_pinned, seq, defer=False):
    """One tiny launch on the current stream writes src[:n] and then the 64-bit word ``seq`` (byte offset 56) into the pinned
    host tensor ``host_pinned`` (>= 64 bytes) with system-scope stores: the host polls the word, no copy, no event.
    ``defer``: the stores ride on the next ``adam_step`` launch of the workspace instead (``publish_flush`` issues them if
    none comes)."""
    dev = _dev(src)
    if src.dtype != torch.float32 or not src.is_contiguous() or not host_pinned.is_pinned() or host_pinned.numel() * host_pinned.element_size() < 64:
        raise FumiHipError("publish_scalars: src must be contiguous fp32 on the device, host_pinned a pinned tensor of >= 64 bytes")
    if defer:
        _check(lib().fumi_hip_publish_scalars_deferred(ws.handle, ctypes.c_void_p(src.data_ptr()), int(n),
                                                       ctypes.c_void_p(host_pinned.data_ptr()), int(seq)),
               "fumi_hip_publish_scalars_deferred")
        return
    _check(lib().fumi_hip_publish_scalars(ws.handle, _stream(dev), ctypes.c_void_p(src.data_ptr()), int(n),
                                          ctypes.c_void_p(host_pinned.data_ptr()), int(seq)), "fumi_hip_publish_scalars")


def publish_flush(ws, device):
    _check(lib().fumi_hip_publish_flush(ws.handle, _stream(device)), "fumi_hip_publish_flush")



# ---- Conv4 image encoder at the im_net seam (include/fumi_hip.h; csrc/conv4.hip) -------------------------------------------
def conv4_feature_dim(nblk, H, W):
    return int(lib().fumi_hip_conv4_feature_dim(int(nblk), int(H), int(W)))


def _conv4_shapes(x_s, y_s, x_q, y_q, theta):
    if x_s.dim() != 5 or x_q.dim() != 5:
        raise FumiHipError("conv4: images must be [B, rows, Cin, H, W]")
    B, S, Cin, H, W = x_s.shape
    Qn = x_q.shape[1]
    if len(theta) % 3 or not theta:
        raise FumiHipError("conv4: theta must hold (conv weight, BN weight, BN bias) per block")
    nblk = len(theta) // 3
    _shape(x_q, (B, Qn, Cin, H, W), "x_q"); _shape(y_s, (B, S), "y_s"); _shape(y_q, (B, Qn), "y_q")
    for l in range(nblk):
        _shape(theta[3 * l], (64, Cin if l == 0 else 64, 3, 3), f"theta[{3 * l}] (conv weight of block {l})")
        _shape(theta[3 * l + 1], (64,), f"theta[{3 * l + 1}]"); _shape(theta[3 * l + 2], (64,), f"theta[{3 * l + 2}]")
    F = conv4_feature_dim(nblk, H, W)
    if F < 64:
        raise FumiHipError(f"conv4: {H}x{W} images are too small for {nblk} blocks")
    return B, S, Qn, Cin, H, W, nblk, F


CONV4_MAX_TAPED_STEPS = 32


def _conv4_tape_limit(T, need_grad, second_order):
    """A second-order Conv4 step keeps the tape of every inner step resident: at most 32 (csrc/conv4.hip CV_MAXTAPE); evaluation
    (no gradient) and first-order steps reuse one tape and take any T (the reference's test default is 100 steps)."""
    if need_grad and second_order and int(T) > CONV4_MAX_TAPED_STEPS:
        raise FumiHipError(f"conv4: {T} taped inner steps requested, at most {CONV4_MAX_TAPED_STEPS} are supported for a second-order "
                           f"meta-gradient (first-order steps and evaluation take any number)")


def _step_outputs(dev, B, Qn, N):
    return (torch.empty(B, Qn, N, device=dev, dtype=torch.float32), torch.empty(B, Qn, device=dev, dtype=torch.int64),
            torch.empty(B, Qn, device=dev, dtype=torch.float32), torch.empty(B, device=dev, dtype=torch.float32),
            torch.empty(B, device=dev, dtype=torch.float32))


def fumi_conv4_step(ws, n_way, x_s, y_s, x_q, y_q, theta, phi, T, alpha, tanh_head, *, cls_text=None, text_s=None,
                    need_grad=True, grad_scale=None, g_theta=None, g_phi=None, stats=None):
    """FuMI meta-step with the Conv4 encoder (fumi/models/fumi.py:146-192 with im_net = Conv4)."""
    _conv4_tape_limit(T, need_grad, True)
    dev = _dev(x_s)
    B, S, Qn, Cin, H, W, nblk, F = _conv4_shapes(x_s, y_s, x_q, y_q, theta)
    N = int(n_way)
    Ht, Dt = int(phi[0].shape[0]), int(phi[0].shape[1])
    _shape(phi[1], (Ht,), "phi[1]"); _shape(phi[2], (F + 1, Ht), "phi[2]"); _shape(phi[3], (F + 1,), "phi[3]")
    if cls_text is not None:
        _shape(cls_text, (B, N, Dt), "cls_text")
    else:
        _shape(text_s, (B, S, Dt), "text_s")
    logits, preds, preds_f, loss_b, acc_b = _step_outputs(dev, B, Qn, N)
    if need_grad:
        g_theta = [torch.empty_like(t) for t in theta] if g_theta is None else g_theta
        g_phi = [torch.empty_like(t) for t in phi] if g_phi is None else g_phi
    rc = lib().fumi_hip_fumi_conv4_step(
        ws.handle, _stream(dev), B, N, S, Qn, Cin, H, W, nblk, Dt, Ht, int(T), float(alpha), int(bool(tanh_head)),
        int(bool(need_grad)), float(1.0 / B if grad_scale is None else grad_scale),
        _f32(x_s, "x_s"), _i64(y_s, "y_s"), _f32(x_q, "x_q"), _i64(y_q, "y_q"),
        _f32(cls_text, "cls_text") if cls_text is not None else None, _f32(text_s, "text_s") if text_s is not None else None,
        _parr(theta, "theta"), _parr(phi, "phi"),
        _f32(logits, "logits"), _i64(preds, "preds"), _f32(preds_f, "preds_f"), _f32(loss_b, "loss_b"), _f32(acc_b, "acc_b"),
        _f32(stats, "stats") if stats is not None else None,
        _parr(g_theta, "g_theta") if need_grad else None, _parr(g_phi, "g_phi") if need_grad else None)
    _check(rc, "fumi_hip_fumi_conv4_step")
    return dict(logits=logits, preds=preds, preds_f=preds_f, loss_b=loss_b, acc_b=acc_b, g_theta=g_theta, g_phi=g_phi, stats=stats)


def maml_conv4_step(ws, x_s, y_s, x_q, y_q, params, T, alpha, first_order=False, *, need_grad=True, grad_scale=None,
                    g_params=None, stats=None):
    """MAML meta-step with the Conv4 encoder: params = theta (3 per block) + [lin_final W [N,F], b [N]]."""
    _conv4_tape_limit(T, need_grad, not first_order)
    dev = _dev(x_s)
    B, S, Qn, Cin, H, W, nblk, F = _conv4_shapes(x_s, y_s, x_q, y_q, params[:-2])
    N = int(params[-2].shape[0])
    _shape(params[-2], (N, F), "lin_final.weight"); _shape(params[-1], (N,), "lin_final.bias")
    logits, preds, preds_f, loss_b, acc_b = _step_outputs(dev, B, Qn, N)
    if need_grad and g_params is None:
        g_params = [torch.empty_like(t) for t in params]
    rc = lib().fumi_hip_maml_conv4_step(
        ws.handle, _stream(dev), B, N, S, Qn, Cin, H, W, nblk, int(T), float(alpha), int(bool(first_order)),
        int(bool(need_grad)), float(1.0 / B if grad_scale is None else grad_scale),
        _f32(x_s, "x_s"), _i64(y_s, "y_s"), _f32(x_q, "x_q"), _i64(y_q, "y_q"), _parr(params, "params"),
        _f32(logits, "logits"), _i64(preds, "preds"), _f32(preds_f, "preds_f"), _f32(loss_b, "loss_b"), _f32(acc_b, "acc_b"),
        _f32(stats, "stats") if stats is not None else None, _parr(g_params, "g_params") if need_grad else None)
    _check(rc, "fumi_hip_maml_conv4_step")
    return dict(logits=logits, preds=preds, preds_f=preds_f, loss_b=loss_b, acc_b=acc_b, g_params=g_params, stats=stats)


# ---- ResNet-12 (bf16) -----------------------------------------------------------------------------------------------
def _resnet12_shapes(x_s, y_s, x_q, y_q, theta):
    if x_s.dim() != 5 or x_q.dim() != 5:
        raise FumiHipError("resnet12: images must be [B, rows, Cin, H, W]")
    B, S, Cin, H, W = x_s.shape
    Qn = x_q.shape[1]
    if len(theta) % 12 or not theta:
        raise FumiHipError("resnet12: theta must hold 12 tensors per block (W1,g1,b1, W2,g2,b2, W3,g3,b3, Ws,gs,bs)")
    nblk = len(theta) // 12
    _shape(x_q, (B, Qn, Cin, H, W), "x_q"); _shape(y_s, (B, S), "y_s"); _shape(y_q, (B, Qn), "y_q")
    channels, ci = [], Cin
    for l in range(nblk):
        c = int(theta[12 * l].shape[0])
        if c % 32:
            raise FumiHipError("resnet12: channel counts must be multiples of 32")
        for k, (cin, ks) in enumerate(((ci, 3), (c, 3), (c, 3), (ci, 1))):
            _shape(theta[12 * l + 3 * k], (c, cin, ks, ks), f"theta[{12 * l + 3 * k}]")
            _shape(theta[12 * l + 3 * k + 1], (c,), f"theta[{12 * l + 3 * k + 1}]")
            _shape(theta[12 * l + 3 * k + 2], (c,), f"theta[{12 * l + 3 * k + 2}]")
        channels.append(c); ci = c
    if min(H, W) >> nblk < 1:
        raise FumiHipError(f"resnet12: {H}x{W} images are too small for {nblk} blocks")
    return B, S, Qn, Cin, H, W, nblk, channels


def _ci(v):
    return (c_int * len(v))(*[int(x) for x in v])


def resnet12_set_budget(gigabytes):
    """Workspace budget (GB) from which the episode chunk of the ResNet-12 steps is derived (0: default 200)."""
    _check(lib().fumi_hip_resnet12_set_budget(float(gigabytes)), "fumi_hip_resnet12_set_budget")


def fumi_resnet12_step(ws, n_way, x_s, y_s, x_q, y_q, theta, phi, T, alpha, tanh_head, *, cls_text=None, text_s=None,
                       need_grad=True, grad_scale=None, g_theta=None, g_phi=None, stats=None, chunk=0):
    """FuMI meta-step with the bf16 ResNet-12 encoder (fumi/models/fumi.py:146-192 with im_net = ResNet-12)."""
    dev = _dev(x_s)
    B, S, Qn, Cin, H, W, nblk, channels = _resnet12_shapes(x_s, y_s, x_q, y_q, theta)
    N, F = int(n_way), channels[-1]
    Ht, Dt = int(phi[0].shape[0]), int(phi[0].shape[1])
    _shape(phi[1], (Ht,), "phi[1]"); _shape(phi[2], (F + 1, Ht), "phi[2]"); _shape(phi[3], (F + 1,), "phi[3]")
    if cls_text is not None:
        _shape(cls_text, (B, N, Dt), "cls_text")
    else:
        _shape(text_s, (B, S, Dt), "text_s")
    logits, preds, preds_f, loss_b, acc_b = _step_outputs(dev, B, Qn, N)
    if need_grad:
        g_theta = [torch.empty_like(t) for t in theta] if g_theta is None else g_theta
        g_phi = [torch.empty_like(t) for t in phi] if g_phi is None else g_phi
    rc = lib().fumi_hip_fumi_resnet12_step(
        ws.handle, _stream(dev), B, N, S, Qn, Cin, H, W, nblk, _ci(channels), Dt, Ht, int(T), float(alpha), int(bool(tanh_head)),
        int(bool(need_grad)), float(1.0 / B if grad_scale is None else grad_scale), int(chunk),
        _f32(x_s, "x_s"), _i64(y_s, "y_s"), _f32(x_q, "x_q"), _i64(y_q, "y_q"),
        _f32(cls_text, "cls_text") if cls_text is not None else None, _f32(text_s, "text_s") if text_s is not None else None,
        _parr(theta, "theta"), _parr(phi, "phi"),
        _f32(logits, "logits"), _i64(preds, "preds"), _f32(preds_f, "preds_f"), _f32(loss_b, "loss_b"), _f32(acc_b, "acc_b"),
        _f32(stats, "stats") if stats is not None else None,
        _parr(g_theta, "g_theta") if need_grad else None, _parr(g_phi, "g_phi") if need_grad else None)
    _check(rc, "fumi_hip_fumi_resnet12_step")
    return dict(logits=logits, preds=preds, preds_f=preds_f, loss_b=loss_b, acc_b=acc_b, g_theta=g_theta, g_phi=g_phi, stats=stats)


def maml_resnet12_step(ws, x_s, y_s, x_q, y_q, params, T, alpha, first_order=False, *, need_grad=True, grad_scale=None,
                       g_params=None, stats=None, chunk=0):
    """MAML meta-step with the bf16 ResNet-12 encoder: params = theta (12 per block) + [lin_final W [N,F], b [N]]."""
    dev = _dev(x_s)
    B, S, Qn, Cin, H, W, nblk, channels = _resnet12_shapes(x_s, y_s, x_q, y_q, params[:-2])
    N, F = int(params[-2].shape[0]), channels[-1]
    _shape(params[-2], (N, F), "lin_final.weight"); _shape(params[-1], (N,), "lin_final.bias")
    logits, preds, preds_f, loss_b, acc_b = _step_outputs(dev, B, Qn, N)
    if need_grad:
        g_params = [torch.empty_like(t) for t in params] if g_params is None else g_params
    rc = lib().fumi_hip_maml_resnet12_step(
        ws.handle, _stream(dev), B, N, S, Qn, Cin, H, W, nblk, _ci(channels), int(T), float(alpha), int(bool(first_order)),
        int(bool(need_grad)), float(1.0 / B if grad_scale is None else grad_scale), int(chunk),
        _f32(x_s, "x_s"), _i64(y_s, "y_s"), _f32(x_q, "x_q"), _i64(y_q, "y_q"), _parr(params, "params"),
        _f32(logits, "logits"), _i64(preds, "preds"), _f32(preds_f, "preds_f"), _f32(loss_b, "loss_b"), _f32(acc_b, "acc_b"),
        _f32(stats, "stats") if stats is not None else None, _parr(g_params, "g_params") if need_grad else None)
    _check(rc, "fumi_hip_maml_resnet12_step")
    return dict(logits=logits, preds=preds, preds_f=preds_f, loss_b=loss_b, acc_b=acc_b, g_params=g_params, stats=stats)


def resnet12_features(ws, x, theta):
    """feats [G, M, F] = ResNet12(x [G, M, Cin, H, W]); batch statistics per group of M images."""
    dev = _dev(x)
    G, M, Cin, H, W = x.shape
    nblk = len(theta) // 12
    channels = [int(theta[12 * l].shape[0]) for l in range(nblk)]
    feats = torch.empty(G, M, channels[-1], device=dev, dtype=torch.float32)
    _check(lib().fumi_hip_resnet12_features(ws.handle, _stream(dev), G, M, Cin, H, W, nblk, _ci(channels), _f32(x, "x"),
                                            _parr(theta, "theta"), _f32(feats, "feats")), "fumi_hip_resnet12_features")
    return feats


def _bf16ptr(t, name):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.bfloat16 and t.is_contiguous()):
        raise FumiHipError(f"{name}: expected a contiguous bfloat16 GPU tensor")
    return c_void_p(t.data_ptr())


def rn12_conv(ws, x, Wt, H, W, transpose=False, want_stats=False):
    """Unit op: x [B, M*(H+2)*(W+2), Cx] bf16 padded channels-last, Wt [B, Cout, Cin, k, k] fp32 -> y bf16 (and [B,2,Cy] stats)."""
    dev = _dev(x)
    B, npix, Cx = x.shape
    Cout, Cin, k = int(Wt.shape[1]), int(Wt.shape[2]), int(Wt.shape[3])
    M = npix // ((H + 2) * (W + 2))
    Cy = Cin if transpose else Cout
    y = torch.empty(B, npix, Cy, device=dev, dtype=torch.bfloat16)
    st = torch.empty(B, 2, Cy, device=dev, dtype=torch.float32) if want_stats else None
    _check(lib().fumi_hip_rn12_conv(ws.handle, _stream(dev), B, M, H, W, Cin, Cout, k * k, int(bool(transpose)), _bf16ptr(x, "x"),
                                    _f32(Wt, "Wt"), _bf16ptr(y, "y"), _f32(st, "stats") if st is not None else None), "fumi_hip_rn12_conv")
    return (y, st) if want_stats else y


def rn12_wgrad(ws, x, dy, H, W, k):
    dev = _dev(x)
    B, npix, Cin = x.shape
    Cout = int(dy.shape[2])
    M = npix // ((H + 2) * (W + 2))
    dW = torch.empty(B, Cout, Cin, k, k, device=dev, dtype=torch.float32)
    _check(lib().fumi_hip_rn12_wgrad(ws.handle, _stream(dev), B, M, H, W, Cin, Cout, k * k, _bf16ptr(x, "x"), _bf16ptr(dy, "dy"),
                                     _f32(dW, "dW")), "fumi_hip_rn12_wgrad")
    return dW


def resnet12_set_option(key, value):
    """fumi_hip_resnet12_set_option: key 0 = probe mode (test hook), key 1 = the reverse sweep stops after inner step `value`."""
    _check(lib().fumi_hip_resnet12_set_option(int(key), int(value)), "fumi_hip_resnet12_set_option")


def rn12_probe(ws, device, pass_, kind, block=0, idx=0):
    """Test hook: one stored intermediate of the last single-chunk ResNet-12 step run in probe mode, flat (bf16 maps as
    torch.bfloat16, everything else fp32) -- fumi_hip_rn12_probe."""
    n, bf = c_size_t(0), c_int(0)
    dummy = torch.empty(4, device=device, dtype=torch.float32)
    args = (ws.handle, _stream(device), int(pass_), int(kind), int(block), int(idx))
    _check(lib().fumi_hip_rn12_probe(*args, c_void_p(dummy.data_ptr()), 0, ctypes.byref(n), ctypes.byref(bf)), "fumi_hip_rn12_probe")
    out = torch.empty(n.value // (2 if bf.value else 4), device=device, dtype=torch.bfloat16 if bf.value else torch.float32)
    _check(lib().fumi_hip_rn12_probe(*args, c_void_p(out.data_ptr()), n.value, ctypes.byref(n), ctypes.byref(bf)), "fumi_hip_rn12_probe")
    return out


def conv4_set_option(key, value):
    """fumi_hip_conv4_set_option: key 0 = fused block 1 (default 1)."""
    _check(lib().fumi_hip_conv4_set_option(int(key), int(value)), "fumi_hip_conv4_set_option")


def conv4_features(ws, x, theta):
    """[G, M, F] = Conv4(x [G, M, Cin, H, W]) with the batch statistics of each group of M images (forward only)."""
    dev = _dev(x)
    if x.dim() != 5:
        raise FumiHipError("conv4_features: x must be [groups, images, Cin, H, W]")
    G, M, Cin, H, W = x.shape
    nblk = len(theta) // 3
    F = conv4_feature_dim(nblk, H, W)
    out = torch.empty(G, M, F, device=dev, dtype=torch.float32)
    _check(lib().fumi_hip_conv4_features(ws.handle, _stream(dev), G, M, Cin, H, W, nblk, _f32(x, "x"), _parr(theta, "theta"),
                                         _f32(out, "feats")), "fumi_hip_conv4_features")
    return out


def conv4_encode(ws, x_s, x_q, theta, keep_tape=False):
    """(feats_s [B,S,F], feats_q [B,Qn,F]) = Conv4 of every episode's support / query images (one batch-statistics group each).
    keep_tape: the activations stay laid out in ``ws`` for ``conv4_encode_bwd`` -- no other call may use ``ws`` in between."""
    dev = _dev(x_s)
    if x_s.dim() != 5 or x_q.dim() != 5 or x_s.shape[0] != x_q.shape[0] or x_s.shape[2:] != x_q.shape[2:]:
        raise FumiHipError("conv4_encode: x_s [B,S,C,H,W] and x_q [B,Qn,C,H,W] expected")
    B, S, Cin, H, W = x_s.shape
    Qn = x_q.shape[1]
    nblk = len(theta) // 3
    F = conv4_feature_dim(nblk, H, W)
    fs = torch.empty(B, S, F, device=dev, dtype=torch.float32)
    fq = torch.empty(B, Qn, F, device=dev, dtype=torch.float32)
    _check(lib().fumi_hip_conv4_encode(ws.handle, _stream(dev), B, S, Qn, Cin, H, W, nblk, _f32(x_s, "x_s"), _f32(x_q, "x_q"),
                                       _parr(theta, "theta"), _f32(fs, "feats_s"), _f32(fq, "feats_q"), int(bool(keep_tape))),
           "fumi_hip_conv4_encode")
    return fs, fq


def conv4_encode_bwd(ws, x_s, x_q, dfeats_s, dfeats_q, theta_like, scale=1.0, g_theta=None):
    """Gradient of sum <dfeats, Conv4(x)> w.r.t. the encoder's parameters (summed over episodes, times ``scale``) from the tape the
    last ``conv4_encode(..., keep_tape=True)`` left in ``ws``."""
    dev = _dev(x_s)
    B, S, Cin, H, W = x_s.shape
    Qn = x_q.shape[1]
    nblk = len(theta_like) // 3
    F = conv4_feature_dim(nblk, H, W)
    _shape(dfeats_s, (B, S, F), "dfeats_s"); _shape(dfeats_q, (B, Qn, F), "dfeats_q")
    if g_theta is None:
        g_theta = [torch.empty_like(t) for t in theta_like]
    _check(lib().fumi_hip_conv4_encode_bwd(ws.handle, _stream(dev), B, S, Qn, Cin, H, W, nblk, _f32(x_s, "x_s"), _f32(x_q, "x_q"),
                                           _f32(dfeats_s, "dfeats_s"), _f32(dfeats_q, "dfeats_q"), float(scale),
                                           _parr(g_theta, "g_theta")), "fumi_hip_conv4_encode_bwd")
    return g_theta


def conv4_probe(ws, device, pass_, kind, block=0):
    """Test hook: one intermediate tensor of the last conv4 step as a flat fp32 tensor (fumi_hip_conv4_probe)."""
    n = c_size_t(0)
    dummy = torch.empty(1, device=device, dtype=torch.float32)
    _check(lib().fumi_hip_conv4_probe(ws.handle, _stream(device), int(pass_), int(kind), int(block), _f32(dummy, "out"), 0,
                                      ctypes.byref(n)), "fumi_hip_conv4_probe")
    out = torch.empty(n.value, device=device, dtype=torch.float32)
    _check(lib().fumi_hip_conv4_probe(ws.handle, _stream(device), int(pass_), int(kind), int(block), _f32(out, "out"), n.value,
                                      ctypes.byref(n)), "fumi_hip_conv4_probe")
    return out


def _conv3x3(fn, name, ws, a, b, out_shape):
    dev = _dev(a)
    M, H, W, C = a.shape
    if C != 64:
        raise FumiHipError(f"{name}: channels-last tensors with 64 channels expected")
    out = torch.empty(out_shape, device=dev, dtype=torch.float32)
    _check(fn(ws.handle, _stream(dev), M, H, W, _f32(a, "a"), _f32(b, "b"), _f32(out, "out")), name)
    return out


def conv3x3_fwd(ws, x, Wt):
    """y [M,H,W,64] = conv3x3(x [M,H,W,64], Wt [64,64,3,3]), pad 1 (channels-last)."""
    _shape(Wt, (64, 64, 3, 3), "Wt")
    return _conv3x3(lib().fumi_hip_conv3x3_fwd, "fumi_hip_conv3x3_fwd", ws, x, Wt, x.shape)


def conv3x3_bwd_data(ws, dy, Wt):
    _shape(Wt, (64, 64, 3, 3), "Wt")
    return _conv3x3(lib().fumi_hip_conv3x3_bwd_data, "fumi_hip_conv3x3_bwd_data", ws, dy, Wt, dy.shape)


def conv3x3_bwd_weight(ws, x, dy):
    _shape(dy, x.shape, "dy")
    return _conv3x3(lib().fumi_hip_conv3x3_bwd_weight, "fumi_hip_conv3x3_bwd_weight", ws, x, dy, (64, 64, 3, 3))


def sgd_axpy(ws, p, step_size, g, out=None):
    dev = _dev(p)
    out = torch.empty_like(p) if out is None else out
    _check(lib().fumi_hip_sgd_axpy(ws.handle, _stream(dev), p.numel(), _f32(p, "p"), float(step_size), _f32(g, "g"), _f32(out, "out")),
           "fumi_hip_sgd_axpy")
    return out


def ce_fwd_bwd(ws, z, y):
    """(mean cross-entropy [1], dz [M,N], first arg-max [M]) of logits z [M,N] and labels y [M]."""
    dev = _dev(z)
    M, N = z.shape
    loss = torch.empty(1, device=dev, dtype=torch.float32)
    dz = torch.empty_like(z)
    preds = torch.empty(M, device=dev, dtype=torch.int64)
    _check(lib().fumi_hip_ce_fwd_bwd(ws.handle, _stream(dev), M, N, _f32(z, "z"), _i64(y, "y"), _f32(loss, "loss"), _f32(dz, "dz"),
                                     _i64(preds, "preds")), "fumi_hip_ce_fwd_bwd")
    return loss, dz, preds


def proto_reduce(ws, x, y, n_way):
    """Per-class means [B,N,P] of x [B,S,P] (fumi/utils/utils.py:331-376, counts clamped to >= 1)."""
    dev = _dev(x)
    B, S, P = x.shape
    out = torch.empty(B, n_way, P, device=dev, dtype=torch.float32)
    _check(lib().fumi_hip_proto_reduce(ws.handle, _stream(dev), B, S, int(n_way), P, _f32(x, "x"), _i64(y, "y"), _f32(out, "out")),
           "fumi_hip_proto_reduce")
    return out


CLIP_KEYS = ["text_fc.weight", "text_fc.bias", "text_fc2.weight", "text_fc2.bias",
             "image_fc.weight", "image_fc.bias", "image_fc2.weight", "image_fc2.bias"]


def clip_step(ws, text, image, w, *, need_loss=True, need_grad=True, g_w=None):
    """CLIP baseline (fumi/models/clip.py): sim [nt, ni]; with need_loss the symmetric cross-entropy [1] and, with need_grad, its
    gradients w.r.t. the eight tensors of CLIP_KEYS order."""
    dev = _dev(text)
    nt, Dt = text.shape
    ni, D = image.shape
    P = int(w[0].shape[0])
    for t, shp, k in zip(w, [(P, Dt), (P,), (P, P), (P,), (P, D), (P,), (P, P), (P,)], CLIP_KEYS):
        _shape(t, shp, k)
    sim = torch.empty(nt, ni, device=dev, dtype=torch.float32)
    loss = torch.empty(1, device=dev, dtype=torch.float32) if need_loss else None
    if need_grad and g_w is None:
        g_w = [torch.empty_like(t) for t in w]
    _check(lib().fumi_hip_clip_step(ws.handle, _stream(dev), nt, ni, Dt, D, P, _f32(text, "text"), _f32(image, "image"), _parr(w, "w"),
                                    int(bool(need_grad)), _f32(sim, "sim"), _f32(loss, "loss") if need_loss else None,
                                    _parr(g_w, "g_w") if need_grad else None), "fumi_hip_clip_step")
    return dict(sim=sim, loss=loss, grads=g_w if need_grad else None)


def lstm_bidir(ws, tokens, table, lstm_w, pad_id, use_cell):
    """[..., 2H] final states of a bidirectional LSTM over the non-PAD prefix of token rows [..., L] (common.py:44-161)."""
    dev = _dev(tokens)
    L = tokens.shape[-1]
    R = tokens.numel() // L
    V, E = table.shape
    H = int(lstm_w[1].shape[1])
    for d in range(2):
        _shape(lstm_w[4 * d], (4 * H, E), "weight_ih"); _shape(lstm_w[4 * d + 1], (4 * H, H), "weight_hh")
        _shape(lstm_w[4 * d + 2], (4 * H,), "bias_ih"); _shape(lstm_w[4 * d + 3], (4 * H,), "bias_hh")
    out = torch.empty(*tokens.shape[:-1], 2 * H, device=dev, dtype=torch.float32)
    _check(lib().fumi_hip_lstm_bidir(ws.handle, _stream(dev), R, L, E, H, _i64(tokens, "tokens"), int(pad_id), _f32(table, "table"), V,
                                     _parr(lstm_w, "lstm_w"), int(bool(use_cell)), _f32(out, "out")), "fumi_hip_lstm_bidir")
    return out


def _lstm_dims(tokens, table, lstm_w):
    L = tokens.shape[-1]
    R = tokens.numel() // L
    V, E = table.shape
    H = int(lstm_w[1].shape[1])
    for d in range(2):
        _shape(lstm_w[4 * d], (4 * H, E), "weight_ih"); _shape(lstm_w[4 * d + 1], (4 * H, H), "weight_hh")
        _shape(lstm_w[4 * d + 2], (4 * H,), "bias_ih"); _shape(lstm_w[4 * d + 3], (4 * H,), "bias_hh")
    return R, L, V, E, H


def lstm_bidir_train(ws, tokens, table, lstm_w, pad_id, use_cell):
    """(out [..., 2H], tape): lstm_bidir keeping what lstm_bidir_bwd needs (--fine_tune with RNN / RNNhid, fumi.py:65-67)."""
    dev = _dev(tokens)
    R, L, V, E, H = _lstm_dims(tokens, table, lstm_w)
    out = torch.empty(*tokens.shape[:-1], 2 * H, device=dev, dtype=torch.float32)
    tape = torch.empty(int(lib().fumi_hip_lstm_tape_floats(R, L, E, H)), device=dev, dtype=torch.float32)
    _check(lib().fumi_hip_lstm_bidir_train(ws.handle, _stream(dev), R, L, E, H, _i64(tokens, "tokens"), int(pad_id), _f32(table, "table"),
                                           V, _parr(lstm_w, "lstm_w"), int(bool(use_cell)), _f32(out, "out"), _f32(tape, "tape")),
           "fumi_hip_lstm_bidir_train")
    return out, tape


def lstm_bidir_bwd(ws, tokens, table, lstm_w, pad_id, use_cell, tape, d_out):
    """Gradients of the 8 LSTM tensors for the output adjoint d_out [..., 2H] of the lstm_bidir_train call that wrote `tape`."""
    dev = _dev(tokens)
    R, L, V, E, H = _lstm_dims(tokens, table, lstm_w)
    if tape.numel() != int(lib().fumi_hip_lstm_tape_floats(R, L, E, H)):
        raise ValueError("tape: not the tape of a lstm_bidir_train call of these shapes")
    if d_out.numel() != R * 2 * H:
        raise ValueError(f"d_out: expected {R * 2 * H} elements, got {d_out.numel()}")
    g_w = [torch.empty_like(t) for t in lstm_w]
    _check(lib().fumi_hip_lstm_bidir_bwd(ws.handle, _stream(dev), R, L, E, H, _i64(tokens, "tokens"), int(pad_id), _parr(lstm_w, "lstm_w"),
                                         int(bool(use_cell)), _f32(tape, "tape"), _f32(d_out, "d_out"), _parr(g_w, "g_w")),
           "fumi_hip_lstm_bidir_bwd")
    return g_w


def want_text_grad(ws, g_cls_text):
    """Arm the next fumi_step(need_grad=True) on ws to write d loss / d class text rows into g_cls_text [B*N, Dt] (None disarms)."""
    _check(lib().fumi_hip_want_text_grad(ws.handle, None if g_cls_text is None else _f32(g_cls_text, "g_cls_text")),
           "fumi_hip_want_text_grad")
