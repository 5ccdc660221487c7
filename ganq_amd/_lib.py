"""ctypes binding of libganq_hip.so (C-ABI declared in include/ganq_hip.h).

The library is the product: there is no CPU or PyTorch fallback.  If the shared object is
missing, fails to load or lacks a symbol, importing a compute entry point raises
``GanqHipError`` -- loudly, by design.

Tensors are torch-owned; this layer only borrows ``data_ptr()`` for the duration of a call
and passes the current torch stream of the tensors' device, with that device made current (reference boundary: the tensors that
``GANQ._perform_quantization_loop`` receives, gptqmodel/quantization/ganq.py:456).
"""
import ctypes
import os
import subprocess
import threading

import torch

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GANQ_HIP_LIB") or os.path.join(_PKG_DIR, "libganq_hip.so")  # override: developer builds
CSRC_DIR = os.path.join(_PKG_DIR, "csrc")

FLAG_ALIAS_Q = 1
FLAG_NO_HELPERS = 2

_c_i64 = ctypes.c_int64
_c_vp = ctypes.c_void_p
_c_sz = ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/ganq_hip.h one to one
SIGNATURES = {
    "ganq_hip_version": (ctypes.c_int, []),
    "ganq_hip_last_error": (ctypes.c_char_p, []),
    "ganq_hip_selftest": (ctypes.c_int, [_c_vp]),
    "ganq_debug_div_check": (ctypes.c_int, [ctypes.c_uint64, ctypes.c_uint32, _c_vp, _c_vp, _c_vp]),
    "ganq_debug_wh_product": (ctypes.c_int, [_c_vp, _c_vp, _c_i64, _c_i64, _c_vp, _c_vp, _c_vp]),
    "ganq_debug_gemm_h16": (ctypes.c_int, [_c_vp, _c_vp, _c_vp, _c_vp, ctypes.c_int, _c_i64, _c_i64, _c_i64, _c_vp, _c_vp]),
    "ganq_hessian_workspace_bytes": (_c_sz, [_c_i64, _c_i64]),
    "ganq_hessian_accum": (ctypes.c_int, [_c_vp, _c_vp, ctypes.c_int, _c_i64, _c_i64, _c_i64, _c_i64, _c_vp, _c_sz, _c_vp]),
    "ganq_debug_hessian_t_cut": (ctypes.c_int, [_c_i64, _c_i64, ctypes.c_int, _c_vp, _c_vp, ctypes.c_int, _c_vp, ctypes.c_int]),
    "ganq_hessian_t_supported": (ctypes.c_int, [_c_i64, _c_i64]),
    "ganq_hessian_t_workspace_bytes": (_c_sz, [_c_i64]),
    "ganq_hessian_stage_t": (ctypes.c_int, [_c_vp, _c_i64, _c_vp, _c_i64, _c_i64, _c_i64, _c_vp]),
    "ganq_hessian_accum_t": (ctypes.c_int, [_c_vp, _c_vp, _c_i64, ctypes.c_int, _c_i64, _c_i64, _c_i64, _c_i64, _c_vp, _c_sz, _c_vp]),
    "ganq_cholesky_workspace_bytes": (_c_sz, [_c_i64]),
    "ganq_cholesky": (ctypes.c_int, [_c_vp, _c_i64, _c_i64, _c_vp, _c_vp, _c_sz, _c_vp]),
    "ganq_prologue_rowstats": (ctypes.c_int, [_c_vp, _c_i64, _c_vp, _c_vp, _c_vp]),
    "ganq_prologue_gather": (ctypes.c_int, [_c_vp, _c_vp, _c_vp, _c_i64, _c_vp, _c_vp, ctypes.c_int, _c_vp, _c_vp, ctypes.c_int, _c_vp,
                                            _c_vp, ctypes.c_int, _c_vp]),
    "ganq_prologue_weights": (ctypes.c_int, [_c_vp, _c_vp, _c_vp, _c_i64, _c_i64, ctypes.c_int, _c_vp, _c_vp]),
    "ganq_kmeans_workspace_bytes": (_c_sz, [_c_i64, _c_i64, ctypes.c_int]),
    "ganq_kmeans_init": (ctypes.c_int, [_c_vp, _c_vp, _c_i64, _c_i64, ctypes.c_int, _c_vp, _c_vp, _c_sz, _c_vp]),
    "ganq_solve_s_workspace_bytes": (_c_sz, [_c_i64, _c_i64, ctypes.c_int]),
    "ganq_solve_s": (ctypes.c_int, [_c_vp, _c_vp, _c_i64, _c_vp, _c_i64, _c_i64, ctypes.c_int, _c_vp, _c_vp, _c_vp,
                                    _c_sz, _c_vp]),
    "ganq_matmul_f32": (ctypes.c_int, [_c_vp, _c_vp, _c_i64, _c_i64, _c_i64, _c_vp, _c_vp]),
    "ganq_update_t_workspace_bytes": (_c_sz, [_c_i64, _c_i64, ctypes.c_int]),
    "ganq_update_t": (ctypes.c_int, [_c_vp, _c_vp, _c_vp, _c_i64, _c_i64, ctypes.c_int, ctypes.c_double, _c_vp, _c_vp,
                                     _c_vp, _c_vp, _c_sz, _c_vp]),
    "ganq_quad_loss_workspace_bytes": (_c_sz, [_c_i64, _c_i64, ctypes.c_int]),
    "ganq_quad_loss": (ctypes.c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, ctypes.c_int, _c_vp, _c_vp, _c_sz,
                                      _c_vp]),
    "ganq_dequant_losses": (ctypes.c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_i64, _c_i64, ctypes.c_int, _c_vp, _c_vp,
                                           _c_vp]),
    "ganq_run_layer_workspace_bytes": (_c_sz, [_c_i64, _c_i64, ctypes.c_int]),
    "ganq_run_layer": (ctypes.c_int, [_c_vp, _c_vp, _c_vp, _c_i64, _c_vp, _c_i64, _c_i64, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_uint32, ctypes.c_double, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_sz,
                                      _c_vp]),
    "ganq_run_layer_rows": (ctypes.c_int, [_c_vp, _c_vp, _c_vp, _c_i64, _c_vp, _c_i64, _c_i64, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_uint32, ctypes.c_double, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp,
                                           _c_vp, _c_sz, _c_vp]),
    "ganq_select_best": (ctypes.c_int, [_c_vp, _c_i64, ctypes.c_int, _c_vp, _c_vp, _c_vp]),
    "ganq_debug_set_option": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_longlong]),
    "ganq_debug_get_option": (ctypes.c_int, [ctypes.c_char_p, ctypes.POINTER(ctypes.c_longlong)]),
    "ganq_debug_reset_option": (ctypes.c_int, [ctypes.c_char_p]),
    "ganq_lut_linear_workspace_bytes": (_c_sz, [_c_i64, _c_i64, _c_i64, ctypes.c_int]),
    "ganq_lut_linear_workspace_init": (ctypes.c_int, [_c_vp, _c_sz, _c_vp]),
    "ganq_lut_linear_fwd": (ctypes.c_int, [_c_vp, _c_vp, _c_vp, _c_vp, ctypes.c_int, _c_i64, _c_i64, _c_i64,
                                           ctypes.c_int, _c_vp, _c_vp, _c_sz, _c_vp]),
    "ganq_lut_linear_fwd_add": (ctypes.c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, ctypes.c_int, _c_i64, _c_i64, _c_i64,
                                               ctypes.c_int, _c_vp, _c_vp, _c_sz, _c_vp]),
    "ganq_lut_linear_outliers_workspace_bytes": (_c_sz, [_c_i64, _c_i64, _c_i64, ctypes.c_int]),
    "ganq_lut_linear_fwd_outliers": (ctypes.c_int, [_c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp, ctypes.c_int, _c_i64, _c_i64,
                                                    _c_i64, ctypes.c_int, _c_vp, _c_vp, _c_sz, _c_vp]),
    "ganq_lut_dequant": (ctypes.c_int, [_c_vp, _c_vp, ctypes.c_int, _c_i64, _c_i64, ctypes.c_int, _c_vp, _c_vp]),
    "ganq_outlier_cutoffs": (ctypes.c_int, [_c_vp, _c_i64, _c_i64, ctypes.c_double, _c_vp, _c_vp, _c_vp, _c_vp]),
    "ganq_outlier_extract": (ctypes.c_int, [_c_vp, _c_i64, _c_i64, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp]),
    "ganq_outlier_matmul": (ctypes.c_int, [_c_vp, ctypes.c_int, _c_i64, _c_i64, _c_i64, _c_vp, _c_vp, _c_vp, _c_vp, _c_vp]),
    "ganq_pack_indices": (ctypes.c_int, [_c_vp, _c_i64, _c_i64, ctypes.c_int, _c_vp, _c_vp]),
    "ganq_unpack_indices": (ctypes.c_int, [_c_vp, _c_i64, _c_i64, ctypes.c_int, _c_vp, _c_vp]),
    "ganq_profile_enable": (ctypes.c_int, [ctypes.c_int]),
    "ganq_profile_reset": (ctypes.c_int, []),
    "ganq_profile_select": (ctypes.c_int, [ctypes.c_int]),
    "ganq_profile_num_kernels": (ctypes.c_int, []),
    "ganq_profile_kernel_name": (ctypes.c_char_p, [ctypes.c_int]),
    "ganq_profile_get": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_c_i64)]),
}


class GanqHipError(RuntimeError):
    pass


_lib = None
_lock = threading.Lock()


def build(force: bool = False) -> str:
    """Compile libganq_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC_DIR, "-j8"] + (["-B"] if force else [])
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if proc.returncode != 0:
        raise GanqHipError("building libganq_hip.so failed:\n" + proc.stdout[-4000:])
    return LIB_PATH


def lib():
    """Load the shared library and bind every symbol the header declares."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise GanqHipError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
                                   f"g.build()'` (or `make -C ganq_amd/csrc`). There is no CPU fallback.")
            try:
                handle = ctypes.CDLL(LIB_PATH)
            except OSError as e:
                raise GanqHipError(f"cannot load {LIB_PATH}: {e}") from e
            for name, (res, args) in SIGNATURES.items():
                try:
                    fn = getattr(handle, name)
                except AttributeError as e:
                    raise GanqHipError(f"{LIB_PATH} does not export {name}") from e
                fn.restype = res
                fn.argtypes = args
            _lib = handle
    return _lib


def _check(rc: int, what: str):
    if rc != 0:
        msg = lib().ganq_hip_last_error()
        raise GanqHipError(f"{what} failed (rc={rc}): {msg.decode() if msg else ''}")


def _stream(device=None) -> int:
    """hipStream_t of torch's current stream ON `device` (default: the current device)"""
    return torch.cuda.current_stream(device).cuda_stream


class _on:
    """Make the device of `t` the current HIP device for the duration of a library call: the library launches on the
    current device (kernel attributes, probe results and helper streams are kept per device), and the stream handed
    over must be that device's.  All tensors of a call must live on ONE device."""

    def __init__(self, *tensors):
        devs = {t.device for t in tensors if t is not None and hasattr(t, "device")}
        if len(devs) != 1:
            raise GanqHipError(f"all tensors of a call must live on one GPU, got {sorted(map(str, devs))}")
        self.device = devs.pop()
        if self.device.type != "cuda":
            raise GanqHipError(f"tensors must live on the GPU (got {self.device}); the HIP path has no CPU fallback")
        self._guard = torch.cuda.device(self.device)

    def __enter__(self):
        self._guard.__enter__()
        return _stream(self.device)

    def __exit__(self, *exc):
        return self._guard.__exit__(*exc)


_ST = object()  # placeholder for the stream argument of a library call


def _call(name: str, tensors, *args):
    """one library call on the device of `tensors` (all on one GPU) with that device current and its current stream"""
    with _on(*tensors) as st:
        rc = getattr(lib(), name)(*[st if a is _ST else a for a in args])
    _check(rc, name)


def debug_option(name: str, value=None):
    """developer / test switches of the library (include/ganq_hip.h); value None resets to the default"""
    if value is None:
        _check(lib().ganq_debug_reset_option(name.encode()), "ganq_debug_reset_option")
    else:
        _check(lib().ganq_debug_set_option(name.encode(), int(value)), "ganq_debug_set_option")


def debug_option_get(name: str) -> int:
    v = ctypes.c_longlong(0)
    _check(lib().ganq_debug_get_option(name.encode(), ctypes.byref(v)), "ganq_debug_get_option")
    return v.value


def _ptr(t):
    return None if t is None else t.data_ptr()


def _dev_f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise GanqHipError(f"{name} must live on the GPU (got {t.device}); the HIP path has no CPU fallback")
    if t.dtype != torch.float32:
        raise GanqHipError(f"{name} must be float32, got {t.dtype}")
    return t.contiguous()


def _workspace(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def selftest(device=None):
    """MFMA accumulation-order probe on `device` (default: the current device); cached per device by the library"""
    with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
        _check(lib().ganq_hip_selftest(_stream()), "ganq_hip_selftest")


def debug_div_check(count: int, seed: int = 1):
    """(mismatches, (a, b)) of the reciprocal-based quotient vs IEEE division over `count` pseudo-random pairs"""
    bad = torch.zeros((), dtype=torch.int64, device="cuda")
    first = torch.zeros(2, dtype=torch.float32, device="cuda")
    _call("ganq_debug_div_check", (bad, first), int(count), int(seed), bad.data_ptr(), first.data_ptr(), _ST)
    return int(bad), tuple(first.tolist())


def debug_wh_product(W, H):
    """(WH, H_fixed) in fp64: the product the fused driver feeds to the T-update and the fixed-point H behind it"""
    W, H = _dev_f32(W, "W"), _dev_f32(H, "H")
    m, n = W.shape
    WH = torch.empty((m, n), dtype=torch.float64, device=W.device)
    Hf = torch.empty((n, n), dtype=torch.float64, device=W.device)
    _call("ganq_debug_wh_product", (W, H, WH, Hf), W.data_ptr(), H.data_ptr(), m, n, WH.data_ptr(), Hf.data_ptr(), _ST)
    return WH, Hf


def solve_s(W, L, T, want_err=False):
    """ganq.py:533-565.  W [m,n], L [n,n] lower, T [m,V] (fp32, cuda) -> Q uint8 [m,n] (, Err fp32 [m,n])."""
    W, T = _dev_f32(W, "W"), _dev_f32(T, "T")
    if not (L.is_cuda and L.dtype == torch.float32 and L.stride(1) == 1):
        L = _dev_f32(L, "L")
    m, n = W.shape
    V = T.shape[1]
    if L.shape != (n, n) or T.shape[0] != m:
        raise GanqHipError(f"shape mismatch W{tuple(W.shape)} L{tuple(L.shape)} T{tuple(T.shape)}")
    Q = torch.empty((m, n), dtype=torch.uint8, device=W.device)
    Err = torch.empty((m, n), dtype=torch.float32, device=W.device) if want_err else None
    ws = _workspace(lib().ganq_solve_s_workspace_bytes(m, n, V), W.device)
    _call("ganq_solve_s", (W, L, T, Q, Err, ws), W.data_ptr(), L.data_ptr(), L.stride(0), T.data_ptr(), m, n, V, Q.data_ptr(), _ptr(Err),
                              ws.data_ptr(), ws.numel(), _ST)
    return (Q, Err) if want_err else Q


def matmul_f32(A, B):
    A, B = _dev_f32(A, "A"), _dev_f32(B, "B")
    m, k = A.shape
    if B.shape[0] != k:
        raise GanqHipError(f"shape mismatch A{tuple(A.shape)} B{tuple(B.shape)}")
    n = B.shape[1]
    C = torch.empty((m, n), dtype=torch.float32, device=A.device)
    _call("ganq_matmul_f32", (A, B, C), A.data_ptr(), B.data_ptr(), m, k, n, C.data_ptr(), _ST)
    return C


def update_t(WH, H, Q, V, rcond=-1.0, want_ab=False):
    """ganq.py:570-591 (gelsd branch).  -> T [m,V] (, A [m,V,V], b [m,V])."""
    WH, H = _dev_f32(WH, "WH"), _dev_f32(H, "H")
    if Q.dtype != torch.uint8 or not Q.is_cuda:
        raise GanqHipError("Q must be a uint8 cuda tensor")
    Q = Q.contiguous()
    m, n = WH.shape
    T = torch.empty((m, V), dtype=torch.float32, device=WH.device)
    A = torch.empty((m, V, V), dtype=torch.float32, device=WH.device) if want_ab else None
    b = torch.empty((m, V), dtype=torch.float32, device=WH.device) if want_ab else None
    ws = _workspace(lib().ganq_update_t_workspace_bytes(m, n, V), WH.device)
    _call("ganq_update_t", (WH, H, Q, T, A, b, ws), WH.data_ptr(), H.data_ptr(), Q.data_ptr(), m, n, V, float(rcond), T.data_ptr(), _ptr(A),
                               _ptr(b), ws.data_ptr(), ws.numel(), _ST)
    return (T, A, b) if want_ab else T


def quad_loss(W, H, T, Q):
    """ganq.py:392-395 on Wq = T.gather(1,Q).  Returns a 0-dim float64 cuda tensor (no sync)."""
    W, H, T = _dev_f32(W, "W"), _dev_f32(H, "H"), _dev_f32(T, "T")
    Q = Q.contiguous()
    m, n = W.shape
    out = torch.empty((), dtype=torch.float64, device=W.device)
    ws = _workspace(lib().ganq_quad_loss_workspace_bytes(m, n, T.shape[1]), W.device)
    _call("ganq_quad_loss", (W, H, T, Q, out, ws), W.data_ptr(), H.data_ptr(), T.data_ptr(), Q.data_ptr(), m, n, T.shape[1],
                                out.data_ptr(), ws.data_ptr(), ws.numel(), _ST)
    return out


def dequant_losses(W, T, Q, hinv_diag, want_losses=True):
    """ganq.py:633-638.  -> (Wq [m,n], Losses [m,n] or None)."""
    W, T = _dev_f32(W, "W"), _dev_f32(T, "T")
    Q = Q.contiguous()
    m, n = W.shape
    Wq = torch.empty((m, n), dtype=torch.float32, device=W.device)
    Lo = torch.empty((m, n), dtype=torch.float32, device=W.device) if want_losses else None
    hd = _dev_f32(hinv_diag, "hinv_diag") if want_losses else None
    _call("ganq_dequant_losses", (W, T, Q, hd, Wq, Lo), W.data_ptr(), T.data_ptr(), Q.data_ptr(), _ptr(hd), m, n, T.shape[1],
                                     Wq.data_ptr(), _ptr(Lo), _ST)
    return Wq, Lo


def run_layer_workspace(m, n, V, device):
    return _workspace(lib().ganq_run_layer_workspace_bytes(m, n, V), device)


def run_layer(W, H, L, T0, K, alias_q=True, rcond=-1.0, workspace=None, helpers=True):
    """ganq.py:516-634.  -> (T_best [m,V], Q uint8 [m,n], dists float64 [K], best_k int32 0-dim); all on the GPU."""
    W, H, T0 = _dev_f32(W, "W"), _dev_f32(H, "H"), _dev_f32(T0, "T0")
    if not (L.is_cuda and L.dtype == torch.float32 and L.stride(1) == 1):
        L = _dev_f32(L, "L")
    m, n = W.shape
    V = T0.shape[1]
    if H.shape != (n, n) or L.shape != (n, n) or T0.shape[0] != m:
        raise GanqHipError(f"shape mismatch W{tuple(W.shape)} H{tuple(H.shape)} L{tuple(L.shape)} T0{tuple(T0.shape)}")
    T = torch.empty((m, V), dtype=torch.float32, device=W.device)
    Q = torch.empty((m, n), dtype=torch.uint8, device=W.device)
    dists = torch.zeros((max(K, 1),), dtype=torch.float64, device=W.device)
    best_k = torch.full((), -1, dtype=torch.int32, device=W.device)
    ws = workspace if workspace is not None else run_layer_workspace(m, n, V, W.device)
    _call("ganq_run_layer", (W, H, L, T0, T, Q, dists, best_k, ws), W.data_ptr(), H.data_ptr(), L.data_ptr(), L.stride(0), T0.data_ptr(), m, n, V, int(K),
          (FLAG_ALIAS_Q if alias_q else 0) | (0 if helpers else FLAG_NO_HELPERS), float(rcond), T.data_ptr(), Q.data_ptr(),
          dists.data_ptr(), best_k.data_ptr(), ws.data_ptr(), ws.numel(), _ST)
    return T, Q, dists[:K], best_k


def run_layer_rows(W, H, L, T0, K, alias_q=True, rcond=-1.0, workspace=None, want_q_all=None, helpers=True):
    """ganq_run_layer_rows: the fused loop on a slice of a layer's rows with the per-iteration records the owner of the
    whole layer needs.  -> dict(T_all [K,m,V], loss_rows_all [K,m] fp64, Q_last [m,n] uint8 (indices of the last
    iteration), Q_all [K,m,n] or None, dists [K] / best_k / T_best of the slice alone)."""
    W, H, T0 = _dev_f32(W, "W"), _dev_f32(H, "H"), _dev_f32(T0, "T0")
    if not (L.is_cuda and L.dtype == torch.float32 and L.stride(1) == 1):
        L = _dev_f32(L, "L")
    m, n = W.shape
    V = T0.shape[1]
    if H.shape != (n, n) or L.shape != (n, n) or T0.shape[0] != m:
        raise GanqHipError(f"shape mismatch W{tuple(W.shape)} H{tuple(H.shape)} L{tuple(L.shape)} T0{tuple(T0.shape)}")
    if want_q_all is None:
        want_q_all = not alias_q
    K = int(K)
    dev = W.device
    T = torch.empty((m, V), dtype=torch.float32, device=dev)
    Q = torch.empty((m, n), dtype=torch.uint8, device=dev)
    dists = torch.zeros((max(K, 1),), dtype=torch.float64, device=dev)
    best_k = torch.full((), -1, dtype=torch.int32, device=dev)
    T_all = torch.empty((K, m, V), dtype=torch.float32, device=dev)
    loss_all = torch.empty((K, m), dtype=torch.float64, device=dev)
    Q_all = torch.empty((K, m, n), dtype=torch.uint8, device=dev) if want_q_all else None
    ws = workspace if workspace is not None else run_layer_workspace(m, n, V, dev)
    # always run the aliasing variant inside the slice: Q then holds the LAST iteration's indices, which is what the
    # reference's torch branch returns; the per-iteration indices (if wanted) come back in Q_all
    _call("ganq_run_layer_rows", (W, H, L, T0, T, Q, dists, best_k, T_all, loss_all, Q_all, ws), W.data_ptr(), H.data_ptr(),
          L.data_ptr(), L.stride(0), T0.data_ptr(), m, n, V, K, FLAG_ALIAS_Q | (0 if helpers else FLAG_NO_HELPERS), float(rcond),
          T.data_ptr(), Q.data_ptr(),
          dists.data_ptr(), best_k.data_ptr(), T_all.data_ptr(), loss_all.data_ptr(), _ptr(Q_all), ws.data_ptr(), ws.numel(),
          _ST)
    return dict(T_all=T_all, loss_rows_all=loss_all, Q_last=Q, Q_all=Q_all, dists=dists[:K], best_k=best_k, T_best=T)


def select_best(loss_rows_all):
    """ganq_select_best: loss_rows_all [K,m] fp64 (all rows of the layer, row order) -> (dists [K] fp64, best_k int32 0-dim)"""
    if loss_rows_all.dtype != torch.float64 or not loss_rows_all.is_cuda or loss_rows_all.dim() != 2:
        raise GanqHipError("select_best: loss_rows_all must be a [K, m] float64 cuda tensor")
    la = loss_rows_all.contiguous()
    K, m = la.shape
    dists = torch.zeros((max(K, 1),), dtype=torch.float64, device=la.device)
    best_k = torch.full((), -1, dtype=torch.int32, device=la.device)
    _call("ganq_select_best", (la, dists, best_k), la.data_ptr(), m, K, dists.data_ptr(), best_k.data_ptr(), _ST)
    return dists[:K], best_k


def profile_enable(on: bool = True, only: str = None):
    """HIP-event timing of the library's kernels; `only`: instrument just this kernel (the events of a fully
    instrumented run leave idle gaps between the dependent launches)."""
    lib().ganq_profile_reset()
    kid = -1
    if only is not None:
        names = [lib().ganq_profile_kernel_name(i).decode() for i in range(lib().ganq_profile_num_kernels())]
        kid = names.index(only)
    lib().ganq_profile_select(kid)
    lib().ganq_profile_enable(1 if on else 0)


def profile_report():
    """{kernel name: (total_ms, launches)} for everything recorded since profile_enable(); synchronises first."""
    torch.cuda.synchronize()
    out = {}
    for kid in range(lib().ganq_profile_num_kernels()):
        ms, cnt = ctypes.c_double(0.0), _c_i64(0)
        _check(lib().ganq_profile_get(kid, ctypes.byref(ms), ctypes.byref(cnt)), "ganq_profile_get")
        if cnt.value:
            out[lib().ganq_profile_kernel_name(kid).decode()] = (ms.value, cnt.value)
    return out


_DTYPE_CODE = {torch.float16: 0, torch.bfloat16: 1}


def _act_dtype(t, name):
    if t.dtype not in _DTYPE_CODE:
        raise GanqHipError(f"{name} must be float16 or bfloat16, got {t.dtype}")
    if not t.is_cuda:
        raise GanqHipError(f"{name} must live on the GPU; the HIP path has no CPU fallback")
    return _DTYPE_CODE[t.dtype]


def debug_gemm_h16(x, w, bias=None, addend=None):
    """developer / tests: y = x @ w.T (+ bias) (+ addend) by csrc/gemm_h16.hip; x [M,K], w [N,K] fp16 / bf16"""
    code = _act_dtype(x, "x")
    if w.dtype != x.dtype or not w.is_cuda:
        raise GanqHipError("w must have x's dtype and live on the GPU")
    x, w = x.contiguous(), w.contiguous()
    M, K = x.shape
    N = w.shape[0]
    y = torch.empty((M, N), dtype=x.dtype, device=x.device)
    _call("ganq_debug_gemm_h16", (x, w, y), x.data_ptr(), w.data_ptr(), _ptr(bias), _ptr(addend), code, M, N, K, y.data_ptr(), _ST)
    return y


def hessian_accum(H, X, nsamples_before: int, batch: int):
    """gptq.py:96-131.  H [n,n] fp32 (updated in place), X [rows, n] fp16/bf16 (one calibration batch of
    `batch` sequences, flattened)."""
    if not (H.is_cuda and H.dtype == torch.float32 and H.is_contiguous()):
        raise GanqHipError("H must be a contiguous float32 cuda tensor")
    code = _act_dtype(X, "X")
    X = X.contiguous()
    rows, n = X.shape
    if H.shape != (n, n):
        raise GanqHipError(f"shape mismatch H{tuple(H.shape)} X{tuple(X.shape)}")
    # partial tiles of the token-split launches: torch-owned scratch on the current stream (the caching allocator
    # re-uses it from call to call; nothing outlives the call on the library's side)
    with torch.cuda.device(H.device):
        nbytes = lib().ganq_hessian_workspace_bytes(rows, n)
    ws = _workspace(nbytes, H.device) if nbytes else None
    _call("ganq_hessian_accum", (H, X, ws), H.data_ptr(), X.data_ptr(), code, rows, n, int(nsamples_before), int(batch),
          ws.data_ptr() if ws is not None else None, int(nbytes), _ST)
    return H


def hessian_t_supported(n: int, stage_tokens: int, device=None) -> bool:
    """whether a layer of n in_features is served by the transposed staging path (staging buffer Xt [n, stage_tokens])"""
    with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
        return bool(lib().ganq_hessian_t_supported(int(n), int(stage_tokens)))


def hessian_stage_t(Xt, X, tok0: int):
    """the transposing staging copy of one calibration batch: Xt[:, tok0 : tok0 + rows] = X^T  (X [rows, n], Xt [n, ldt], fp16 / bf16)"""
    code = _act_dtype(X, "X")
    if not (Xt.is_cuda and Xt.dtype == X.dtype and Xt.dim() == 2 and Xt.is_contiguous()) or _act_dtype(Xt, "Xt") != code:
        raise GanqHipError("Xt must be a contiguous cuda tensor [in_features, tokens] of X's dtype")
    X = X.contiguous()
    rows, n = X.shape
    if Xt.shape[0] != n or tok0 < 0 or tok0 + rows > Xt.shape[1]:
        raise GanqHipError(f"hessian_stage_t: batch {tuple(X.shape)} at token {tok0} does not fit Xt{tuple(Xt.shape)}")
    _call("ganq_hessian_stage_t", (Xt, X), Xt.data_ptr(), Xt.shape[1], X.data_ptr(), rows, n, int(tok0), _ST)
    return Xt


def hessian_accum_t(H, Xt, rows: int, nsamples_before: int, batch: int):
    """gptq.py:96-131 for a staged group in transposed layout: H [n,n] fp32 (in place), Xt [n, ldt], its first `rows` tokens
    (a multiple of 32) = `batch` sequences"""
    if not (H.is_cuda and H.dtype == torch.float32 and H.is_contiguous()):
        raise GanqHipError("H must be a contiguous float32 cuda tensor")
    code = _act_dtype(Xt, "Xt")
    if not (Xt.dim() == 2 and Xt.is_contiguous() and H.shape == (Xt.shape[0], Xt.shape[0])):
        raise GanqHipError(f"shape mismatch H{tuple(H.shape)} Xt{tuple(Xt.shape)}")
    n, ldt = Xt.shape
    with torch.cuda.device(H.device):
        nbytes = lib().ganq_hessian_t_workspace_bytes(n)
    ws = _workspace(nbytes, H.device)
    _call("ganq_hessian_accum_t", (H, Xt, ws), H.data_ptr(), Xt.data_ptr(), ldt, code, int(rows), n, int(nsamples_before), int(batch),
          ws.data_ptr(), int(nbytes), _ST)
    return H


def cholesky(H, check: bool = True):
    """torch.linalg.cholesky(H) for a symmetric fp32 matrix on the GPU (lower factor, new tensor).  With check=True a
    non-positive pivot raises torch.linalg.LinAlgError like torch does (one host sync); otherwise -> (L, info tensor)."""
    H = _dev_f32(H, "H")
    n = H.shape[0]
    if H.shape != (n, n):
        raise GanqHipError(f"cholesky: square matrix expected, got {tuple(H.shape)}")
    L = H.clone()
    info = torch.zeros((), dtype=torch.int32, device=H.device)
    ws = _workspace(lib().ganq_cholesky_workspace_bytes(n), H.device)
    _call("ganq_cholesky", (L, info, ws), L.data_ptr(), n, L.stride(0) if n else 0, info.data_ptr(), ws.data_ptr(), ws.numel(), _ST)
    if not check:
        return L, info
    bad = int(info)
    if bad:
        raise torch.linalg.LinAlgError(f"ganq_cholesky: the input is not positive-definite (leading minor of order {bad})")
    return L


def cholesky_inplace(A, check: bool = True):
    """ganq_cholesky on A itself (fp32 [n,n], contiguous rows): the lower factor replaces A, the strict upper part is zeroed.
    -> A, or (A, info tensor) with check=False"""
    if not (A.is_cuda and A.dtype == torch.float32 and A.dim() == 2 and A.shape[0] == A.shape[1] and A.stride(1) == 1):
        raise GanqHipError("cholesky_inplace: a square fp32 cuda matrix with contiguous rows expected")
    n = A.shape[0]
    info = torch.zeros((), dtype=torch.int32, device=A.device)
    ws = _workspace(lib().ganq_cholesky_workspace_bytes(n), A.device)
    _call("ganq_cholesky", (A, info, ws), A.data_ptr(), n, A.stride(0) if n else 0, info.data_ptr(), ws.data_ptr(), ws.numel(), _ST)
    if not check:
        return A, info
    bad = int(info)
    if bad:
        raise torch.linalg.LinAlgError(f"ganq_cholesky: the input is not positive-definite (leading minor of order {bad})")
    return A


def prologue_rowstats(H):
    """gptq.py:267-269,289-291 in one read of H -> (diag [n], rowabs [n] = sum_j |H[i][j]|), fp32"""
    H = _dev_f32(H, "H")
    n = H.shape[0]
    diag = torch.empty((n,), dtype=torch.float32, device=H.device)
    rowabs = torch.empty((n,), dtype=torch.float32, device=H.device)
    _call("ganq_prologue_rowstats", (H, diag, rowabs), H.data_ptr(), n, diag.data_ptr(), rowabs.data_ptr(), _ST)
    return diag, rowabs


def prologue_gather(H, perm, diag_fixed, outputs):
    """One read of H, every output written directly: outputs = [(add [n] fp32 or None, flip: bool), ...] (up to three) ->
    list of new [n,n] fp32 tensors, out[i][j] = H'[perm i][perm j] + (i == j ? add[i] : 0), index-reversed when flip;
    H' = H with diag_fixed on its diagonal; perm int64 [n] or None."""
    H = _dev_f32(H, "H")
    n = H.shape[0]
    if not 1 <= len(outputs) <= 3:
        raise GanqHipError("prologue_gather: one to three outputs")
    diag_fixed = _dev_f32(diag_fixed, "diag_fixed")
    perm_c = None if perm is None else perm.to(device=H.device, dtype=torch.int64).contiguous()
    outs, args, keep = [], [], [H, perm_c, diag_fixed]
    for k in range(3):
        if k < len(outputs):
            add, flip = outputs[k]
            add_c = None if add is None else _dev_f32(add, "add")
            o = torch.empty((n, n), dtype=torch.float32, device=H.device)
            outs.append(o)
            keep += [o, add_c]
            args += [o.data_ptr(), _ptr(add_c), 1 if flip else 0]
        else:
            args += [None, None, 0]
    _call("ganq_prologue_gather", tuple(keep), H.data_ptr(), _ptr(perm_c), diag_fixed.data_ptr(), n, *args, _ST)
    return outs


def prologue_weights(W, perm, dead, mean_fill: bool):
    """gptq.py:270-276,283: W_out[r][c] = dead[perm c] ? fill_r : W[r][perm c] (fill: 0, or the row mean over the live columns)"""
    W = _dev_f32(W, "W")
    m, n = W.shape
    dead_c = dead.to(device=W.device, dtype=torch.uint8).contiguous()
    perm_c = None if perm is None else perm.to(device=W.device, dtype=torch.int64).contiguous()
    out = torch.empty((m, n), dtype=torch.float32, device=W.device)
    _call("ganq_prologue_weights", (W, perm_c, dead_c, out), W.data_ptr(), _ptr(perm_c), dead_c.data_ptr(), m, n, 1 if mean_fill else 0,
          out.data_ptr(), _ST)
    return out


def kmeans_init(W, col_weight, V: int):
    """ganq.py:423-438.  W [m,n] fp32, col_weight [n] float64 (diag(Hinv)^-4) or None -> T0 [m,V] fp32."""
    W = _dev_f32(W, "W")
    m, n = W.shape
    cw = None
    if col_weight is not None:
        cw = col_weight.to(device=W.device, dtype=torch.float64).contiguous()
        if cw.shape != (n,):
            raise GanqHipError("col_weight must have shape [n]")
    T0 = torch.empty((m, V), dtype=torch.float32, device=W.device)
    ws = _workspace(lib().ganq_kmeans_workspace_bytes(m, n, V), W.device)
    _call("ganq_kmeans_init", (W, cw, T0, ws), W.data_ptr(), _ptr(cw), m, n, V, T0.data_ptr(), ws.data_ptr(), ws.numel(), _ST)
    return T0


def pack_indices(Q, bits: int):
    """Q uint8 [m,n] (original column order) -> qweight int32 [n*bits/32, m] (GPTQ layout)."""
    if Q.dtype != torch.uint8 or not Q.is_cuda:
        raise GanqHipError("Q must be a uint8 cuda tensor")
    Q = Q.contiguous()
    m, n = Q.shape
    qw = torch.empty((n * bits // 32, m), dtype=torch.int32, device=Q.device)
    _call("ganq_pack_indices", (Q, qw), Q.data_ptr(), m, n, bits, qw.data_ptr(), _ST)
    return qw


def unpack_indices(qweight, n: int, bits: int):
    if qweight.dtype != torch.int32 or not qweight.is_cuda:
        raise GanqHipError("qweight must be an int32 cuda tensor")
    qweight = qweight.contiguous()
    m = qweight.shape[1]
    Q = torch.empty((m, n), dtype=torch.uint8, device=qweight.device)
    _call("ganq_unpack_indices", (qweight, Q), qweight.data_ptr(), m, n, bits, Q.data_ptr(), _ST)
    return Q


def lut_dequant(qweight, lut, n: int, bits: int):
    code = _act_dtype(lut, "lut")
    qweight, lut = qweight.contiguous(), lut.contiguous()
    m = lut.shape[0]
    Wq = torch.empty((m, n), dtype=lut.dtype, device=lut.device)
    _call("ganq_lut_dequant", (qweight, lut, Wq), qweight.data_ptr(), lut.data_ptr(), code, m, n, bits, Wq.data_ptr(), _ST)
    return Wq


_LUT_WS = {}  # (device index, stream) -> zero-initialised workspace, kept across calls (the kernel leaves it clean)


def _lut_workspace(nbytes: int, device):
    key = (device.index if device.index is not None else torch.cuda.current_device(), _stream(device))
    ws = _LUT_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _call("ganq_lut_linear_workspace_init", (ws,), ws.data_ptr(), ws.numel(), _ST)
        _LUT_WS[key] = ws
    return ws


def outlier_split(W, ratio: float):
    """Algorithm 2 of the paper on the device.  W [m,n] fp32 is overwritten by W_dense (outliers zeroed); returns the
    outliers as CSR (rowptr int32 [m+1], cols int32 [nnz] ascending per row, vals fp32 [nnz]) and the cut-offs [m,2]."""
    W = _dev_f32(W, "W")
    m, n = W.shape
    cut = torch.empty((m, 2), dtype=torch.float32, device=W.device)
    counts = torch.empty((m,), dtype=torch.int32, device=W.device)
    rowptr = torch.empty((m + 1,), dtype=torch.int32, device=W.device)
    _call("ganq_outlier_cutoffs", (W, cut, counts, rowptr), W.data_ptr(), m, n, float(ratio), cut.data_ptr(), counts.data_ptr(), rowptr.data_ptr(),
                                      _ST)
    nnz = int(rowptr[m])  # the one host read: sizes of the CSR arrays
    cols = torch.empty((nnz,), dtype=torch.int32, device=W.device)
    vals = torch.empty((nnz,), dtype=torch.float32, device=W.device)
    if nnz:
        _call("ganq_outlier_extract", (W, cut, rowptr, cols, vals), W.data_ptr(), m, n, cut.data_ptr(), rowptr.data_ptr(), cols.data_ptr(), vals.data_ptr(),
                                          _ST)
    return rowptr, cols, vals, cut


def outlier_matmul(x, rowptr, cols, vals, m: int):
    """x [M,n] fp16/bf16, CSR outliers of a [m,n] weight (vals in x's dtype) -> fp32 [M,m] = x @ W_sparse^T"""
    code = _act_dtype(x, "x")
    if vals.dtype != x.dtype or rowptr.dtype != torch.int32 or cols.dtype != torch.int32:
        raise GanqHipError("outlier_matmul: vals must have x's dtype, rowptr / cols must be int32")
    if rowptr.numel() != m + 1:
        raise GanqHipError("outlier_matmul: rowptr must have out_features + 1 entries")
    x = x.contiguous()
    M, n = x.shape
    out = torch.empty((M, m), dtype=torch.float32, device=x.device)
    _call("ganq_outlier_matmul", (x, rowptr, cols, vals, out), x.data_ptr(), code, M, m, n, rowptr.data_ptr(), _ptr(cols) if cols.numel() else None,
                                     _ptr(vals) if vals.numel() else None, out.data_ptr(), _ST)
    return out


def lut_linear_outliers(x, qweight, lut, bias, bits: int, rowptr, cols, vals):
    """LUT forward of a layer with sparse outliers (CSR by output feature, vals in x's dtype) in one library call."""
    code = _act_dtype(x, "x")
    if lut.dtype != x.dtype or (bias is not None and bias.dtype != x.dtype) or vals.dtype != x.dtype:
        raise GanqHipError("x, lut, bias and outlier values must share one dtype")
    if rowptr.dtype != torch.int32 or cols.dtype != torch.int32 or rowptr.numel() != lut.shape[0] + 1:
        raise GanqHipError("lut_linear_outliers: rowptr / cols must be int32, rowptr of length out_features + 1")
    x, qweight, lut = x.contiguous(), qweight.contiguous(), lut.contiguous()
    M, n = x.shape
    m = lut.shape[0]
    max_rows = ((1 << 31) - 1) // (2 * max(n, 1)) - 256  # see lut_linear
    if M > max_rows:
        step = max(256, max_rows // 256 * 256)
        return torch.cat([lut_linear_outliers(x[r:r + step], qweight, lut, bias, bits, rowptr, cols, vals) for r in range(0, M, step)], dim=0)
    y = torch.empty((M, m), dtype=x.dtype, device=x.device)
    ws = _lut_workspace(lib().ganq_lut_linear_outliers_workspace_bytes(M, m, n, bits), x.device)
    _call("ganq_lut_linear_fwd_outliers", (x, qweight, lut, bias, rowptr, cols, vals, y, ws), x.data_ptr(), qweight.data_ptr(), lut.data_ptr(), _ptr(bias), rowptr.data_ptr(),
                                              cols.data_ptr() if cols.numel() else None,
                                              vals.data_ptr() if vals.numel() else None, code, M, m, n, bits, y.data_ptr(),
                                              ws.data_ptr(), ws.numel(), _ST)
    return y


def lut_linear(x, qweight, lut, bias, bits: int, addend=None):
    """x [M,n] fp16/bf16 (any M: decode kernels up to 64 rows, the fused LUT-dequant GEMM above), qweight int32 [n*bits/32, m],
    lut [m,V], bias [m] or None -> y [M,m].
    addend: optional fp32 [M,m] added before the rounding to x's dtype (the sparse-outlier product)."""
    code = _act_dtype(x, "x")
    if lut.dtype != x.dtype or (bias is not None and bias.dtype != x.dtype):
        raise GanqHipError("x, lut and bias must share one dtype")
    x, qweight, lut = x.contiguous(), qweight.contiguous(), lut.contiguous()
    M, n = x.shape
    m = lut.shape[0]
    # the GEMM kernels address x through 32-bit buffer offsets: more than ~2 GB of activations go in row chunks
    max_rows = ((1 << 31) - 1) // (2 * max(n, 1)) - 256
    if M > max_rows:
        step = max(256, max_rows // 256 * 256)
        return torch.cat([lut_linear(x[r:r + step], qweight, lut, bias, bits, None if addend is None else addend[r:r + step].contiguous())
                          for r in range(0, M, step)], dim=0)
    y = torch.empty((M, m), dtype=x.dtype, device=x.device)
    ws = _lut_workspace(lib().ganq_lut_linear_workspace_bytes(M, m, n, bits), x.device)
    if addend is not None:
        if addend.dtype != torch.float32 or tuple(addend.shape) != (M, m) or not addend.is_contiguous():
            raise GanqHipError("lut_linear: addend must be a contiguous fp32 [M, out_features] tensor")
        _call("ganq_lut_linear_fwd_add", (x, qweight, lut, bias, addend, y, ws), x.data_ptr(), qweight.data_ptr(), lut.data_ptr(), _ptr(bias), addend.data_ptr(), code,
                                             M, m, n, bits, y.data_ptr(), ws.data_ptr(), ws.numel(), _ST)
        return y
    _call("ganq_lut_linear_fwd", (x, qweight, lut, bias, y, ws), x.data_ptr(), qweight.data_ptr(), lut.data_ptr(), _ptr(bias), code, M, m, n, bits,
                                     y.data_ptr(), ws.data_ptr(), ws.numel(), _ST)
    return y
