"""numpy binding of oracle/ganq_oracle.c (see that file's header; test infrastructure only)."""
import ctypes
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, "libganq_oracle.so")
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)
_u8p = ctypes.POINTER(ctypes.c_uint8)
_u16p = ctypes.POINTER(ctypes.c_uint16)
_i32p = ctypes.POINTER(ctypes.c_int32)
_i64 = ctypes.c_int64


def build(force: bool = False) -> str:
    src = os.path.join(_DIR, "ganq_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _DIR, "-B"], check=True, stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.ganq_oracle_kmeans_cost.restype = ctypes.c_double
    return _lib


def _p(a, t):
    return None if a is None else a.ctypes.data_as(t)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def num_threads() -> int:
    return lib().ganq_oracle_num_threads()


def set_num_threads(t: int) -> None:
    lib().ganq_oracle_set_num_threads(int(t))


def solve_s(W, L, T, want_err=False):
    W, L, T = _f32(W), _f32(L), _f32(T)
    m, n = W.shape
    V = T.shape[1]
    assert L.shape == (n, n) and T.shape[0] == m
    Q = np.empty((m, n), dtype=np.uint8)
    Err = np.empty((m, n), dtype=np.float32) if want_err else None
    rc = lib().ganq_oracle_solve_s(_p(W, _f32p), _p(L, _f32p), _i64(n), _p(T, _f32p), _i64(m), _i64(n), V,
                                   _p(Q, _u8p), _p(Err, _f32p))
    if rc:
        raise RuntimeError(f"ganq_oracle_solve_s failed rc={rc}")
    return (Q, Err) if want_err else Q


def matmul(A, B):
    A, B = _f32(A), _f32(B)
    m, k = A.shape
    n = B.shape[1]
    C = np.empty((m, n), dtype=np.float32)
    lib().ganq_oracle_matmul(_p(A, _f32p), _p(B, _f32p), _i64(m), _i64(k), _i64(n), _p(C, _f32p))
    return C


def update_t(WH, H, Q, V, rcond=-1.0, want_ab=False):
    WH, H = _f32(WH), _f32(H)
    Q = np.ascontiguousarray(Q, dtype=np.uint8)
    m, n = WH.shape
    T = np.empty((m, V), dtype=np.float32)
    A = np.empty((m, V, V), dtype=np.float32) if want_ab else None
    b = np.empty((m, V), dtype=np.float32) if want_ab else None
    rc = lib().ganq_oracle_update_t(_p(WH, _f32p), _p(H, _f32p), _p(Q, _u8p), _i64(m), _i64(n), V,
                                    ctypes.c_double(rcond), _p(T, _f32p), _p(A, _f32p), _p(b, _f32p))
    if rc:
        raise RuntimeError(f"ganq_oracle_update_t failed rc={rc}")
    return (T, A, b) if want_ab else T


def minnorm_solve(A, b, rcond=-1.0):
    A, b = _f32(A), _f32(b)
    m, V = b.shape
    T = np.empty((m, V), dtype=np.float32)
    lib().ganq_oracle_minnorm_solve(_p(A, _f32p), _p(b, _f32p), _i64(m), V, ctypes.c_double(rcond), _p(T, _f32p))
    return T


def quad_loss(W, H, T, Q, want_rows=False):
    W, H, T = _f32(W), _f32(H), _f32(T)
    Q = np.ascontiguousarray(Q, dtype=np.uint8)
    m, n = W.shape
    out = ctypes.c_double(0.0)
    rows = np.empty(m, dtype=np.float64) if want_rows else None
    lib().ganq_oracle_quad_loss(_p(W, _f32p), _p(H, _f32p), _p(T, _f32p), _p(Q, _u8p), _i64(m), _i64(n),
                                T.shape[1], ctypes.byref(out), _p(rows, _f64p))
    return (out.value, rows) if want_rows else out.value


def dequant_losses(W, T, Q, hinv_diag):
    W, T, hinv_diag = _f32(W), _f32(T), _f32(hinv_diag)
    Q = np.ascontiguousarray(Q, dtype=np.uint8)
    m, n = W.shape
    Wq = np.empty((m, n), dtype=np.float32)
    Lo = np.empty((m, n), dtype=np.float32)
    lib().ganq_oracle_dequant_losses(_p(W, _f32p), _p(T, _f32p), _p(Q, _u8p), _p(hinv_diag, _f32p), _i64(m),
                                     _i64(n), T.shape[1], _p(Wq, _f32p), _p(Lo, _f32p))
    return Wq, Lo


def kmeans_init(W, weights, V):
    W = _f32(W)
    m, n = W.shape
    wts = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
    T0 = np.empty((m, V), dtype=np.float32)
    rc = lib().ganq_oracle_kmeans_init(_p(W, _f32p), _p(wts, _f64p), _i64(m), _i64(n), V, _p(T0, _f32p))
    if rc:
        raise RuntimeError(f"ganq_oracle_kmeans_init failed rc={rc}")
    return T0


def kmeans_cost(w, weights, labels, V):
    w = _f32(w)
    wts = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    return lib().ganq_oracle_kmeans_cost(_p(w, _f32p), _p(wts, _f64p), _i64(w.shape[0]), _p(labels, _i32p), V)


def run_layer(W, H, L, T0, K, alias_q=True, rcond=-1.0):
    W, H, L, T0 = _f32(W), _f32(H), _f32(L), _f32(T0)
    m, n = W.shape
    V = T0.shape[1]
    T = np.empty((m, V), dtype=np.float32)
    Q = np.empty((m, n), dtype=np.uint8)
    dists = np.empty(K, dtype=np.float64)
    best_k = ctypes.c_int(-1)
    rc = lib().ganq_oracle_run_layer(_p(W, _f32p), _p(H, _f32p), _p(L, _f32p), _p(T0, _f32p), _i64(m), _i64(n), V,
                                     K, int(bool(alias_q)), ctypes.c_double(rcond), _p(T, _f32p), _p(Q, _u8p),
                                     _p(dists, _f64p), ctypes.byref(best_k))
    if rc:
        raise RuntimeError(f"ganq_oracle_run_layer failed rc={rc}")
    return T, Q, dists, best_k.value


def run_layer_trace(W, H, L, T0, K, rcond=-1.0):
    """all K iterations of the loop on these rows, every iteration recorded: -> dict(T_all [K,m,V], Q_all [K,m,n],
    loss_rows_all [K,m] fp64, dists [K] (these rows only), best_k (of these rows only))"""
    W, H, L, T0 = _f32(W), _f32(H), _f32(L), _f32(T0)
    m, n = W.shape
    V = T0.shape[1]
    T = np.empty((m, V), dtype=np.float32)
    Q = np.empty((m, n), dtype=np.uint8)
    dists = np.empty(K, dtype=np.float64)
    best_k = ctypes.c_int(-1)
    T_all = np.empty((K, m, V), dtype=np.float32)
    Q_all = np.empty((K, m, n), dtype=np.uint8)
    loss_all = np.empty((K, m), dtype=np.float64)
    rc = lib().ganq_oracle_run_layer_trace(_p(W, _f32p), _p(H, _f32p), _p(L, _f32p), _p(T0, _f32p), _i64(m), _i64(n), V,
                                           K, 1, ctypes.c_double(rcond), _p(T, _f32p), _p(Q, _u8p), _p(dists, _f64p),
                                           ctypes.byref(best_k), _p(T_all, _f32p), _p(Q_all, _u8p), _p(loss_all, _f64p))
    if rc:
        raise RuntimeError(f"ganq_oracle_run_layer_trace failed rc={rc}")
    return dict(T_all=T_all, Q_all=Q_all, loss_rows_all=loss_all, dists=dists, best_k=best_k.value)


def hessian_accum(H, X_half, nsamples_before, b):
    """H [n,n] fp32 updated in place; X_half [rows,n] np.float16."""
    X = np.ascontiguousarray(X_half, dtype=np.float16)
    rows, n = X.shape
    assert H.dtype == np.float32 and H.flags.c_contiguous and H.shape == (n, n)
    rc = lib().ganq_oracle_hessian_accum(_p(H, _f32p), _p(X.view(np.uint16), _u16p), _i64(rows), _i64(n),
                                         _i64(nsamples_before), _i64(b))
    if rc:
        raise RuntimeError(f"ganq_oracle_hessian_accum failed rc={rc}")
    return H


def lut_linear(x_half, Q, lut_half, bias_half=None):
    x = np.ascontiguousarray(x_half, dtype=np.float16)
    lut = np.ascontiguousarray(lut_half, dtype=np.float16)
    Q = np.ascontiguousarray(Q, dtype=np.uint8)
    bias = None if bias_half is None else np.ascontiguousarray(bias_half, dtype=np.float16)
    M, n = x.shape
    m, V = lut.shape
    y = np.empty((M, m), dtype=np.float32)
    lib().ganq_oracle_lut_linear(_p(x.view(np.uint16), _u16p), _p(Q, _u8p), _p(lut.view(np.uint16), _u16p),
                                 None if bias is None else _p(bias.view(np.uint16), _u16p), _i64(M), _i64(m),
                                 _i64(n), V, _p(y, _f32p))
    return y


def det_cholesky(A):
    """reproducible lower Cholesky factor of a symmetric fp32 matrix (input generator of the large golden cases; see the
    C function's comment) -> fp32 [n,n]"""
    A = _f32(A)
    n = A.shape[0]
    assert A.shape == (n, n)
    out = np.empty((n, n), dtype=np.float32)
    rc = lib().ganq_oracle_det_cholesky(_p(A, _f32p), _i64(n), _p(out, _f32p))
    if rc:
        raise RuntimeError(f"ganq_oracle_det_cholesky: pivot {rc - 1} is not positive" if rc > 0 else "out of memory")
    return out
