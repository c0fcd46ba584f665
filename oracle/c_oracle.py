"""ctypes view of oracle/liboracle.so (extend_matrices.c)  --  TEST INFRASTRUCTURE ONLY.

The compiled CPU twin of :func:`oracle.qp_oracle.extend_matrices` (reference
tools.py:14-33, C++ twin cpp/src/tools.cc:83-144).  Built by ``make -C oracle``
(``__graft_entry__.build()`` does it); only tests, smoke() and bench tools load it.
"""
import ctypes
import os

import numpy as np

_lib = None


def load():
    global _lib
    if _lib is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/liboracle.so is not built: make -C oracle")
        lib = ctypes.CDLL(path)
        ptr = ctypes.c_void_p
        lib.oracle_extend_matrices_batch.restype = ctypes.c_int
        lib.oracle_extend_matrices_batch.argtypes = [ptr, ptr, ptr, ptr] + [ctypes.c_int] * 5
        _lib = lib
    return _lib


def extend_matrices_batch(A, B, N, ltv=False):
    """``A``: ``(count, n, n)`` (LTV: ``(count, N, n, n)``), ``B`` likewise with ``m`` columns.
    Returns ``S (count, N, n, n)`` and ``U (count, m, N, N, n)``."""
    A = np.ascontiguousarray(A, dtype=np.float64)
    B = np.ascontiguousarray(B, dtype=np.float64)
    count, n, m = A.shape[0], A.shape[-1], B.shape[-1]
    S = np.empty((count, N, n, n))
    U = np.empty((count, m, N, N, n))
    rc = load().oracle_extend_matrices_batch(A.ctypes.data, B.ctypes.data, S.ctypes.data,
                                             U.ctypes.data, count, N, n, m, 1 if ltv else 0)
    if rc:
        raise MemoryError("oracle_extend_matrices_batch")
    return S, U


def extend_matrices(N, A, B):
    """Same signature and return value as the reference's tools.extend_matrices."""
    A = np.asarray(A, dtype=np.float64)
    B = np.asarray(B, dtype=np.float64).reshape(A.shape[0], -1)
    S, U = extend_matrices_batch(A[None], B[None], N)
    return S[0], [U[0, j] for j in range(B.shape[1])]
