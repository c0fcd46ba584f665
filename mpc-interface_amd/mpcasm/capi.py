"""ctypes binding of libmpcasm.so (C ABI declared in include/mpcasm.h).

The library is built in-tree by ``make -C mpc-interface_amd`` (or
``__graft_entry__.build()``) into this directory.  Loading fails loudly when it
is missing: there is no Python or CPU fallback for the kernels.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (MPCASM_LIB: another build of the library, for A/B measurements)
LIB_PATH = os.environ.get("MPCASM_LIB") or os.path.join(_HERE, "libmpcasm.so")

OK = 0
ERR_LIMIT = -5
OPT_PATH = 1
OPT_PHASE_MASK = 2
OPT_RESIDENT_PER_CU = 3
OPT_RESIDENT_GRID = 6
OPT_P_DIRECT = 5     # the persistent kernel writes P 1: straight to HBM, 2: through LDS when it fits, 0: by the launch's size (read at plan creation)
OPT_JIT = 4            # 0: specialise the persistent kernel for batches >= 512, 1: always, 2: never
PHASE_DEFAULT = 0xBF   # every phase on, cycle stamps (bit 6) off
PHASE_STAMPS = 0x40
STATUS = {
    0: "MPCASM_OK",
    -1: "MPCASM_ERR_ARG",
    -2: "MPCASM_ERR_PLAN",
    -3: "MPCASM_ERR_HIP",
    -4: "MPCASM_ERR_NODEVICE",
    -5: "MPCASM_ERR_LIMIT",
}

# every symbol include/mpcasm.h declares: name -> (restype, argtypes)
_c_double_p = ctypes.POINTER(ctypes.c_double)
_void_p = ctypes.c_void_p
SIGNATURES = {
    "mpcasm_abi_version": (ctypes.c_int, []),
    "mpcasm_device_count": (ctypes.c_int, []),
    "mpcasm_last_hip": (ctypes.c_int, []),
    "mpcasm_status_string": (ctypes.c_char_p, [ctypes.c_int]),
    "mpcasm_set_option": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "mpcasm_fill_su": (ctypes.c_int, [_void_p, _void_p, _void_p, _void_p, ctypes.c_int,
                                      ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      _void_p]),
    "mpcasm_plan_create": (ctypes.c_int, [_void_p, ctypes.c_size_t, _void_p, ctypes.c_size_t,
                                          ctypes.POINTER(_void_p)]),
    "mpcasm_plan_destroy": (ctypes.c_int, [_void_p]),
    "mpcasm_resident_lds_bytes": (ctypes.c_int, [_void_p, ctypes.c_size_t, _void_p, ctypes.c_size_t,
                                                 ctypes.POINTER(ctypes.c_int64)]),
    "mpcasm_jit_check": (ctypes.c_int, [_void_p, ctypes.c_size_t, _void_p, ctypes.c_size_t,
                                        ctypes.c_char_p, ctypes.c_size_t]),
    "mpcasm_plan_sizes": (ctypes.c_int, [_void_p, ctypes.POINTER(ctypes.c_int64)]),
    "mpcasm_plan_csc_sizes": (ctypes.c_int, [_void_p, ctypes.POINTER(ctypes.c_int64)]),
    "mpcasm_plan_set_option": (ctypes.c_int, [_void_p, ctypes.c_int, ctypes.c_int]),
    "mpcasm_assemble_indexed": (ctypes.c_int, [_void_p, ctypes.POINTER(_void_p),
                                               ctypes.POINTER(ctypes.c_int64), _void_p, _void_p, _void_p,
                                               _void_p, _void_p, _void_p, _void_p, _void_p, ctypes.c_int,
                                               _void_p]),
    "mpcasm_plan_last_kernel": (ctypes.c_int, [_void_p]),
    "mpcasm_plan_prepare": (ctypes.c_int, [_void_p, ctypes.c_int]),
    "mpcasm_jit_stats": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int64)]),
    "mpcasm_workspace_bytes": (ctypes.c_int, [_void_p, ctypes.c_int,
                                              ctypes.POINTER(ctypes.c_size_t)]),
    "mpcasm_assemble": (ctypes.c_int, [_void_p, ctypes.POINTER(_void_p),
                                       ctypes.POINTER(ctypes.c_int64), _void_p, _void_p, _void_p,
                                       _void_p, _void_p, _void_p, _void_p, ctypes.c_int, _void_p]),
    "mpcasm_preview_matrices": (ctypes.c_int, [_void_p, ctypes.POINTER(_void_p),
                                               ctypes.POINTER(ctypes.c_int64), _void_p,
                                               ctypes.c_int, _void_p]),
    "mpcasm_preview": (ctypes.c_int, [_void_p, _void_p, _void_p, _void_p, ctypes.c_int,
                                      ctypes.c_int, ctypes.c_int, ctypes.c_int, _void_p]),
    "mpcasm_preview_direct": (ctypes.c_int, [_void_p, ctypes.POINTER(_void_p),
                                             ctypes.POINTER(ctypes.c_int64), _void_p, _void_p,
                                             _void_p, _void_p, ctypes.c_int, _void_p]),
    "mpcasm_goal_distance": (ctypes.c_int, [_void_p, ctypes.c_int64, _void_p, ctypes.c_int64,
                                            _void_p, ctypes.c_int, ctypes.c_int, _void_p,
                                            ctypes.c_int, _void_p]),
    "mpcasm_preview_goal_distance": (ctypes.c_int, [_void_p, ctypes.POINTER(_void_p),
                                                    ctypes.POINTER(ctypes.c_int64), _void_p, _void_p,
                                                    _void_p, _void_p, ctypes.c_int, ctypes.c_int,
                                                    _void_p, _void_p, ctypes.c_int, _void_p]),
    "mpcasm_admm": (ctypes.c_int, [ctypes.c_int, ctypes.c_int] + [_void_p] * 8 + [ctypes.c_double] * 3 +
                    [ctypes.c_int, ctypes.c_int, ctypes.c_int, _void_p, ctypes.c_int, _void_p]),
    "mpcasm_gather": (ctypes.c_int, [_void_p, ctypes.c_int64, _void_p, ctypes.c_int, _void_p,
                                     ctypes.c_int, _void_p]),
    "mpcasm_box_transform": (ctypes.c_int, [_void_p, ctypes.c_int64, ctypes.c_int, _void_p,
                                            ctypes.c_int, ctypes.c_int, _void_p, ctypes.c_int64,
                                            _void_p]),
    "mpcasm_box_transform_ss": (ctypes.c_int, [_void_p, ctypes.c_int64, ctypes.c_int, _void_p,
                                               ctypes.c_int, ctypes.c_int, _void_p, ctypes.c_int,
                                               ctypes.c_int, _void_p, ctypes.c_int64, _void_p]),
}
KERNEL_NAMES = {0: "none", 1: "resident_assemble_kernel (persistent, ahead of time)",
                2: "resident_spec_kernel (persistent, compiled for the plan by hiprtc)",
                3: "fused_assemble_kernel", 4: "staged pipeline (compose_rowsets / hessian / constraints)",
                5: "tiled_assemble_kernel",
                6: "toeplitz_scan_kernel (tiled, scan form: P summed along diagonals)",
                7: "ltv_sweep_kernel (per-step dynamics, no horizon matrix)",
                8: "shared_p_kernel / shared_g_kernel (tiled, shared-model form: weighted sums of per-term matrices)"}
BOX_RECENTER, BOX_TRANSLATE, BOX_ROTATE, BOX_SCALE, BOX_MARGIN = range(5)


class MpcasmError(RuntimeError):
    """A C-ABI call returned a negative status."""

    def __init__(self, status, where, hip=0):
        self.status = status
        self.hip = hip
        msg = "%s failed: %s" % (where, STATUS.get(status, str(status)))
        if hip:
            msg += " (hipError_t %d)" % hip
        super().__init__(msg)


_lib = None


def load():
    """The loaded library (cached); raises RuntimeError when it was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libmpcasm.so is missing (%s): build it with `make -C mpc-interface_amd` "
                "or __graft_entry__.build(); the assembly kernels have no fallback" % LIB_PATH
            )
        try:
            # torch ships its own HIP runtime: load it first so that this library binds to the
            # same copy (two runtimes in one process do not share devices or streams)
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(lib, name)       # AttributeError = symbol not exported
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = lib
    return _lib


def check(status, where):
    if status != OK:
        raise MpcasmError(status, where, load().mpcasm_last_hip())
