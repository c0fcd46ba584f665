"""Shared helpers of the tests: golden loading, error measure, tolerances."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# BASELINE.json north_star: "fp64 P,q,G,h within 1e-10 relative"; measured as
# max |x - ref| / max |ref| over a whole block (an all-zero reference block must be
# reproduced exactly: the error is then max |x| itself).
RTOL = 1e-10
# what the kernels actually achieve on these problems (regression guard)
RTOL_TIGHT = 1e-13


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def rel_err(x, ref):
    x, ref = np.asarray(x, dtype=float), np.asarray(ref, dtype=float)
    assert x.shape == ref.shape, (x.shape, ref.shape)
    if ref.size == 0:
        return 0.0
    scale = float(np.max(np.abs(ref)))
    err = float(np.max(np.abs(x - ref)))
    return err / scale if scale > 0.0 else err


def assert_close(x, ref, tol=RTOL, what=""):
    err = rel_err(x, ref)
    assert err <= tol, "%s: relative error %.3e > %.1e" % (what, err, tol)
    return err


def ranges_from_json(text):
    return {k: range(v[0], v[1]) for k, v in json.loads(str(text)).items()}
