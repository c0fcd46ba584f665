"""Repository-level pytest configuration: make the in-tree packages importable
(`mpc_interface`, `mpcasm` live under `mpc-interface_amd/`; `oracle` at the root)."""
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
for path in (os.path.join(_ROOT, "mpc-interface_amd"), _ROOT):
    if path not in sys.path:
        sys.path.insert(0, path)
