"""mpcasm: MI355X-native batched QP assembly behind the mpc_interface API.

Sub-modules (imported on demand; importing this package alone touches neither
torch nor the GPU):

* ``mpcasm.capi``     ctypes binding of the C-ABI library (include/mpcasm.h)
* ``mpcasm.plan``     plan compiler: Formulation structure -> flat device tables
* ``mpcasm.engine``   batched assembler on torch-ROCm buffers, sharding helpers
* ``mpcasm.problems`` builders of the BASELINE configurations
"""
