"""mpcasm: MI355X-native batched QP assembly behind the mpc_interface API.

Sub-modules (imported on demand; importing this package alone touches neither
torch nor the GPU):

* ``mpcasm.capi``     ctypes binding of the C-ABI library (include/mpcasm.h)
* ``mpcasm.plan``     plan compiler: Formulation structure -> flat device tables
* ``mpcasm.engine``   batched assembler on torch-ROCm buffers, sharding helpers
* ``mpcasm.problems`` builders of the BASELINE configurations
"""

import os as _os

# Kernel arguments in device memory.  The HIP runtime keeps them in host memory unless told otherwise, and
# then the first thing every launch does -- read its own arguments -- is a trip across PCIe: 2.5 us of the 27 us
# of a C2 launch at B = 4096 (measured both ways).  The runtime reads the setting once, when
# it initialises (the first HIP call of the process, e.g. torch's first use of the GPU): importing this package
# before that is enough; a process that has already initialised HIP keeps what it had.  An explicit setting of
# the user's wins.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
