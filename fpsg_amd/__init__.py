"""fpsg_amd -- MI355X-native hot path of voidstrike/FPSG.

HIP kernels + C ABI live in ``fpsg_amd/csrc`` (built to ``fpsg_amd/libfpsg_hip.so``); the
modules here mirror the reference's Python interface for the same path.
"""
import os as _os

# Kernel arguments in device memory: the command processor then reads a launch's arguments from HBM instead of fetching
# them from host memory across PCIe for every dispatch.  The c5 step is ~350 dependent launches per episode; measured
# same box, alternating: 45.41 -> 46.40 episodes/s (+2.2 %, profiles/r05/hip_force_dev_kernarg_ab.txt).  The HIP runtime
# reads the variable when it initialises (first HIP call), so it is set here, at import, unless the caller already chose.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

__version__ = "0.1.0"
