"""fpsg_amd -- MI355X-native hot path of voidstrike/FPSG.

HIP kernels + C ABI live in ``fpsg_amd/csrc`` (built to ``fpsg_amd/libfpsg_hip.so``); the
modules here mirror the reference's Python interface for the same path.
"""
__version__ = "0.1.0"
