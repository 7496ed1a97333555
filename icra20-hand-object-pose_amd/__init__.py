"""MI355X-native hot path of wenbowen123/icra20-hand-object-pose.

`hop_amd.synth`  seeded synthetic inputs
`hop_amd.api`    ctypes binding of the C-ABI in include/hop.h (libhop.so, HIP) + host mirror classes
"""
from . import synth  # noqa: F401
