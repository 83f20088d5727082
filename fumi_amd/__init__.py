"""fumi_amd: MI355X-native episodic meta-training engine for the FuMI / MAML / AM3 hot path of s-a-malik/fumi.

Layout
  csrc/     hand-written HIP kernels for gfx950 + the C ABI (include/fumi_hip.h) -> lib/libfumi_hip.so
  hip.py    ctypes binding (raw device pointers; PyTorch only owns memory and streams)
  models/   host-side mirror of the reference's nn.Module surface (FUMI, PureImageNetwork, AM3)
  utils/    host-side mirror of the reference's flag parser / factories / meters
  main.py   the reference's CLI
"""
__version__ = "0.1.0"
