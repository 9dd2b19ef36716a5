"""curvecloudnet_amd: MI355X-native implementation of the CurveCloudNet curve-aggregation hot path.

Host side: Python on PyTorch-ROCm (device memory, streams, torch.distributed).  All arithmetic on
the path runs in hand-written HIP kernels for gfx950 behind the C-ABI of ``libccn_hip.so``
(``include/ccn_hip.h``).  There is no CPU fallback: ops raise if the library or a GPU is missing.
"""
__version__ = "0.1.0"
