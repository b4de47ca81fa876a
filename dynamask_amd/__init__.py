"""dynamask_amd -- MI355X-native (gfx950) DynaMask dynamic mask-head hot path.

Layout:
  csrc/                 HIP kernels + the C ABI (include/dynamask_hip.h)
  _lib.py, ops.py       ctypes binding of libdynamask_hip.so over torch device memory
  registry.py           mmdet-style HEADS / ROI_EXTRACTORS / LOSSES registries + Config
  roi_extractors.py, mask_heads.py, losses.py, roi_head.py
                        host-side mirror of the reference's plugin classes
  synth.py              synthetic workload
  dist.py               RCCL gradient all-reduce of the mask-head parameters

There is no CPU or eager-PyTorch fallback: every operator raises if
libdynamask_hip.so is missing or its input is not on a HIP device.
"""
__version__ = '0.1.0'
