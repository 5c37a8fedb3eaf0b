"""MI355X-native 2x super-resolution: Python view of the C ABI in include/srx.h.

The compute lives in superresolution_amd/libsrx.so (HIP kernels for gfx950 +
C ABI); this module only binds it with ctypes.  There is no CPU fallback: if the
library is missing or the device is not a gfx950 GPU, calls raise.
"""
from ._capi import (SrxError, SuperResolutionContext, FramePipe, lib, lib_path, ASSETS_DIR, default_weights_path,
                    FLAG_KEEP_TENSORS, FLAG_VERBOSE_WEIGHTS, FLAG_GENERIC_LAYERS, FLAG_FP32, FLAG_NHWC,
                    layer_conv2d, layer_upsample2d, layer_add, layer_plan, TENSOR_NAMES, build_library,
                    fold_weights, packed_weights, fused_weights)

__all__ = ["SrxError", "SuperResolutionContext", "FramePipe", "lib", "lib_path", "ASSETS_DIR", "default_weights_path",
           "FLAG_KEEP_TENSORS", "FLAG_VERBOSE_WEIGHTS", "FLAG_GENERIC_LAYERS", "FLAG_FP32", "FLAG_NHWC",
           "layer_conv2d", "layer_upsample2d", "layer_add", "layer_plan", "TENSOR_NAMES", "build_library", "fold_weights", "packed_weights", "fused_weights"]
