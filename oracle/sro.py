"""ctypes view of the CPU oracle (oracle/libsroracle.so) and of oracle/_ref.

TEST INFRASTRUCTURE ONLY.  Importable from tests/, from __graft_entry__.smoke()
and from the cpu_baseline leg of bench.py; the shipped package never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libsroracle.so")
REF_PATH = os.path.join(HERE, "_ref", "libsrref.so")

T_INPUT, T_CONV1, T_CONV2, T_CONV3, T_CONV_UP1, T_CONV4, T_CONV5, T_CONV6, T_OUTPUT = range(9)
TENSOR_NAMES = ["input", "conv1", "conv2", "conv3", "conv_up1", "conv4", "conv5", "conv6", "output"]


def build(force=False):
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    if force or not os.path.exists(LIB_PATH) or \
            os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(HERE, "sr_oracle.c")):
        subprocess.check_call(["make", "-C", HERE, "libsroracle.so"])
    if os.path.isdir("/root/reference/SuperResolution/Common"):
        subprocess.check_call(["make", "-C", HERE, "ref"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.sro_f16_compress.restype = C.c_uint16
        L.sro_f16_compress.argtypes = [C.c_float]
        L.sro_f16_compress_array.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.sro_f32_to_f16_rne.restype = C.c_uint16
        L.sro_f32_to_f16_rne.argtypes = [C.c_float]
        L.sro_f16_to_f32.restype = C.c_float
        L.sro_f16_to_f32.argtypes = [C.c_uint16]
        L.sro_load_weights.restype = C.c_int
        L.sro_load_weights.argtypes = [C.c_char_p, C.POINTER(C.c_void_p), C.c_int]
        L.sro_free_weights.argtypes = [C.c_void_p]
        L.sro_weights_count.restype = C.c_int
        L.sro_weights_count.argtypes = [C.c_void_p]
        L.sro_weights_name.restype = C.c_char_p
        L.sro_weights_name.argtypes = [C.c_void_p, C.c_int]
        L.sro_weights_len.restype = C.c_uint32
        L.sro_weights_len.argtypes = [C.c_void_p, C.c_int]
        L.sro_weights_data.restype = C.POINTER(C.c_float)
        L.sro_weights_data.argtypes = [C.c_void_p, C.c_int]
        L.sro_create_weight_tensors.restype = C.c_int
        L.sro_create_weight_tensors.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p,
                                                C.POINTER(C.c_uint32), C.c_int, C.c_int,
                                                C.c_void_p, C.c_void_p]
        L.sro_get_strides.argtypes = [C.POINTER(C.c_uint32), C.c_int, C.POINTER(C.c_uint32)]
        L.sro_model_create.restype = C.c_void_p
        L.sro_model_create.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.sro_model_destroy.argtypes = [C.c_void_p]
        L.sro_run.restype = C.c_int
        L.sro_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_void_p, C.c_int]
        L.sro_tensor.restype = C.POINTER(C.c_float)
        L.sro_tensor.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                 C.POINTER(C.c_int)]
        L.sro_image_to_tensor.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_void_p]
        L.sro_tensor_to_image.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _lib = L
    return _lib


_ref = None


def ref():
    """oracle/_ref/libsrref.so (reference sources compiled where they lie) or None."""
    global _ref
    if _ref is None and os.path.exists(REF_PATH):
        R = C.CDLL(REF_PATH)
        R.ref_f16_compress.restype = C.c_uint16
        R.ref_f16_compress.argtypes = [C.c_float]
        R.ref_f16_compress_array.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        R.ref_load_image.restype = C.c_void_p
        R.ref_load_image.argtypes = [C.c_char_p] + [C.POINTER(C.c_int)] * 4
        R.ref_free.argtypes = [C.c_void_p]
        _ref = R
    return _ref


LAYERS = [
    ("conv1/weights", "conv1/BatchNorm/scale", "conv1/BatchNorm/shift", (32, 3, 5, 5)),
    ("conv2/weights", "conv2/BatchNorm/scale", "conv2/BatchNorm/shift", (64, 32, 3, 3)),
    ("conv3/weights", "conv3/BatchNorm/scale", "conv3/BatchNorm/shift", (64, 64, 3, 3)),
    ("conv_up1/conv/weights", "conv_up1/conv/BatchNorm/scale", "conv_up1/conv/BatchNorm/shift", (32, 64, 5, 5)),
    ("conv4/weights", "conv4/BatchNorm/scale", "conv4/BatchNorm/shift", (32, 32, 3, 3)),
    ("conv5/weights", "conv5/BatchNorm/scale", "conv5/BatchNorm/shift", (32, 32, 3, 3)),
    ("conv6/weights", None, None, (3, 32, 3, 3)),
]


class Weights:
    def __init__(self, path, verbose=False):
        self._h = C.c_void_p()
        rc = lib().sro_load_weights(path.encode(), C.byref(self._h), int(verbose))
        if rc != 0:
            raise RuntimeError("loadWeights")      # SuperResolution.cpp:185-186 throws on failure

    def __del__(self):
        if getattr(self, "_h", None):
            lib().sro_free_weights(self._h)
            self._h = None

    def tensors(self):
        L = lib()
        out = {}
        for i in range(L.sro_weights_count(self._h)):
            n = L.sro_weights_len(self._h, i)
            out[L.sro_weights_name(self._h, i).decode()] = np.ctypeslib.as_array(
                L.sro_weights_data(self._h, i), shape=(n,)).copy()
        return out

    def create_weight_tensors(self, conv, scale, shift, sizes, layout=0, dtype=0):
        n = int(np.prod(sizes))
        dt = np.uint16 if dtype == 0 else np.float32
        f = np.zeros(n, dt)
        b = np.zeros(sizes[0], dt)
        s = (C.c_uint32 * 4)(*sizes)
        rc = lib().sro_create_weight_tensors(self._h, conv.encode(),
                                             scale.encode() if scale else None,
                                             shift.encode() if shift else None, s, layout, dtype,
                                             f.ctypes.data, b.ctypes.data)
        if rc < 0:
            raise KeyError(conv)
        return f, (b if rc == 1 else None)


class Model:
    def __init__(self, weights, dtype=0, up1_mode=0):
        self._w = weights
        self._h = lib().sro_model_create(weights._h, dtype, up1_mode)
        if not self._h:
            raise RuntimeError("model create failed")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().sro_model_destroy(self._h)
            self._h = None

    def run(self, rgba, keep_tensors=False):
        """rgba: uint8 [H,W,4] -> uint8 [2H,2W,4]"""
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
        h, w, ch = rgba.shape
        assert ch == 4
        out = np.empty((2 * h, 2 * w, 4), np.uint8)
        rc = lib().sro_run(self._h, rgba.ctypes.data, w, h, w * 4, out.ctypes.data, int(keep_tensors))
        if rc != 0:
            raise RuntimeError("sro_run failed")
        return out

    def tensor(self, tid):
        """float32 [C,H,W] copy of a kept tensor"""
        c, h, w = C.c_int(), C.c_int(), C.c_int()
        p = lib().sro_tensor(self._h, tid, C.byref(c), C.byref(h), C.byref(w))
        if not p:
            return None
        return np.ctypeslib.as_array(p, shape=(c.value, h.value, w.value)).copy()


def f16_compress(x):
    x = np.ascontiguousarray(x, np.float32).ravel()
    out = np.empty(x.size, np.uint16)
    lib().sro_f16_compress_array(x.ctypes.data, out.ctypes.data, x.size)
    return out


def load_rgba(path):
    """Decode like XUSGTextureLoader.h:17-32: 3-channel files become RGBA (alpha 255)."""
    from PIL import Image
    im = Image.open(path)
    if im.mode not in ("RGB", "RGBA"):
        im = im.convert("RGBA")
    a = np.asarray(im.convert("RGBA"), dtype=np.uint8)
    return np.ascontiguousarray(a)
