"""ctypes binding of libsrx.so (include/srx.h)."""
import ctypes as C
import os
import subprocess
import weakref

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ASSETS_DIR = os.path.join(HERE, "Assets")
FLAG_KEEP_TENSORS = 1
FLAG_VERBOSE_WEIGHTS = 2
FLAG_GENERIC_LAYERS = 4
FLAG_FP32 = 8
FLAG_NHWC = 16
TENSOR_NAMES = ["input", "conv1", "conv2", "conv3", "conv_up1", "conv4", "conv5", "conv6", "output"]

# every symbol include/srx.h declares
SYMBOLS = ["srx_create", "srx_destroy", "srx_get_out_size", "srx_get_in_size", "srx_upload_rgba8",
           "srx_upload_rgb8", "srx_input_device_ptr", "srx_output_device_ptr", "srx_image_to_tensor",
           "srx_process", "srx_tensor_to_image", "srx_upscale", "srx_sync", "srx_read_tensor",
           "srx_tensor_shape", "srx_fold_weights", "srx_packed_weights", "srx_num_stages", "srx_stage_name",
           "srx_launch_stage", "srx_frames_per_launch", "srx_device_info", "srx_last_error", "srx_upscale_frames",
           "srx_pipe_create", "srx_pipe_destroy", "srx_pipe_slots", "srx_pipe_input", "srx_pipe_submit",
           "srx_pipe_wait", "srx_layer_conv2d", "srx_layer_upsample2d", "srx_layer_add", "srx_fused_weights", "srx_pipe_set_channels", "srx_layer_plan", "srx_stage_issued_flops",
           "srx_upscale_static", "srx_static_scratch_bytes"]


class SrxError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"srx status {status}: {message}")
        self.status = status


def lib_path():
    # SRX_LIB names another build of the library (A/B measurements)
    return os.environ.get("SRX_LIB") or os.path.join(HERE, "libsrx.so")


def default_weights_path():
    return os.path.join(ASSETS_DIR, "weights.bin")


def build_library():
    """Compile libsrx.so in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
    subprocess.check_call(["make", "-C", os.path.join(HERE, "csrc")])


_lib = None


def lib():
    """Load libsrx.so.  Raises if it has not been built: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise SrxError(-1, f"{path} is missing: run __graft_entry__.build() (make -C superresolution_amd/csrc)")
    try:
        # torch ships its own libamdhip64.so.7; importing it first makes this library share that
        # runtime (same SONAME) instead of loading a second HIP runtime into the process.
        import torch  # noqa: F401
    except Exception:
        pass
    _lib = load_library(path)
    return _lib


def load_library(path):
    """Load and type one build of libsrx.so (A/B timing loads several builds into one process)."""
    L = C.CDLL(path)
    vp, u32, i32, sz = C.c_void_p, C.c_uint32, C.c_int32, C.c_size_t
    L.srx_create.argtypes = [C.POINTER(vp), C.c_int, C.c_char_p, u32, u32, u32, u32]
    L.srx_destroy.argtypes = [vp]
    L.srx_get_out_size.argtypes = [vp, C.POINTER(u32), C.POINTER(u32)]
    L.srx_get_in_size.argtypes = [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]
    L.srx_upload_rgba8.argtypes = [vp, u32, vp, sz, vp]
    L.srx_upload_rgb8.argtypes = [vp, u32, vp, sz, vp]
    L.srx_input_device_ptr.argtypes = [vp, u32, C.POINTER(vp), C.POINTER(sz)]
    L.srx_output_device_ptr.argtypes = [vp, u32, C.POINTER(vp), C.POINTER(sz)]
    L.srx_image_to_tensor.argtypes = [vp, vp]
    L.srx_process.argtypes = [vp, vp]
    L.srx_tensor_to_image.argtypes = [vp, u32, vp, sz, vp]
    L.srx_upscale.argtypes = [vp, vp]
    L.srx_sync.argtypes = [vp, vp]
    L.srx_read_tensor.argtypes = [vp, u32, i32, vp, sz]
    L.srx_tensor_shape.argtypes = [vp, i32, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]
    L.srx_fold_weights.argtypes = [C.c_char_p, i32, vp, sz, vp, sz, C.POINTER(sz), C.POINTER(sz)]
    L.srx_packed_weights.argtypes = [C.c_char_p, i32, vp, sz, C.POINTER(sz)]
    L.srx_fused_weights.argtypes = [C.c_char_p, i32, vp, sz, C.POINTER(sz)]
    L.srx_num_stages.argtypes = [vp]
    L.srx_stage_name.argtypes = [vp, i32]
    L.srx_stage_name.restype = C.c_char_p
    L.srx_launch_stage.argtypes = [vp, i32, vp]
    L.srx_stage_issued_flops.argtypes = [vp, i32, C.POINTER(C.c_double)]
    L.srx_last_error.restype = C.c_char_p
    L.srx_frames_per_launch.argtypes = [vp]
    L.srx_device_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.c_char_p, C.c_size_t]
    L.srx_upscale_frames.argtypes = [vp, u32, u32, vp]
    L.srx_upscale_static.argtypes = [vp, vp, sz, vp, sz, u32, u32, vp]
    L.srx_static_scratch_bytes.argtypes = []
    L.srx_pipe_create.argtypes = [vp, u32, C.POINTER(vp)]
    L.srx_pipe_destroy.argtypes = [vp]
    L.srx_pipe_slots.argtypes = [vp]
    L.srx_pipe_input.argtypes = [vp, u32, C.POINTER(vp), C.POINTER(sz), C.POINTER(sz)]
    L.srx_pipe_submit.argtypes = [vp, u32]
    L.srx_pipe_set_channels.argtypes = [vp, u32, u32]
    L.srx_pipe_wait.argtypes = [vp, u32, C.POINTER(vp), C.POINTER(sz), C.POINTER(sz)]
    u32x4 = C.POINTER(u32 * 4)
    L.srx_layer_conv2d.argtypes = [i32, i32, vp, u32x4, vp, u32x4, vp, vp, vp]
    L.srx_layer_upsample2d.argtypes = [i32, i32, vp, u32x4, u32, u32, i32, vp, vp]
    L.srx_layer_add.argtypes = [i32, vp, vp, vp, sz, vp]
    L.srx_layer_plan.argtypes = [i32, i32, u32x4, u32x4, u32x4, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    for name in SYMBOLS:
        if name not in ("srx_stage_name", "srx_last_error", "srx_frames_per_launch", "srx_pipe_slots", "srx_static_scratch_bytes"):
            getattr(L, name).restype = i32
    L.srx_frames_per_launch.restype = u32
    L.srx_static_scratch_bytes.restype = sz
    L.srx_pipe_slots.restype = u32
    return L


def _check(status, library=None):
    if status != 0:
        raise SrxError(status, (library or lib()).srx_last_error().decode(errors="replace"))


class SuperResolutionContext:
    """One GPU's worth of the path: weights, workspace, input and output images in HBM."""

    def __init__(self, width, height, batch=1, device=0, weights_path=None, flags=0, library=None):
        self._h = C.c_void_p()
        self._L = library or lib()
        _check(self._L.srx_create(C.byref(self._h), device, (weights_path or default_weights_path()).encode(),
                                  width, height, batch, flags))
        self.width, self.height, self.batch, self.flags = width, height, batch, flags
        self._pipes = weakref.WeakSet()

    def close(self):
        if getattr(self, "_h", None):
            # a pipe holds streams and pinned buffers that refer to this ctx: it goes first
            for pipe in list(getattr(self, "_pipes", ())):
                pipe.close()
            _check(self._L.srx_destroy(self._h))
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def out_size(self):
        w, h = C.c_uint32(), C.c_uint32()
        _check(self._L.srx_get_out_size(self._h, C.byref(w), C.byref(h)))
        return w.value, h.value

    def upload_rgba8(self, frame, rgba, stream=None):
        rgba = np.ascontiguousarray(rgba, np.uint8)
        assert rgba.shape == (self.height, self.width, 4), rgba.shape
        _check(self._L.srx_upload_rgba8(self._h, frame, rgba.ctypes.data, self.width * 4, stream))
        _check(self._L.srx_sync(self._h, stream))

    def upload_rgb8(self, frame, rgb, stream=None):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        assert rgb.shape == (self.height, self.width, 3), rgb.shape
        _check(self._L.srx_upload_rgb8(self._h, frame, rgb.ctypes.data, self.width * 3, stream))

    def input_device_ptr(self, frame):
        p, pitch = C.c_void_p(), C.c_size_t()
        _check(self._L.srx_input_device_ptr(self._h, frame, C.byref(p), C.byref(pitch)))
        return p.value, pitch.value

    def output_device_ptr(self, frame):
        p, pitch = C.c_void_p(), C.c_size_t()
        _check(self._L.srx_output_device_ptr(self._h, frame, C.byref(p), C.byref(pitch)))
        return p.value, pitch.value

    def image_to_tensor(self, stream=None):
        _check(self._L.srx_image_to_tensor(self._h, stream))

    def process(self, stream=None):
        _check(self._L.srx_process(self._h, stream))

    def upscale(self, stream=None):
        _check(self._L.srx_upscale(self._h, stream))

    def upscale_frames(self, first, count, stream=None):
        _check(self._L.srx_upscale_frames(self._h, first, count, stream))

    def sync(self, stream=None):
        _check(self._L.srx_sync(self._h, stream))

    def upscale_static(self, in_ptr, in_pitch, out_ptr, out_pitch, width, height, stream=None):
        """srx_upscale_static on caller-owned device images (addresses as integers): allocates nothing."""
        _check(self._L.srx_upscale_static(self._h, in_ptr, in_pitch, out_ptr, out_pitch, width, height, stream))

    def tensor_to_image(self, frame=0, stream=None):
        w, h = self.out_size
        out = np.empty((h, w, 4), np.uint8)
        _check(self._L.srx_tensor_to_image(self._h, frame, out.ctypes.data, w * 4, stream))
        _check(self._L.srx_sync(self._h, stream))
        return out

    def tensor_shape(self, tid):
        c, h, w = C.c_uint32(), C.c_uint32(), C.c_uint32()
        _check(self._L.srx_tensor_shape(self._h, tid, C.byref(c), C.byref(h), C.byref(w)))
        return c.value, h.value, w.value

    def read_tensor(self, tid, frame=0):
        shape = self.tensor_shape(tid)
        out = np.empty(shape, np.float32 if self.flags & FLAG_FP32 else np.float16)
        _check(self._L.srx_read_tensor(self._h, frame, tid, out.ctypes.data, out.nbytes))
        return out

    def stage_names(self):
        return [self._L.srx_stage_name(self._h, i).decode() for i in range(self._L.srx_num_stages(self._h))]

    @property
    def frames_per_launch(self):
        return self._L.srx_frames_per_launch(self._h)

    def device_info(self):
        """{'compute_units', 'clock_khz', 'arch'} of the ctx's GPU."""
        cus, khz = C.c_int32(0), C.c_int32(0)
        arch = C.create_string_buffer(64)
        _check(self._L.srx_device_info(self._h, C.byref(cus), C.byref(khz), arch, 64))
        return {"compute_units": cus.value, "clock_khz": khz.value, "arch": arch.value.decode()}

    def launch_stage(self, i, stream=None):
        _check(self._L.srx_launch_stage(self._h, i, stream))

    def stage_issued_flops(self, i):
        """Matrix-instruction FLOPs one launch of fused stage i issues, padding included."""
        v = C.c_double()
        _check(self._L.srx_stage_issued_flops(self._h, i, C.byref(v)))
        return v.value

    def run(self, rgba, frame=0):
        """Upload one RGBA8 frame, upscale, return the RGBA8 result (uint8 [2H,2W,4])."""
        self.upload_rgba8(frame, rgba)
        self.upscale()
        return self.tensor_to_image(frame)


class FramePipe:
    """Frames in flight between host memory and one ctx: pinned buffers per slot, three streams."""

    def __init__(self, ctx, frames_per_slot, in_channels=4, out_channels=4):
        self._ctx, self._L, self._h = ctx, ctx._L, C.c_void_p()
        _check(self._L.srx_pipe_create(ctx._h, frames_per_slot, C.byref(self._h)))
        self.frames_per_slot = frames_per_slot
        self.slots = int(self._L.srx_pipe_slots(self._h))
        self.in_channels, self.out_channels = in_channels, out_channels
        ctx._pipes.add(self)
        if (in_channels, out_channels) != (4, 4):
            _check(self._L.srx_pipe_set_channels(self._h, in_channels, out_channels))

    def close(self):
        if getattr(self, "_h", None):
            self._L.srx_pipe_destroy(self._h)
            self._h = None
            self._ctx._pipes.discard(self)

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _view(self, ptr, nbytes):
        # The array's base object refers to the pipe, so an array that is still referenced keeps the
        # pipe (and its pinned memory) from being collected.  An explicit close() still frees the
        # memory under any view: views are for use while the pipe is open.
        buf = (C.c_uint8 * nbytes).from_address(ptr)
        buf._pipe = self
        return np.frombuffer(buf, dtype=np.uint8)

    def input(self, slot):
        """The slot's pinned input frames as a writable uint8 array [frames_per_slot, H, W, channels]:
        a view of pinned memory, valid until close()."""
        ptr, pitch, stride = C.c_void_p(), C.c_size_t(), C.c_size_t()
        _check(self._L.srx_pipe_input(self._h, slot, C.byref(ptr), C.byref(pitch), C.byref(stride)))
        return self._view(ptr.value, self.frames_per_slot * stride.value).reshape(
            self.frames_per_slot, self._ctx.height, self._ctx.width, self.in_channels)

    def submit(self, slot):
        _check(self._L.srx_pipe_submit(self._h, slot))

    def wait(self, slot, copy=False):
        """Block until the slot is done; returns its output frames [frames_per_slot, 2H, 2W, channels].
        By default the pinned buffer itself (what srx_pipe_wait hands a C caller: no host copy), valid
        until the slot is submitted again or the pipe is closed; copy=True returns a copy that outlives
        both, at the price of one host memcpy of the slot (25 MB per 1080p RGB frame)."""
        ptr, pitch, stride = C.c_void_p(), C.c_size_t(), C.c_size_t()
        _check(self._L.srx_pipe_wait(self._h, slot, C.byref(ptr), C.byref(pitch), C.byref(stride)))
        out = self._view(ptr.value, self.frames_per_slot * stride.value).reshape(
            self.frames_per_slot, 2 * self._ctx.height, 2 * self._ctx.width, self.out_channels)
        return out.copy() if copy else out


def _sizes(v):
    return (C.c_uint32 * 4)(*[int(x) for x in v])


def layer_plan(kind, dtype, in_sizes, param, in_bytes=0, out_bytes=0):
    """Host-only: (out_sizes, in_bytes, out_bytes) the layer factory reports (kind 0 conv, 1 upsample, 2 add)."""
    out = (C.c_uint32 * 4)()
    ib, ob = C.c_uint64(in_bytes), C.c_uint64(out_bytes)
    _check(lib().srx_layer_plan(kind, dtype, C.byref(_sizes(in_sizes)), C.byref(_sizes(param)), C.byref(out),
                                C.byref(ib), C.byref(ob)))
    return tuple(out), ib.value, ob.value


def layer_conv2d(dtype, layout, in_ptr, in_sizes, filter_ptr, filter_sizes, bias_ptr, out_ptr, stream=None):
    """ML::Util::CreateConvolutionLayer + dispatch on device pointers (dtype 0 fp16 / 1 fp32, layout 0 NCHW / 1 NHWC)."""
    _check(lib().srx_layer_conv2d(dtype, layout, in_ptr, C.byref(_sizes(in_sizes)), filter_ptr,
                                  C.byref(_sizes(filter_sizes)), bias_ptr, out_ptr, stream))


def layer_upsample2d(dtype, layout, in_ptr, in_sizes, scale_x, scale_y, interpolation, out_ptr, stream=None):
    _check(lib().srx_layer_upsample2d(dtype, layout, in_ptr, C.byref(_sizes(in_sizes)), scale_x, scale_y,
                                      interpolation, out_ptr, stream))


def layer_add(dtype, a_ptr, b_ptr, out_ptr, count, stream=None):
    _check(lib().srx_layer_add(dtype, a_ptr, b_ptr, out_ptr, count, stream))


def fold_weights(layer, weights_path=None):
    """Host-only: folded + truncated fp16 (filter OIHW, bias or None) of conv layer 0..6."""
    L = lib()
    path = (weights_path or default_weights_path()).encode()
    fn, bn = C.c_size_t(), C.c_size_t()
    _check(L.srx_fold_weights(path, layer, None, 0, None, 0, C.byref(fn), C.byref(bn)))
    f = np.empty(fn.value, np.uint16)
    b = np.empty(bn.value, np.uint16)
    _check(L.srx_fold_weights(path, layer, f.ctypes.data, f.size, b.ctypes.data if bn.value else None, b.size,
                              None, None))
    return f, (b if bn.value else None)


def fused_weights(layer, weights_path=None):
    """Host-only: the fragments of conv layer 0..6 as the fused kernels hold them."""
    L = lib()
    path = (weights_path or default_weights_path()).encode()
    n = C.c_size_t()
    _check(L.srx_fused_weights(path, layer, None, 0, C.byref(n)))
    out = np.empty(n.value, np.uint16)
    _check(L.srx_fused_weights(path, layer, out.ctypes.data, out.size, None))
    return out


def packed_weights(layer, weights_path=None):
    """Host-only: the MFMA fragment layout [tile][kstep][64][8] of conv layer 0..6."""
    L = lib()
    path = (weights_path or default_weights_path()).encode()
    n = C.c_size_t()
    _check(L.srx_packed_weights(path, layer, None, 0, C.byref(n)))
    out = np.empty(n.value, np.uint16)
    _check(L.srx_packed_weights(path, layer, out.ctypes.data, out.size, None))
    return out
