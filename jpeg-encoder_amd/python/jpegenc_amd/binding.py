"""ctypes binding of libjpegenc_hip (the C ABI declared in include/jpegenc_hip.h).

There is no CPU fallback: if the shared library is missing or no GPU is usable the calls
raise.  Device pointers are plain integers (e.g. ``torch.Tensor.data_ptr()``).
"""
import ctypes as C
import os
import sys

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.abspath(os.path.join(_PKG, "..", "..", "lib", "libjpegenc_hip.so"))

OK = 0
FILTER_SEPIA, FILTER_MULTIPLY = 0, 1
ERR_INVALID_ARGUMENT, ERR_HIP, ERR_BUFFER_TOO_SMALL, ERR_BAD_STATE, ERR_NO_DEVICE, ERR_PEER_FAILED = -1, -2, -3, -4, -5, -6
RGB8, RGBF32 = 0, 1
STAGE_YCBCR, STAGE_SUBSAMPLE, STAGE_ALL = 1, 2, 3
DC, AC = 0, 1

EXPORTS = [
    "jpeg_hip_create", "jpeg_hip_destroy", "jpeg_hip_last_error", "jpeg_hip_status_string", "jpeg_hip_stream",
    "jpeg_hip_wait_for_stream", "jpeg_hip_stream_wait_for_ctx", "jpeg_hip_allreduce_status",
    "jpeg_hip_compute_layout", "jpeg_hip_encode", "jpeg_hip_encode_static", "jpeg_hip_static_workspace_bytes", "jpeg_hip_encode_host", "jpeg_hip_encode_batch", "jpeg_hip_encode_batch_host", "jpeg_hip_begin", "jpeg_hip_upload_pixels", "jpeg_hip_upload_planes",
    "jpeg_hip_bind_pixels", "jpeg_hip_transform", "jpeg_hip_histograms", "jpeg_hip_histograms_host",
    "jpeg_hip_huffman_from_histogram", "jpeg_hip_huffman_from_histogram_limit", "jpeg_hip_standard_huffman", "jpeg_hip_write_dht", "jpeg_hip_write_headers", "jpeg_hip_set_huffman",
    "jpeg_hip_build_huffman", "jpeg_hip_get_huffman", "jpeg_hip_emit_scan", "jpeg_hip_emit_scan_host",
    "jpeg_hip_get_coefficients",
    "jpeg_hip_coefficients_device", "jpeg_hip_encode_sharded", "jpeg_hip_allreduce_histograms", "jpeg_hip_comm_unique_id", "jpeg_hip_comm_create", "jpeg_hip_comm_destroy", "jpeg_hip_analyze_band", "jpeg_hip_finish_band", "jpeg_hip_finish_band_start", "jpeg_hip_finish_band_wait", "jpeg_hip_set_profiling",
    "jpeg_hip_get_timings", "jpeg_hip_get_host_timeline", "jpeg_hip_get_profile_sums", "jpeg_hip_decode_info", "jpeg_hip_decode", "jpeg_hip_decode_host",
    "jpeg_hip_decoded_coefficients", "jpeg_hip_psnr", "jpeg_hip_synthetic_images", "jpeg_hip_filter_pixels", "jpeg_hip_get_pixels", "jpeg_hip_channel_gain",
]


class Params(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("sampling", C.c_int32 * 12),
                ("qtable", (C.c_float * 64) * 3), ("restart_interval", C.c_int32), ("pad_ones", C.c_int32),
                ("mcu_row_begin", C.c_int32), ("mcu_row_end", C.c_int32),
                ("standard_huffman", C.c_int32)]


class Layout(C.Structure):
    _fields_ = [("padded_width", C.c_int32), ("padded_height", C.c_int32), ("mcu_width", C.c_int32),
                ("mcu_height", C.c_int32), ("mcus_per_row", C.c_int32), ("mcu_rows", C.c_int32),
                ("band_mcu_rows", C.c_int32), ("h_factor", C.c_int32 * 3), ("v_factor", C.c_int32 * 3),
                ("blocks_per_mcu", C.c_int32), ("blocks", C.c_int64 * 3), ("total_blocks", C.c_int64),
                ("intervals", C.c_int64), ("max_jpeg_bytes", C.c_int64)]


class ImageInfo(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("h_samp", C.c_int32 * 3), ("v_samp", C.c_int32 * 3),
                ("restart_interval", C.c_int32), ("total_blocks", C.c_int64)]


class JpegHipError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("jpegenc_hip status %d (%s): %s" % (status, status_string(status), message))
        self.status = status


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libjpegenc_hip.so is not built (%s); run __graft_entry__.build()" % LIB_PATH)
        # PyTorch ships its own libamdhip64 under the same SONAME.  A process can only run
        # one HIP runtime, so let torch load its copy first; the encoder library then binds
        # to that one instead of pulling in a second runtime from /opt/rocm.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
        L.jpeg_hip_create.argtypes = [i32, C.POINTER(vp)]
        L.jpeg_hip_destroy.argtypes = [vp]
        L.jpeg_hip_destroy.restype = None
        L.jpeg_hip_last_error.argtypes = [vp]
        L.jpeg_hip_last_error.restype = C.c_char_p
        L.jpeg_hip_status_string.argtypes = [i32]
        L.jpeg_hip_status_string.restype = C.c_char_p
        L.jpeg_hip_stream.argtypes = [vp]
        L.jpeg_hip_stream.restype = vp
        L.jpeg_hip_wait_for_stream.argtypes = [vp, vp]
        L.jpeg_hip_stream_wait_for_ctx.argtypes = [vp, vp]
        L.jpeg_hip_allreduce_status.argtypes = [vp, vp, i32, C.POINTER(i32)]
        L.jpeg_hip_compute_layout.argtypes = [C.POINTER(Params), C.POINTER(Layout)]
        L.jpeg_hip_encode.argtypes = [vp, vp, i32, i64, C.POINTER(Params), vp, i64, C.POINTER(i64)]
        L.jpeg_hip_encode_host.argtypes = [vp, vp, i32, i64, C.POINTER(Params), vp, i64, C.POINTER(i64)]
        L.jpeg_hip_encode_static.argtypes = [vp, vp, i32, i64, C.POINTER(Params), vp, i64, C.POINTER(i64)]
        L.jpeg_hip_static_workspace_bytes.argtypes = []
        L.jpeg_hip_static_workspace_bytes.restype = i64
        L.jpeg_hip_encode_batch.argtypes = [vp, vp, i32, i64, i64, i32, C.POINTER(Params), vp, i64, vp]
        L.jpeg_hip_encode_batch_host.argtypes = [vp, vp, i32, i64, i64, i32, C.POINTER(Params), vp, i64, vp]
        L.jpeg_hip_begin.argtypes = [vp, C.POINTER(Params)]
        L.jpeg_hip_decode_info.argtypes = [vp, i64, C.POINTER(ImageInfo)]
        L.jpeg_hip_decode.argtypes = [vp, vp, i64, vp, i64, i64, C.POINTER(ImageInfo)]
        L.jpeg_hip_decode_host.argtypes = [vp, vp, i64, vp, i64, i64, C.POINTER(ImageInfo)]
        L.jpeg_hip_decoded_coefficients.argtypes = [vp, vp, i64]
        L.jpeg_hip_psnr.argtypes = [vp, vp, i64, vp, i64, i32, i32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.jpeg_hip_upload_pixels.argtypes = [vp, vp, i32, i64]
        L.jpeg_hip_upload_planes.argtypes = [vp, vp, vp, vp, i64]
        L.jpeg_hip_bind_pixels.argtypes = [vp, vp, i32, i64]
        L.jpeg_hip_transform.argtypes = [vp, i32]
        L.jpeg_hip_filter_pixels.argtypes = [vp, i32, i32, C.c_float]
        L.jpeg_hip_channel_gain.argtypes = [vp, i32, C.c_float, i32]
        L.jpeg_hip_get_pixels.argtypes = [vp, vp, i64]
        L.jpeg_hip_histograms.argtypes = [vp, vp]
        L.jpeg_hip_histograms_host.argtypes = [vp, vp]
        L.jpeg_hip_huffman_from_histogram.argtypes = [vp, vp, vp]
        L.jpeg_hip_huffman_from_histogram_limit.argtypes = [vp, i32, vp, vp]
        L.jpeg_hip_standard_huffman.argtypes = [vp, vp]
        L.jpeg_hip_write_dht.argtypes = [vp, vp, i32, i32, vp, i32]
        L.jpeg_hip_write_headers.argtypes = [C.POINTER(Params), vp, vp, vp, i32]
        L.jpeg_hip_set_huffman.argtypes = [vp, vp, vp]
        L.jpeg_hip_build_huffman.argtypes = [vp, vp]
        L.jpeg_hip_get_huffman.argtypes = [vp, vp, vp]
        L.jpeg_hip_emit_scan.argtypes = [vp, vp, i64, C.POINTER(i64)]
        L.jpeg_hip_emit_scan_host.argtypes = [vp, vp, i64, C.POINTER(i64)]
        L.jpeg_hip_get_coefficients.argtypes = [vp, i32, vp, i64]
        L.jpeg_hip_coefficients_device.argtypes = [vp]
        L.jpeg_hip_coefficients_device.restype = vp
        L.jpeg_hip_analyze_band.argtypes = [vp, vp, i32, i64, C.POINTER(Params), vp]
        L.jpeg_hip_finish_band.argtypes = [vp, vp, vp, i64, C.POINTER(i64)]
        L.jpeg_hip_finish_band_start.argtypes = [vp, vp, vp, i64]
        L.jpeg_hip_finish_band_wait.argtypes = [vp, C.POINTER(i64)]
        L.jpeg_hip_encode_sharded.argtypes = [vp, vp, vp, i32, i64, C.POINTER(Params), vp, i64, C.POINTER(i64)]
        L.jpeg_hip_allreduce_histograms.argtypes = [vp, vp, vp]
        L.jpeg_hip_comm_unique_id.argtypes = [vp]
        L.jpeg_hip_comm_create.argtypes = [vp, i32, i32, i32, C.POINTER(vp)]
        L.jpeg_hip_comm_destroy.argtypes = [vp]
        L.jpeg_hip_comm_destroy.restype = None
        L.jpeg_hip_synthetic_images.argtypes = [vp, i32, C.c_int32, C.c_int32, C.c_int32, C.c_uint32, vp, i64, i64, i32]
        L.jpeg_hip_set_profiling.argtypes = [vp, i32]
        L.jpeg_hip_get_timings.argtypes = [vp, vp]
        L.jpeg_hip_get_host_timeline.argtypes = [vp, vp]
        L.jpeg_hip_get_profile_sums.argtypes = [vp, vp, vp, C.POINTER(i64), C.POINTER(i64)]
        _lib = L
    return _lib


def status_string(status):
    return lib().jpeg_hip_status_string(int(status)).decode()


def _np(a):
    return a.ctypes.data_as(C.c_void_p)


def make_params(width, height, sampling, qtables, restart_interval=0, pad_ones=False, mcu_rows=None,
                standard_huffman=False):
    p = Params()
    p.width, p.height = int(width), int(height)
    s = np.asarray(sampling, np.int32).reshape(12)
    for i in range(12):
        p.sampling[i] = int(s[i])
    q = np.asarray(qtables, np.float32).reshape(3, 64)
    for c in range(3):
        for i in range(64):
            p.qtable[c][i] = float(q[c, i])
    p.restart_interval = int(restart_interval)
    p.pad_ones = 1 if pad_ones else 0
    p.mcu_row_begin, p.mcu_row_end = (0, 0) if mcu_rows is None else (int(mcu_rows[0]), int(mcu_rows[1]))
    p.standard_huffman = 1 if standard_huffman else 0
    return p


def standard_huffman():
    """ITU-T T.81 Annex K.3 tables as (lengths uint8[4,256], codes uint16[4,256])."""
    lengths = np.zeros((4, 256), np.uint8)
    codes = np.zeros((4, 256), np.uint16)
    st = lib().jpeg_hip_standard_huffman(_np(lengths), _np(codes))
    if st != OK:
        raise JpegHipError(st, "standard tables")
    return lengths, codes


def decode_info(jpeg):
    """Frame parameters of a baseline JPEG (bytes) without decoding it."""
    buf = np.frombuffer(jpeg, np.uint8)
    info = ImageInfo()
    st = lib().jpeg_hip_decode_info(_np(buf), buf.size, C.byref(info))
    if st != OK:
        raise JpegHipError(st, "unsupported or damaged JPEG headers")
    return info


def compute_layout(params):
    lay = Layout()
    st = lib().jpeg_hip_compute_layout(C.byref(params), C.byref(lay))
    if st != OK:
        raise JpegHipError(st, "invalid parameters")
    return lay


def huffman_from_histogram(counts):
    """Pure host: (lengths u8[256], codes u16[256]) for a 256-bin histogram."""
    cnt = np.ascontiguousarray(counts, np.uint64).reshape(256)
    lengths = np.zeros(256, np.uint8)
    codes = np.zeros(256, np.uint16)
    st = lib().jpeg_hip_huffman_from_histogram(_np(cnt), _np(lengths), _np(codes))
    if st != OK:
        raise JpegHipError(st, "huffman_from_histogram")
    return lengths, codes


def huffman_from_histogram_limit(counts, code_word_length):
    """Pure host: (lengths u8[256], codes u64[256]) with no code longer than code_word_length (9..64)."""
    cnt = np.ascontiguousarray(counts, np.uint64).reshape(256)
    lengths = np.zeros(256, np.uint8)
    codes = np.zeros(256, np.uint64)
    st = lib().jpeg_hip_huffman_from_histogram_limit(_np(cnt), int(code_word_length), _np(lengths), _np(codes))
    if st != OK:
        raise JpegHipError(st, "huffman_from_histogram_limit")
    return lengths, codes


def write_dht(lengths, codes, table_id, coefficient_type):
    lengths = np.ascontiguousarray(lengths, np.uint8).reshape(256)
    codes = np.ascontiguousarray(codes, np.uint16).reshape(256)
    out = np.zeros(512, np.uint8)
    n = lib().jpeg_hip_write_dht(_np(lengths), _np(codes), int(table_id), int(coefficient_type), _np(out), out.size)
    if n < 0:
        raise JpegHipError(n, "write_dht")
    return out[:n].tobytes()


def write_headers(params, lengths, codes):
    lengths = np.ascontiguousarray(lengths, np.uint8).reshape(4, 256)
    codes = np.ascontiguousarray(codes, np.uint16).reshape(4, 256)
    out = np.zeros(4096, np.uint8)
    n = lib().jpeg_hip_write_headers(C.byref(params), _np(lengths), _np(codes), _np(out), out.size)
    if n < 0:
        raise JpegHipError(n, "write_headers")
    return out[:n].tobytes()


def comm_unique_id():
    """128 bytes that every rank of a job passes to Communicator (made by one rank, handed to the others)."""
    buf = np.zeros(128, np.uint8)
    st = lib().jpeg_hip_comm_unique_id(_np(buf))
    if st != OK:
        raise JpegHipError(st, "ncclGetUniqueId failed")
    return buf


class Communicator:
    """An RCCL communicator created through the library (ncclCommInitRank on `device`)."""

    def __init__(self, unique_id, rank, world_size, device):
        uid = np.ascontiguousarray(unique_id, np.uint8).reshape(128)
        self._h = C.c_void_p()
        st = lib().jpeg_hip_comm_create(_np(uid), int(rank), int(world_size), int(device), C.byref(self._h))
        if st != OK:
            self._h = C.c_void_p()
            raise JpegHipError(st, "ncclCommInitRank failed (rank %d of %d on GPU %d)" % (rank, world_size, device))
        self.rank, self.world_size, self.device = rank, world_size, device

    @property
    def handle(self):
        return self._h

    def close(self):
        if self._h:
            lib().jpeg_hip_comm_destroy(self._h)
            self._h = C.c_void_p()


class Context:
    """One encoder context on one GPU (wraps jpeg_hip_ctx)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        st = lib().jpeg_hip_create(int(device), C.byref(self._h))
        if st != OK:
            self._h = C.c_void_p()
            raise JpegHipError(st, "cannot open GPU %d" % device)
        self.device = device

    def close(self):
        if self._h:
            lib().jpeg_hip_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, st):
        if st != OK:
            raise JpegHipError(st, lib().jpeg_hip_last_error(self._h).decode())

    @property
    def stream(self):
        return lib().jpeg_hip_stream(self._h)

    # ---- ordering against the caller's stream ----
    def wait_for_stream(self, hip_stream):
        """Everything the context launches from now on waits for what is queued on hip_stream (an integer handle, 0 = default stream)."""
        self._check(lib().jpeg_hip_wait_for_stream(self._h, C.c_void_p(hip_stream)))

    def stream_wait_for_ctx(self, hip_stream):
        """hip_stream waits for what the context has launched so far."""
        self._check(lib().jpeg_hip_stream_wait_for_ctx(self._h, C.c_void_p(hip_stream)))

    def _torch_stream(self):
        # Only when the process uses torch on a GPU: device pointers handed to this class are then
        # tensors torch fills, uploads and reads on its current stream, while the library works
        # on the context's own.
        torch = sys.modules.get("torch")
        if torch is None or not torch.cuda.is_initialized():
            return None
        return torch.cuda.current_stream(self.device)

    def _after_torch(self):
        """Called before handing a device pointer to the library: orders the context's stream
        behind torch's pending work (a fill or an upload that is still running would otherwise
        race with the library's kernels).  An idle stream needs nothing."""
        s = self._torch_stream()
        if s is not None and not s.query():
            self.wait_for_stream(s.cuda_stream)

    def release_to_torch(self):
        """For the caller to use after an entry point that returns before its kernels have
        finished (transform, histograms_device, analyze_band, allreduce_histograms): torch's
        current stream waits for them, so tensors they write can be used by torch right away.
        Not done automatically: an event in the context's stream costs the host and the GPU some
        ten microseconds, which a caller that goes on with finish_band has no use for."""
        s = self._torch_stream()
        if s is not None:
            self.stream_wait_for_ctx(s.cuda_stream)

    # ---- one shot ----
    def encode(self, d_rgb, fmt, stride, params, d_out, capacity):
        self._after_torch()
        n = C.c_int64(0)
        self._check(lib().jpeg_hip_encode(self._h, C.c_void_p(d_rgb), fmt, stride, C.byref(params), C.c_void_p(d_out),
                                          capacity, C.byref(n)))
        return n.value

    def encode_static(self, d_rgb, fmt, stride, params, d_out, capacity):
        """jpeg_hip_encode on the library's fixed workspace: allocates no device memory."""
        self._after_torch()
        n = C.c_int64(0)
        self._check(lib().jpeg_hip_encode_static(self._h, C.c_void_p(d_rgb), fmt, stride, C.byref(params),
                                                 C.c_void_p(d_out), capacity, C.byref(n)))
        return n.value

    def encode_batch(self, d_rgb, fmt, stride, image_stride, count, params, d_out, slot_bytes):
        """`count` images of one shape, all resident in HBM -> int64 array of file lengths."""
        self._after_torch()
        lens = np.zeros(count, np.int64)
        self._check(lib().jpeg_hip_encode_batch(self._h, C.c_void_p(d_rgb), fmt, stride, image_stride, count,
                                                C.byref(params), C.c_void_p(d_out), slot_bytes, _np(lens)))
        return lens

    def encode_batch_host(self, frames, params, slot_bytes):
        """frames: uint8/float32 [N, H, W, 3] on the host -> list of JFIF byte strings."""
        frames = np.ascontiguousarray(frames)
        fmt = RGBF32 if frames.dtype == np.float32 else RGB8
        n = frames.shape[0]
        out = np.zeros(n * slot_bytes, np.uint8)
        lens = np.zeros(n, np.int64)
        self._check(lib().jpeg_hip_encode_batch_host(self._h, _np(frames), fmt, frames.strides[1], frames.strides[0], n,
                                                     C.byref(params), _np(out), slot_bytes, _np(lens)))
        return [out[k * slot_bytes:k * slot_bytes + int(lens[k])].tobytes() for k in range(n)]

    def encode_host(self, rgb, params, stride=None):
        """rgb: numpy uint8/float32 [H, W, 3] on the host -> JFIF bytes."""
        rgb = np.ascontiguousarray(rgb)
        fmt = RGBF32 if rgb.dtype == np.float32 else RGB8
        if stride is None:
            stride = rgb.strides[0]
        lay = compute_layout(params)
        out = np.empty(int(lay.max_jpeg_bytes), np.uint8)   # worst case: large, so not cleared
        n = C.c_int64(0)
        self._check(lib().jpeg_hip_encode_host(self._h, _np(rgb), fmt, stride, C.byref(params), _np(out), out.size,
                                               C.byref(n)))
        return out[:n.value].tobytes()

    # ---- staged ----
    def begin(self, params):
        self._check(lib().jpeg_hip_begin(self._h, C.byref(params)))

    def upload_pixels(self, rgb, stride=None):
        rgb = np.ascontiguousarray(rgb)
        fmt = RGBF32 if rgb.dtype == np.float32 else RGB8
        if stride is None:
            stride = rgb.strides[0]
        self._check(lib().jpeg_hip_upload_pixels(self._h, _np(rgb), fmt, stride))

    def upload_planes(self, r, g, b, stride=None):
        """Three float32 planes (rows ``stride`` bytes apart) instead of interleaved pixels."""
        planes = [np.asarray(p) for p in (r, g, b)]
        for p in planes:
            if p.dtype != np.float32 or p.ndim != 2 or p.strides[1] != 4:
                raise ValueError("planes are 2-D float32 arrays with packed rows")
        if stride is None:
            stride = planes[0].strides[0]
        if any(p.strides[0] != stride for p in planes):
            raise ValueError("the planes must share one row stride")
        self._check(lib().jpeg_hip_upload_planes(self._h, *[C.c_void_p(p.ctypes.data) for p in planes], stride))

    def bind_pixels(self, d_rgb, fmt, stride):
        self._after_torch()
        self._check(lib().jpeg_hip_bind_pixels(self._h, C.c_void_p(d_rgb), fmt, stride))

    def transform(self, stages=STAGE_ALL):
        self._check(lib().jpeg_hip_transform(self._h, stages))

    def histograms_device(self, d_hist):
        self._after_torch()
        self._check(lib().jpeg_hip_histograms(self._h, C.c_void_p(d_hist)))

    def histograms(self):
        h = np.zeros((4, 256), np.uint64)
        self._check(lib().jpeg_hip_histograms_host(self._h, _np(h)))
        return h

    def build_huffman(self, hist):
        h = np.ascontiguousarray(hist, np.uint64).reshape(4, 256)
        self._check(lib().jpeg_hip_build_huffman(self._h, _np(h)))

    def set_huffman(self, lengths, codes):
        lengths = np.ascontiguousarray(lengths, np.uint8).reshape(4, 256)
        codes = np.ascontiguousarray(codes, np.uint16).reshape(4, 256)
        self._check(lib().jpeg_hip_set_huffman(self._h, _np(lengths), _np(codes)))

    def get_huffman(self):
        lengths = np.zeros((4, 256), np.uint8)
        codes = np.zeros((4, 256), np.uint16)
        self._check(lib().jpeg_hip_get_huffman(self._h, _np(lengths), _np(codes)))
        return lengths, codes

    def emit_scan(self, d_out, capacity):
        self._after_torch()
        n = C.c_int64(0)
        self._check(lib().jpeg_hip_emit_scan(self._h, C.c_void_p(d_out), capacity, C.byref(n)))
        return n.value

    def emit_scan_host(self):
        n = C.c_int64(0)
        self._check(lib().jpeg_hip_emit_scan_host(self._h, None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), np.uint8)
        self._check(lib().jpeg_hip_emit_scan_host(self._h, _np(out), out.size, C.byref(n)))
        return out[:n.value].tobytes()

    def coefficients(self, channel, layout):
        n = int(layout.blocks[channel]) * 64
        out = np.zeros(n, np.int16)
        self._check(lib().jpeg_hip_get_coefficients(self._h, channel, _np(out), n))
        return out.reshape(-1, 64)

    # ---- bands ----
    def analyze_band(self, d_rgb, fmt, stride, params, d_hist):
        self._after_torch()
        self._check(lib().jpeg_hip_analyze_band(self._h, C.c_void_p(d_rgb), fmt, stride, C.byref(params),
                                                C.c_void_p(d_hist)))

    def encode_sharded(self, comm, d_rgb, fmt, stride, params, d_out, capacity):
        """This rank's piece of a sharded image: analyze, all-reduce of the histograms over `comm`, finish."""
        self._after_torch()
        n = C.c_int64(0)
        self._check(lib().jpeg_hip_encode_sharded(self._h, comm.handle, C.c_void_p(d_rgb), fmt, stride, C.byref(params),
                                                  C.c_void_p(d_out), capacity, C.byref(n)))
        return n.value

    def allreduce_histograms(self, comm, d_hist):
        self._after_torch()
        self._check(lib().jpeg_hip_allreduce_histograms(self._h, comm.handle, C.c_void_p(d_hist)))

    def allreduce_status(self, comm, local_status):
        """Number of ranks of `comm` that report a failure (local_status != OK); a collective."""
        n = C.c_int(0)
        self._check(lib().jpeg_hip_allreduce_status(self._h, comm.handle, int(local_status), C.byref(n)))
        return n.value

    def finish_band(self, d_hist_sum, d_out, capacity):
        self._after_torch()
        n = C.c_int64(0)
        # d_hist_sum None: the band is the whole image, use the histograms as analyze_band counted them
        self._check(lib().jpeg_hip_finish_band(self._h, C.c_void_p(d_hist_sum) if d_hist_sum else None, C.c_void_p(d_out), capacity,
                                               C.byref(n)))
        return n.value

    def finish_band_start(self, d_hist_sum, d_out, capacity):
        self._after_torch()
        self._check(lib().jpeg_hip_finish_band_start(self._h, C.c_void_p(d_hist_sum) if d_hist_sum else None,
                                                     C.c_void_p(d_out), capacity))

    def finish_band_wait(self):
        n = C.c_int64(0)
        self._check(lib().jpeg_hip_finish_band_wait(self._h, C.byref(n)))
        return n.value

    def set_profiling(self, enabled):
        self._check(lib().jpeg_hip_set_profiling(self._h, 1 if enabled else 0))

    # ---- pixel filters ----
    def filter_pixels(self, which, channel=0, value=1.0):
        self._check(lib().jpeg_hip_filter_pixels(self._h, which, channel, value))

    def channel_gain(self, channel, value, after_subsampling=False):
        """Encoder::multiplyColorChannelBy after convertToYCbCr: component 0 Y, 1 Cb, 2 Cr."""
        self._check(lib().jpeg_hip_channel_gain(self._h, channel, value, 1 if after_subsampling else 0))

    def get_pixels(self, width, rows):
        out = np.zeros((rows, width, 3), np.float32)
        self._check(lib().jpeg_hip_get_pixels(self._h, _np(out), out.size))
        return out

    # ---- verification decoder ----
    def decode_host(self, jpeg):
        """bytes -> (uint8 [H, W, 3], ImageInfo)."""
        buf = np.frombuffer(jpeg, np.uint8)
        info = decode_info(jpeg)
        out = np.zeros((info.height, info.width, 3), np.uint8)
        self._check(lib().jpeg_hip_decode_host(self._h, _np(buf), buf.size, _np(out), info.width * 3, out.size,
                                               C.byref(info)))
        return out, info

    def decode(self, jpeg, d_rgb, stride, capacity):
        """bytes -> RGB8 at device pointer d_rgb; returns ImageInfo."""
        self._after_torch()
        buf = np.frombuffer(jpeg, np.uint8)
        info = ImageInfo()
        self._check(lib().jpeg_hip_decode(self._h, _np(buf), buf.size, C.c_void_p(d_rgb), stride, capacity,
                                          C.byref(info)))
        return info

    def decoded_coefficients(self, info):
        out = np.zeros((int(info.total_blocks), 64), np.int16)
        self._check(lib().jpeg_hip_decoded_coefficients(self._h, _np(out), out.size))
        return out

    def psnr(self, d_a, stride_a, d_b, stride_b, width, height):
        """(mse, psnr in dB) between two RGB8 images in device memory."""
        self._after_torch()
        mse, psnr = C.c_double(0), C.c_double(0)
        self._check(lib().jpeg_hip_psnr(self._h, C.c_void_p(d_a), stride_a, C.c_void_p(d_b), stride_b, width, height,
                                        C.byref(mse), C.byref(psnr)))
        return mse.value, psnr.value

    def synthetic_images(self, kind, width, y0, y1, seed, d_out, stride, image_stride=0, count=1):
        """Rows y0..y1 of `count` synthetic RGB8 images (seeds seed, seed + 1, ...) into device memory."""
        self._after_torch()
        self._check(lib().jpeg_hip_synthetic_images(self._h, {"noise": 0, "natural": 1}[kind], width, y0, y1,
                                                    seed & 0xFFFFFFFF, C.c_void_p(d_out), stride, image_stride, count))

    def host_timeline(self):
        """Host clock (microseconds) at the eight stations of the last analyze / finish pair."""
        t = np.zeros(8, np.float64)
        self._check(lib().jpeg_hip_get_host_timeline(self._h, _np(t)))
        return t

    def profile_sums(self):
        """(stage_ms[4], host_us[8], images, entropy stage launches) added up since set_profiling(True)."""
        ms, us, n, runs = np.zeros(4, np.float64), np.zeros(8, np.float64), C.c_int64(0), C.c_int64(0)
        self._check(lib().jpeg_hip_get_profile_sums(self._h, _np(ms), _np(us), C.byref(n), C.byref(runs)))
        return ms, us, n.value, runs.value

    def timings(self):
        t = np.zeros(4, np.float32)
        self._check(lib().jpeg_hip_get_timings(self._h, _np(t)))
        return t
