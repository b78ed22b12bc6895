"""ctypes front end of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module; the shipped encoder never does.  The shared object is built by
``make -C oracle`` (or ``__graft_entry__.build()``) into ``oracle/_build``.
"""
import ctypes as C
import os
import subprocess
import time

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libjpeg_oracle.so")
_SO_AVX2 = os.path.join(_HERE, "_build", "libjpeg_oracle_avx2.so")
_lib = None
_lib_avx2 = None


def build(force=False):
    srcs = [os.path.join(_HERE, n) for n in ("jpeg_oracle.cpp", "jpeg_oracle_avx2.cpp")]
    stale = any(not os.path.exists(so) or any(os.path.exists(c) and os.path.getmtime(c) > os.path.getmtime(so) for c in srcs)
                for so in (_SO, _SO_AVX2))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def lib_avx2():
    """The AVX2 + OpenMP restatement (jpeg_oracle_avx2.cpp): the scalar library's interface plus
    orcv_encode.  The CPU baseline of bench.py; checked against the scalar one by the tests."""
    global _lib_avx2
    if _lib_avx2 is None:
        if not os.path.exists(_SO_AVX2):
            build()
        L = _bind(C.CDLL(_SO_AVX2))
        L.orcv_encode.restype = C.c_void_p
        L.orcv_encode.argtypes = L.orc_encode.argtypes
        L.orcv_threads.restype = C.c_int
        _lib_avx2 = L
    return _lib_avx2


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = _bind(C.CDLL(_SO))
    return _lib


def _bind(L):
    L.orc_encode.restype = C.c_void_p
    L.orc_encode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_int,
                             C.c_int, C.c_void_p, C.c_int, C.c_int]
    L.orc_free.argtypes = [C.c_void_p]
    L.orc_set_fixed_tables.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_set_fixed_tables.restype = None
    L.orc_set_channel_gains.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_set_channel_gains.restype = None
    for name in ("orc_jpeg_size", "orc_scan_offset", "orc_scan_len"):
        getattr(L, name).restype = C.c_int64
        getattr(L, name).argtypes = [C.c_void_p]
    L.orc_jpeg_copy.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_coef_count.restype = C.c_int64
    L.orc_coef_count.argtypes = [C.c_void_p, C.c_int]
    L.orc_coef_copy.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    L.orc_hist_copy.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_table_copy.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.orc_dims.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_constants.argtypes = [C.c_void_p]
    L.orc_color_convert.argtypes = [C.c_float, C.c_float, C.c_float, C.c_void_p]
    L.orc_dct_block.argtypes = [C.c_void_p, C.c_void_p]
    L.orc_dct_quant_block.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_huffman.restype = C.c_int
    L.orc_huffman.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.orc_bitbuffer_replay.restype = C.c_int64
    L.orc_bitbuffer_replay.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64,
                                       C.c_void_p]
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def constants():
    out = np.zeros(12, np.float32)
    lib().orc_constants(_p(out))
    return out


def color_convert(r, g, b):
    out = np.zeros(3, np.float32)
    lib().orc_color_convert(np.float32(r), np.float32(g), np.float32(b), _p(out))
    return out


def dct_block(block):
    a = np.ascontiguousarray(block, np.float32).reshape(64)
    out = np.zeros(64, np.float32)
    lib().orc_dct_block(_p(a), _p(out))
    return out.reshape(8, 8)


def dct_quant_block(block01, qtable):
    a = np.ascontiguousarray(block01, np.float32).reshape(64)
    q = np.ascontiguousarray(qtable, np.float32).reshape(64)
    out = np.zeros(64, np.float32)
    lib().orc_dct_quant_block(_p(a), _p(q), _p(out))
    return out.reshape(8, 8)


def huffman(counts, ht_num=0, ht_type=0):
    """-> (lengths[256] u8, codes[256] u16, dht segment bytes)."""
    cnt = np.ascontiguousarray(counts, np.uint64).reshape(256)
    lengths = np.zeros(256, np.uint8)
    codes = np.zeros(256, np.uint16)
    dht = np.zeros(2 + 2 + 1 + 16 + 256, np.uint8)
    n = lib().orc_huffman(_p(cnt), ht_num, ht_type, _p(lengths), _p(codes), _p(dht), dht.size)
    assert n >= 0
    return lengths, codes, dht[:n].tobytes()


def huffman_limit(counts, code_word_length):
    """-> (lengths[256] u8, codes[256] u64), or None where the reference returns a null table."""
    cnt = np.ascontiguousarray(counts, np.uint64).reshape(256)
    lengths = np.zeros(256, np.uint8)
    codes = np.zeros(256, np.uint64)
    L = lib()
    L.orc_huffman_limit.restype = C.c_int
    L.orc_huffman_limit.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    if L.orc_huffman_limit(_p(cnt), int(code_word_length), _p(lengths), _p(codes)) != 0:
        return None
    return lengths, codes


def bitbuffer_replay(pushes, escape=True, fill=True):
    """pushes: list of (nbits, value) with value right-aligned; -> (bytes, bit length)."""
    n = len(pushes)
    nbits = np.zeros(max(n, 1), np.int32)
    src = np.zeros((max(n, 1), 4), np.uint8)
    for i, (nb, v) in enumerate(pushes):
        nbits[i] = nb
        if nb:
            w = (int(v) & ((1 << nb) - 1)) << (32 - nb)
            src[i] = [(w >> 24) & 255, (w >> 16) & 255, (w >> 8) & 255, w & 255]
    cap = 16 + 4 * n * 2
    out = np.zeros(cap, np.uint8)
    bitlen = C.c_int64(0)
    ln = lib().orc_bitbuffer_replay(n, _p(nbits), _p(src), int(escape), int(fill), _p(out), cap, C.byref(bitlen))
    assert ln >= 0
    return out[:ln].tobytes(), bitlen.value


class Encoded:
    """Everything the oracle produced for one image."""

    def __init__(self, jpeg, coef, hist, lengths, codes, scan_offset, scan_len, dims):
        self.jpeg = jpeg
        self.coef = coef            # list of 3 int16 arrays [blocks, 64] (MCU order, zig-zag)
        self.hist = hist            # uint64 [4, 256]: DC-Y, DC-C, AC-Y, AC-C
        self.lengths = lengths      # uint8 [4, 256]
        self.codes = codes          # uint16 [4, 256]
        self.scan_offset = scan_offset
        self.scan_len = scan_len
        self.sim_w, self.sim_h = int(dims[0]), int(dims[1])
        self.channel_sizes = [(int(dims[2 + 2 * c]), int(dims[3 + 2 * c])) for c in range(3)]

    @property
    def scan(self):
        return self.jpeg[self.scan_offset:self.scan_offset + self.scan_len]


def set_fixed_tables(lengths=None, codes=None):
    """Use these Huffman tables (uint8/uint16 [4, 256]: DC-Y, DC-C, AC-Y, AC-C) for the following
    encodes instead of optimised ones; call without arguments to switch back."""
    for L in (lib(), lib_avx2()):         # each library holds its own copy of the setting
        if lengths is None:
            L.orc_set_fixed_tables(None, None)
            continue
        ln = np.ascontiguousarray(lengths, np.uint8).reshape(4, 256)
        cd = np.ascontiguousarray(codes, np.uint16).reshape(4, 256)
        L.orc_set_fixed_tables(_p(ln), _p(cd))


def set_channel_gains(ops=()):
    """Encoder::multiplyColorChannelBy calls after convertToYCbCr, for the following encodes:
    (channel, value, after_subsampling) triples in call order; empty switches them off."""
    ops = list(ops)
    ch = np.array([o[0] for o in ops], np.int32)
    st = np.array([1 if o[2] else 0 for o in ops], np.int32)
    va = np.array([o[1] for o in ops], np.float32)
    for L in (lib(), lib_avx2()):
        L.orc_set_channel_gains(len(ops), _p(ch), _p(st), _p(va))


def encode(rgb, scheme, qtables, restart_interval=0, pad_ones=False, stride=None, hist_override=None, rst_base=0,
           header_height=0, variant="scalar", timing=None):
    """rgb: uint8 or float32 array [H, W, 3]; scheme: 12 ints; qtables: float32 [3, 64].

    hist_override / rst_base / header_height: band of a sharded image (tables from summed
    histograms, global RSTn numbering, whole-image height in the frame header).
    variant "avx2": the vectorised restatement.  timing: a list that receives the seconds the
    encode itself took (without the copies into numpy arrays that follow it)."""
    L = lib() if variant == "scalar" else lib_avx2()
    fn = L.orc_encode if variant == "scalar" else L.orcv_encode
    rgb = np.asarray(rgb)
    h, w = rgb.shape[0], rgb.shape[1]
    if rgb.dtype == np.uint8:
        st = 0
    elif rgb.dtype == np.float32:
        st = 1
    else:
        raise TypeError("rgb must be uint8 or float32")
    rgb = np.ascontiguousarray(rgb)
    if stride is None:
        stride = w * 3
    sch = np.ascontiguousarray(scheme, np.int32).reshape(12)
    qt = np.ascontiguousarray(qtables, np.float32).reshape(3, 64)
    ho = None
    if hist_override is not None:
        ho = np.ascontiguousarray(hist_override, np.uint64).reshape(4, 256)
    t0 = time.perf_counter()
    hnd = fn(_p(rgb), st, w, h, stride, _p(sch), _p(qt), int(restart_interval), int(pad_ones),
             _p(ho) if ho is not None else None, int(rst_base), int(header_height))
    if timing is not None:
        timing.append(time.perf_counter() - t0)
    try:
        n = L.orc_jpeg_size(hnd)
        jpeg = np.zeros(n, np.uint8)
        L.orc_jpeg_copy(hnd, _p(jpeg))
        coef = []
        for c in range(3):
            cnt = L.orc_coef_count(hnd, c)
            a = np.zeros(cnt, np.int16)
            L.orc_coef_copy(hnd, c, _p(a))
            coef.append(a.reshape(-1, 64))
        hist = np.zeros((4, 256), np.uint64)
        L.orc_hist_copy(hnd, _p(hist))
        lengths = np.zeros((4, 256), np.uint8)
        codes = np.zeros((4, 256), np.uint16)
        for t in range(4):
            L.orc_table_copy(hnd, t, _p(lengths[t]), _p(codes[t]))
        dims = np.zeros(8, np.int32)
        L.orc_dims(hnd, _p(dims))
        return Encoded(jpeg.tobytes(), coef, hist, lengths, codes, L.orc_scan_offset(hnd), L.orc_scan_len(hnd), dims)
    finally:
        L.orc_free(hnd)
