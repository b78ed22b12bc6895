"""CPU restatement of the inverse operations the verification decoder runs on the GPU.

TEST INFRASTRUCTURE ONLY (like everything under oracle/): numpy float32, every product and
sum rounded separately and taken in the reference's order.

  * inverse transform: DCT::directIDCT, DCT.cpp:46-79 (values[0][0] += 1024; out[x][y] =
    (2/N) * sum_i sum_j values[i][j]*ci*cj*cos(xFactor*i)*cos(yFactor*j));
  * colour: convertYCbCrToRGBAVXImpl, SIMD256.h:188-216 (Encoder::convertToRGB, Encoder.cpp:25-33);
  * bytes: clamp(0, int(v * 255 + 0.5f), 255), ImageLoader.cpp:168-172.

Parity unpinned: the reference's tests hold no vectors for these functions; the restatement
is additionally compared with an independent decoder (Pillow) in the tests.
"""
import numpy as np

F32 = np.float32
ZIGZAG = np.array([0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7,
                   14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46,
                   53, 60, 61, 54, 47, 55, 62, 63])


def cos_table():
    t = np.zeros((8, 8), F32)
    for x in range(8):
        for i in range(8):
            t[x, i] = F32(np.cos(np.pi * (2 * x + 1) * i / 16.0))
    return t


def inverse_blocks(coef_zz, q_natural):
    """coef_zz: int16 [n, 64] zig-zag order; q_natural: float32 [64] natural order -> float32 [n, 8, 8] samples (0..1)."""
    n = coef_zz.shape[0]
    F = np.zeros((n, 64), F32)
    qz = np.asarray(q_natural, F32).reshape(64)[ZIGZAG]
    F[:, ZIGZAG] = coef_zz.astype(F32) * qz[None, :]
    F[:, 0] += F32(1024.0)
    F = F.reshape(n, 8, 8)
    ct = cos_table()
    sqrt_half = F32(0.70710678118654752440)
    one = F32(1.0)
    out = np.zeros((n, 8, 8), F32)
    for x in range(8):
        for y in range(8):
            temp = np.zeros(n, F32)
            for i in range(8):
                ci = sqrt_half if i == 0 else one
                for j in range(8):
                    cj = sqrt_half if j == 0 else one
                    temp = temp + F[:, i, j] * ci * cj * ct[x, i] * ct[y, j]
            out[:, x, y] = (F32(0.25) * temp) / F32(255.0)
    return out


def planes_from_blocks(samples, mcus_per_row, mcu_rows, h, v):
    """samples: [n, 8, 8] of one channel, blocks in MCU order (row-major inside an MCU) -> plane."""
    plane = np.zeros((mcu_rows * v * 8, mcus_per_row * h * 8), F32)
    k = 0
    for my in range(mcu_rows):
        for mx in range(mcus_per_row):
            for by in range(v):
                for bx in range(h):
                    y0, x0 = (my * v + by) * 8, (mx * h + bx) * 8
                    plane[y0:y0 + 8, x0:x0 + 8] = samples[k]
                    k += 1
    return plane


def to_rgb8(planes, width, height, wf, hf):
    """planes: 3 float32 planes; wf/hf: pixels per sample across/down per channel -> uint8 [H, W, 3]."""
    ys, xs = np.arange(height)[:, None], np.arange(width)[None, :]
    v = [planes[c][ys // hf[c], xs // wf[c]] for c in range(3)]
    cb, cr = v[1] + F32(-0.5), v[2] + F32(-0.5)
    r = v[0] + F32(1.4021) * cr
    g = (v[0] + F32(-0.3442) * cb) + F32(-0.7142) * cr
    b = v[0] + F32(1.7720) * cb
    out = np.zeros((height, width, 3), np.uint8)
    for c, ch in enumerate((r, g, b)):
        t = (ch * F32(255.0) + F32(0.5)).astype(np.int32)   # truncation toward zero, like static_cast<int>
        out[:, :, c] = np.clip(t, 0, 255)
    return out


def decode_from_coefficients(coef, qtables, width, height, inv_w, inv_h):
    """coef: per channel int16 [n_c, 64]; qtables float32 [3, 64] natural; inv_w/inv_h: blocks per MCU across/down."""
    hmax, vmax = max(inv_w), max(inv_h)
    mcus_per_row = (width + hmax * 8 - 1) // (hmax * 8)
    mcu_rows = (height + vmax * 8 - 1) // (vmax * 8)
    planes = []
    for c in range(3):
        s = inverse_blocks(coef[c], qtables[c])
        planes.append(planes_from_blocks(s, mcus_per_row, mcu_rows, inv_w[c], inv_h[c]))
    wf = [hmax // inv_w[c] for c in range(3)]
    hf = [vmax // inv_h[c] for c in range(3)]
    return to_rgb8(planes, width, height, wf, hf)
