"""The synthetic images of ``synth`` generated on the GPU with torch integer ops.

Same bytes as the numpy versions (tests compare them); used to fill large benchmark
inputs without pushing hundreds of megabytes over PCIe.
"""
import torch

_M32 = 0xFFFFFFFF


def _mix32(x):
    x = x ^ (x >> 16)
    x = (x * 0x7FEB352D) & _M32
    x = x ^ (x >> 15)
    x = (x * 0x846CA68B) & _M32
    x = x ^ (x >> 16)
    return x


def _grid(width, y0, y1, device):
    ys = torch.arange(y0, y1, dtype=torch.int64, device=device)[:, None, None]
    xs = torch.arange(width, dtype=torch.int64, device=device)[None, :, None]
    cs = torch.arange(3, dtype=torch.int64, device=device)[None, None, :]
    return ys, xs, cs


def rows(kind, width, y0, y1, seed, device, full_width=None):
    """uint8 tensor [y1 - y0, width, 3] holding image rows y0..y1 of the named image."""
    fw = width if full_width is None else full_width
    ys, xs, cs = _grid(width, y0, y1, device)
    idx = ((ys * fw + xs) * 3 + cs) & _M32
    h = _mix32((seed & _M32) ^ idx)
    if kind == "noise":
        return (h >> 24).to(torch.uint8)
    t = ((xs * (3 + cs) + ys * (5 - cs)) & _M32) >> 3
    lo = t & 255
    tri = torch.where(((t >> 8) & 1) != 0, 255 - lo, lo)
    return torch.clamp(tri + (h >> 28) - 8, 0, 255).to(torch.uint8)


def generate(kind, width, height, seed=None, device="cuda", y0=0, y1=None, chunk_rows=512):
    if seed is None:
        seed = 42 if kind == "noise" else 1
    y1 = height if y1 is None else y1
    out = torch.empty((y1 - y0, width, 3), dtype=torch.uint8, device=device)
    for a in range(y0, y1, chunk_rows):
        b = min(y1, a + chunk_rows)
        out[a - y0:b - y0] = rows(kind, width, a, b, seed, device)
    return out
