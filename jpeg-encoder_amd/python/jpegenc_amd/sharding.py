"""Sharding one large image over the GPUs of a node by restart intervals.

Every rank encodes a contiguous band of MCU rows.  The bands are independent once the
DC predictor and the bit position restart at the band boundary, which the restart
markers (RSTn) provide; the only exchange is one all-reduce of the four 256-bin symbol
histograms so that every rank derives the same Huffman tables.  The reference has no
restart markers (JPEGSegments.h:26-34), so this path has no counterpart there.
"""
import numpy as np


def band_rows(mcu_rows, world_size, rank):
    """Contiguous, balanced split of ``mcu_rows`` MCU rows; returns (begin, end)."""
    if not 0 <= rank < world_size:
        raise ValueError("rank out of range")
    if world_size > mcu_rows:
        raise ValueError("more ranks than MCU rows")
    return mcu_rows * rank // world_size, mcu_rows * (rank + 1) // world_size


def row_restart_interval(mcus_per_row):
    """Restart interval of one MCU row, the coarsest interval every band split honours."""
    if mcus_per_row > 65535:
        raise ValueError("an MCU row does not fit one restart interval")
    return mcus_per_row


def all_reduce_histograms(hist, group=None):
    """Sums the [4, 256] int64 histogram tensor over the ranks in place (one collective)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(hist, op=dist.ReduceOp.SUM, group=group)
    return hist


def failed_ranks(local_ok, group=None, device=None):
    """One small collective after the pieces are written: how many ranks failed locally (an output
    buffer too small for the piece, say).  Every rank gets the same number, so the job either keeps
    all pieces or none."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([0 if local_ok else 1], dtype=torch.int64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return int(t.item())


def concatenate_pieces(pieces):
    """File bytes from the per-band pieces in band order."""
    return b"".join(pieces)


class HipBandEncoder:
    """One rank's half of a sharded encode on its GPU (C ABI: analyze_band / finish_band)."""

    def __init__(self, ctx, binding, width, height, sampling, qtables, world_size, rank, pad_ones=False):
        self.ctx, self.B = ctx, binding
        whole = binding.make_params(width, height, sampling, qtables)
        lay = binding.compute_layout(whole)
        self.mcu_height = lay.mcu_height
        self.rows = band_rows(lay.mcu_rows, world_size, rank)
        self.params = binding.make_params(width, height, sampling, qtables,
                                          restart_interval=row_restart_interval(lay.mcus_per_row),
                                          pad_ones=pad_ones, mcu_rows=self.rows)
        self.layout = binding.compute_layout(self.params)
        self.pixel_rows = (self.rows[0] * lay.mcu_height, min(height, self.rows[1] * lay.mcu_height))

    def analyze(self, d_rgb_band, fmt, stride, d_hist):
        self.ctx.analyze_band(d_rgb_band, fmt, stride, self.params, d_hist)

    def finish(self, d_hist_sum, d_out, capacity):
        return self.ctx.finish_band(d_hist_sum, d_out, capacity)


def as_uint64(hist_int64):
    return np.asarray(hist_int64).view(np.uint64)
