#!/usr/bin/env python3
"""Known-answer vectors that follow from literals in the reference's sources alone.

Nothing here was produced by running the reference: every byte is assembled from constants
that stand in its source files (field initialisers of its segment structs, its
quantization tables, its zig-zag order, the arguments its own ad-hoc test passes), with the
file:line of each literal noted beside the entry.  Writes the "literals" entry of
reference_kats.json.  The values below are data (numbers), not reference code.
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))

ZIGZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
          35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47,
          55, 62, 63]                                             # ZigZag.h:10-19
LUMINANCE = [16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29,
             51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92, 49, 64, 78, 87, 103, 121, 120,
             101, 72, 92, 95, 98, 112, 100, 103, 99]              # QuantizationTables.h:40-49
CHROMINANCE = [17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99,
               99, 99, 99, 99] + [99] * 32                         # QuantizationTables.h:51-60


def be16(v):
    return [(v >> 8) & 255, v & 255]


def sof0(width, height, factors):
    """StartOfFrame0, JPEGSegments.h:77-100: marker, length 0x11, accuracy 8, yResolution, xResolution,
    3 components: {id, (H << 4) | V, id - 1}."""
    out = [0xFF, 0xC0] + be16(0x11) + [0x08] + be16(height) + be16(width) + [0x03]
    for i, (h, v) in enumerate(factors):
        out += [i + 1, (h << 4) | (v & 15), i]
    return bytes(out)


def dqt(channel, table):
    """DefineQuantizationTable, JPEGSegments.h:152-167: marker, length 2 + 1 + 64, info = channel, zig-zag values."""
    return bytes([0xFF, 0xDB] + be16(67) + [channel] + [table[z] for z in ZIGZAG])


def main():
    lit = {}
    lit["_about"] = ("Vectors assembled from literals in the reference's sources only (no run of the reference): "
                     "each entry names the file:line of its literals.")
    # bitBufferTest, JPEG-Encoder-SIMD.cpp:73-75: StartOfFrame0(1680, 900, Sampling::Scheme422); Scheme422 =
    # Y {1,1}, Cb/Cr {2,1} (SamplingScheme.h:53-56), so stepSize = 16 x 8 and inverseFactor = (2,1), (1,1), (1,1)
    # (SamplingScheme.h:32-43).
    lit["sof0_1680x900_422_hex"] = {"hex": sof0(1680, 900, [(2, 1), (1, 1), (1, 1)]).hex(),
                                    "from": "JPEG-Encoder-SIMD.cpp:75; JPEGSegments.h:77-100; SamplingScheme.h:32-43,53-56"}
    lit["sof0_factor_bytes"] = {"444": [0x11, 0x11, 0x11], "422": [0x21, 0x11, 0x11], "411": [0x41, 0x11, 0x11],
                                "420": [0x22, 0x11, 0x11],
                                "from": "SamplingScheme.h:48-66 through JPEGSegments.h:92-97"}
    lit["soi_hex"] = {"hex": "ffd8", "from": "JPEGSegments.h:26-27,41-54"}
    lit["eoi_hex"] = {"hex": "ffd9", "from": "JPEGSegments.h:28,56-60"}
    lit["app0_hex"] = {"hex": bytes([0xFF, 0xE0, 0x00, 0x10, 0x4A, 0x46, 0x49, 0x46, 0x00, 0x01, 0x01, 0x00, 0x00, 0x48, 0x00,
                                     0x48, 0x00, 0x00]).hex(), "from": "JPEGSegments.h:62-75"}
    # bitBufferTest, JPEG-Encoder-SIMD.cpp:85-86: DQT of Y with JPEGQuantization::luminance, of Cb with ::chrominance
    lit["dqt_luminance_id0_hex"] = {"hex": dqt(0, LUMINANCE).hex(),
                                    "from": "JPEG-Encoder-SIMD.cpp:85; JPEGSegments.h:152-167; QuantizationTables.h:40-49; ZigZag.h:10-19"}
    lit["dqt_chrominance_id1_hex"] = {"hex": dqt(1, CHROMINANCE).hex(),
                                      "from": "JPEG-Encoder-SIMD.cpp:86; JPEGSegments.h:152-167; QuantizationTables.h:51-60; ZigZag.h:10-19"}
    # Encoder::serialize writes a third DQT for Cr (id 2) with the set's third table, Encoder.cpp:459-461
    lit["dqt_chrominance_id2_hex"] = {"hex": dqt(2, CHROMINANCE).hex(), "from": "Encoder.cpp:461; QuantizationTables.h:17-22"}
    # DefineHuffmannTable, JPEGSegments.h:102-140, 203-207: marker, length 2 + 1 + 16 + n, info = (id & 15) | (type << 4),
    # 16 counts, n symbols.  bitBufferTest passes id 3, AC, 7 distinct symbols (JPEG-Encoder-SIMD.cpp:77,83).
    lit["dht_layout"] = {"marker_hex": "ffc4", "length_for_7_symbols": 2 + 1 + 16 + 7, "info_id3_ac": (3 & 15) | (1 << 4),
                         "encoder_infos": [0x00, 0x01, 0x10, 0x11],
                         "from": "JPEGSegments.h:21-24,102-112,203-207; JPEG-Encoder-SIMD.cpp:83; Encoder.cpp:455-458"}
    # StartOfScan, JPEGSegments.h:175-190 with acHT = dcHT = {Y, Cb, Cb} = {0, 1, 1} (Encoder.cpp:463-468)
    lit["sos_hex"] = {"hex": bytes([0xFF, 0xDA, 0x00, 0x0C, 0x03, 0x01, 0x00, 0x02, 0x11, 0x03, 0x11, 0x00, 0x3F, 0x00]).hex(),
                      "from": "JPEGSegments.h:175-190; Encoder.cpp:463-470"}
    # Segment order of a file, Encoder.cpp:473-488
    lit["segment_order"] = {"markers": ["d8", "e0", "c0", "c4", "c4", "c4", "c4", "db", "db", "db", "da"],
                            "from": "Encoder.cpp:473-483"}
    # Encoder::lookupBitCategory, Encoder.cpp:354-366: 0 for 0, else 32 - clz(|value|): category k holds the values
    # whose magnitude lies in 2^(k-1) .. 2^k - 1.
    lit["bit_category_ranges"] = {"ranges": [[0, 0, 0]] + [[k, 1 << (k - 1), (1 << k) - 1] for k in range(1, 12)],
                                  "from": "Encoder.cpp:354-366"}
    # Amplitude bits, Encoder.cpp:276-285 and 318-327: the value itself for v >= 0, ~(-v) for v < 0, in `category` bits.
    lit["amplitude_examples"] = {"cases": [[5, 3, "101"], [-5, 3, "010"], [1, 1, "1"], [-1, 1, "0"], [-1023, 10, "0000000000"],
                                           [255, 8, "11111111"], [-256, 9, "011111111"]],
                                 "from": "Encoder.cpp:276-285,318-327"}
    path = os.path.join(HERE, "reference_kats.json")
    kats = json.load(open(path))
    kats["literals"] = lit
    json.dump(kats, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
