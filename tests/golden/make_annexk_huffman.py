"""Writes annexk_huffman.json: the typical Huffman tables of ITU-T T.81 Annex K.3 as libjpeg
(through Pillow) puts them into a file saved without table optimisation.

    python tests/golden/make_annexk_huffman.py
"""
import io
import json
import os

import numpy as np
from PIL import Image


def main():
    rng = np.random.RandomState(0)
    im = Image.fromarray(rng.randint(0, 256, (64, 64, 3)).astype(np.uint8))
    buf = io.BytesIO()
    im.save(buf, "JPEG", quality=90, optimize=False)
    d = buf.getvalue()
    tables = {}
    i = 2
    while i < len(d):
        assert d[i] == 0xFF
        marker, length = d[i + 1], d[i + 2] * 256 + d[i + 3]
        if marker == 0xC4:
            seg, j = d[i + 4:i + 2 + length], 0
            while j < len(seg):
                tc_th = seg[j]
                bits = list(seg[j + 1:j + 17])
                n = sum(bits)
                tables["%d%d" % (tc_th >> 4, tc_th & 15)] = {"bits": bits, "values": list(seg[j + 17:j + 17 + n])}
                j += 17 + n
        if marker == 0xDA:
            break
        i += 2 + length
    out = {"source": "DHT segments of a Pillow/libjpeg file saved with optimize=False",
           "order": ["DC luminance (K.3)", "DC chrominance (K.4)", "AC luminance (K.5)", "AC chrominance (K.6)"],
           "tables": [tables["00"], tables["01"], tables["10"], tables["11"]]}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "annexk_huffman.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
