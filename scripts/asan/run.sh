# Runs the ASan driver and selftest over a few inputs and options (GPU box).
# scripts/asan/out is listed in .gpurunignore (17 MB that no other GPU run reads): take that line out before
# sending this script to the box.
set -e
cd "$(dirname "$0")/../.."
X=scripts/asan/out
python3 - <<PY
import sys
sys.path.insert(0, "jpeg-encoder_amd/python")
from jpegenc_amd import synth
from PIL import Image
import numpy as np
for (w, h, name) in [(1920, 1080, "a"), (333, 211, "b"), (17, 9, "c")]:
    img = synth.generate("natural" if name != "b" else "noise", w, h, seed=1)
    with open("/tmp/%s.ppm" % name, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (w, h)); f.write(img.tobytes())
    Image.fromarray(img).save("/tmp/%s.png" % name)
    Image.fromarray(img).convert("L").save("/tmp/%s_grey.png" % name)
    Image.fromarray(img).convert("P").save("/tmp/%s_pal.png" % name)
    Image.fromarray(np.dstack([img, img[..., :1]])).save("/tmp/%s_rgba.png" % name)
PY
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0
for f in a.ppm a.png b.png b_grey.png b_pal.png b_rgba.png c.ppm c.png; do
  for opt in "420 90" "444 0 restart=5" "422 75 sepia gain=1:0.5" "411 90 ycbcr-gain=1:0.5 reduced-gain=0:1.25" "420 90 standard-huffman"; do
    timeout -k 10 120 $X/encode_jpeg /tmp/$f /tmp/out.jpg $opt > /dev/null
  done
done
# the same with the blocks' halves in two planes (large 4:2:0 images take that layout by themselves)
for f in a.ppm b.png c.ppm; do
  JPEGENC_SPLIT_MIN_BLOCKS=0 timeout -k 10 120 $X/encode_jpeg /tmp/$f /tmp/out.jpg 420 90 > /dev/null
  JPEGENC_SPLIT_MIN_BLOCKS=0 JPEGENC_GROUP_MIN_BLOCKS=0 JPEGENC_HIST_TILES_MIN_BLOCKS=0 timeout -k 10 120 $X/encode_jpeg /tmp/$f /tmp/out.jpg 420 90 restart=8 > /dev/null
done
timeout -k 10 120 $X/host_selftest > /dev/null
timeout -k 10 300 $X/abi_exercise
echo "asan runs clean"
