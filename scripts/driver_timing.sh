set -e
python3 - <<PY
import sys, os
sys.path.insert(0, "jpeg-encoder_amd/python")
from jpegenc_amd import synth
import numpy as np
for (w,h,name) in [(1920,1080,"a"),(3840,2160,"b")]:
    img = synth.generate("natural", w, h, seed=1)
    with open("/tmp/%s.ppm"%name,"wb") as f:
        f.write(b"P6\n%d %d\n255\n"%(w,h)); f.write(img.tobytes())
    from PIL import Image
    Image.fromarray(img).save("/tmp/%s.png"%name)
PY
for f in a.ppm a.png b.ppm b.png; do
  for i in 1 2; do t0=$(date +%s.%N); jpeg-encoder_amd/lib/encode_jpeg /tmp/$f /tmp/out.jpg 420 90 timing; t1=$(date +%s.%N); echo "$f $(python3 -c "print(round(($t1-$t0)*1e3,1))") ms wall (process start to exit)"; done
done
