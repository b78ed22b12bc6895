# The dense bit writer on a picture of few tiles per workgroup (4096 x 4096, 3072 tiles): library variants and both ways of numbering tiles.
export JPEGENC_WRITER=dense
for from in 0 100000; do
  echo "per XCD from $from tiles"
  JPEGENC_DENSE_PER_XCD_FROM=$from VARIANTS="$VARIANTS" bash scripts/ab/dense.sh 4k2_noise_q75_444 || exit 1
done
