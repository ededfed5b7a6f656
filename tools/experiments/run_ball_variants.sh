# same-device A/B of library builds on tools/ab_ball.py.  usage: bash tools/experiments/run_ball_variants.sh base TAG [TAG ...]
for t in "$@"; do
  if [ $t = base ]; then unset RFOPS_LIB; else export RFOPS_LIB=$PWD/rfnet_amd/variants/librfops_$t.so; fi
  echo "== $t"; timeout 120 python tools/ab_ball.py 2>&1 | grep -v amdgpu.ids
done
