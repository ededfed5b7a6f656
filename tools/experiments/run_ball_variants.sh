for t in base qb4 qb16; do
  if [ $t = base ]; then unset RFOPS_LIB; else export RFOPS_LIB=$PWD/rfnet_amd/variants/librfops_$t.so; fi
  echo "== $t"; timeout 120 python tools/ab_ball.py 2>&1 | grep -v amdgpu.ids
done
