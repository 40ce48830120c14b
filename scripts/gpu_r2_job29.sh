export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
cp distributed-matvec_amd/libls_amd.so /tmp/base.so
for v in 6 4 8 6; do
  if [ $v = 6 ]; then cp /tmp/base.so distributed-matvec_amd/libls_amd.so; else cp scripts/tmp_libs/farc$v.bin distributed-matvec_amd/libls_amd.so; fi
  echo "== kChainFarC=$v"; timeout 300 python scripts/order_sweep.py --L 32 --steps 6 --dtype c128 --configs ";" 2>&1 | grep staged | cut -c1-120
done
