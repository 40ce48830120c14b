export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in base new base new; do
  cp scripts/tmp_libs/$v.bin distributed-matvec_amd/libls_amd.so
  echo "== $v"; timeout 300 python scripts/order_sweep.py --L 32 --steps 8 --configs ";" 2>&1 | grep staged | cut -c1-110
done
