export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
cp distributed-matvec_amd/libls_amd.so /tmp/gcp4.so
for v in 4 2 8 4; do
  if [ $v = 4 ]; then cp /tmp/gcp4.so distributed-matvec_amd/libls_amd.so; else cp scripts/tmp_libs/gcp$v.so distributed-matvec_amd/libls_amd.so; fi
  echo "== kGCPull=$v"; timeout 300 python scripts/tile_bench.py --L 36 --symm --steps 5 2>&1 | grep "L="
done
