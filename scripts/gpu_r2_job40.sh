export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in base new base new; do
  cp scripts/tmp_libs/$v.bin distributed-matvec_amd/libls_amd.so
  echo "== $v"
  timeout 300 python scripts/tile_bench.py --L 36 --symm --steps 5 2>&1 | grep "L=" | cut -c60-200
  timeout 300 python scripts/tile_bench.py --L 28 --P 8 --steps 5 2>&1 | grep "L=" | cut -c60-200
done
