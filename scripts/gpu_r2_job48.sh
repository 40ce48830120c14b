export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for v in 0 3000 0 6500 12000 3000; do
  echo "== LS_AMD_CHAIN_LDS_PAD=$v"; LS_AMD_CHAIN_LDS_PAD=$v timeout 300 python scripts/order_sweep.py --L 32 --steps 10 --configs ";" 2>&1 | grep staged | head -1 | cut -c1-110
done
