export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_matvec.py tests/test_gpu_parity_configs.py -m gpu -q -x -k "staged or chain_32 or row_kernel or block_rows" 2>&1 | tail -3
C=""
for m in 0 12 16 20 24 28 31; do C="$C;LS_AMD_CHAIN_MAXLO=$m"; done
timeout 600 python scripts/order_sweep.py --L 32 --steps 8 --configs ";${C#;};LS_AMD_BLOCKS_PER_CU=0" 2>&1 | grep -v amdgpu.ids | cut -c1-220
