# GPU job 4 (round 4): k_pairs_t -- parity on lattices, then square 6x5 / J1-J2 30 / 32 against the generic row kernel
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r4job4; mkdir -p $OUT
( time timeout 900 python -m pytest tests/test_gpu_matvec.py -m gpu -q -x -k "pairs or row_kernel_variants or kagome or square or staged" > $OUT/pytest_c.log 2>&1 ) 2>&1 | grep real; tail -15 $OUT/pytest_c.log
timeout 900 python scripts/nonchain_bench.py --models square_6x5,j1j2_30,j1j2_32 --steps 6 2>&1 | grep model | tee $OUT/nonchain.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['model'], d['mode'], d['kernel'], round(d['kernel_ms'],3),'ms', round(d['gnnz_per_s'],1),'Gnnz/s', 'GB/s', round(d['algorithmic_GBps'],1), 'diff', d['max_rel_diff_vs_first'])"
