# GPU job: parity suite + per-rank compute of the replicated-x mode with block rows
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --maxfail=5 2>&1 | tail -3
timeout 200 python scripts/repl_block_bench.py
LS_AMD_CHAIN=0 timeout 200 python scripts/repl_block_bench.py
