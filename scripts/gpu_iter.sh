# GPU job: parity suite + three repeated default bench runs (run-to-run spread)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --maxfail=5 2>&1 | tail -3
B="timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra"
for i in 1 2 3; do echo "+ default run $i"; $B; done
