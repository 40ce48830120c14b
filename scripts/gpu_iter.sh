# GPU job: smoke + parity suite + default bench line
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 200 python __graft_entry__.py smoke 2>&1 | tail -2
timeout 900 python -m pytest tests -m gpu -q --maxfail=5 2>&1 | tail -3
timeout 300 python bench.py --no-extra > gpurun_out/bench_default_last.json 2>/dev/null; cut -c1-330 gpurun_out/bench_default_last.json
