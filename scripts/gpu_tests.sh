# GPU job: smoke + the -m gpu parity suite (run through gpurun from the repo root)
set -x
export TMPDIR=/tmp
python __graft_entry__.py smoke 2>&1 | tail -5
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 2>&1 | tail -40
