# GPU job: the N > 1 code paths driven with a single rank over RCCL (API-level validation), plus headline benches
set -x
export TMPDIR=/tmp
LS_AMD_FORCE_TILE=1 python bench.py --force-distributed --exchange packets --model heisenberg_chain_28 --steps 5 --warmup 2 --no-extra --no-cpu-baseline
python bench.py --force-distributed --exchange replicated --model heisenberg_chain_28 --steps 5 --warmup 2 --no-extra --no-cpu-baseline
python bench.py --model heisenberg_chain_36_symm --steps 5 --warmup 2 --no-cpu-baseline
python bench.py --steps 10 --warmup 3
