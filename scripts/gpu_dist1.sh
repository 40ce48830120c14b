# GPU job: the N > 1 code paths driven with a single rank over RCCL (API-level validation)
set -x
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_matvec.py -m gpu -q -k "replicated" 2>&1 | tail -4
python bench.py --force-distributed --exchange replicated --model heisenberg_chain_28 --steps 5 --warmup 2 --no-cpu-baseline
LS_AMD_FORCE_TILE=1 python bench.py --force-distributed --exchange packets --model heisenberg_chain_28 --steps 5 --warmup 2 --no-extra --no-cpu-baseline
python bench.py --force-distributed --model heisenberg_chain_32_symm --steps 5 --warmup 2 --no-extra --no-cpu-baseline
