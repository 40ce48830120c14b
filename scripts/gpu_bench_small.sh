# GPU job: quick benches (push and pull, f64) on small chains, then the headline workload
set -x
export TMPDIR=/tmp
for m in heisenberg_chain_24 heisenberg_chain_28; do
  python bench.py --model $m --steps 10 --warmup 3 --mode push --no-cpu-baseline
done
python bench.py --steps 5 --warmup 2 --mode push --cpu-sample 26
