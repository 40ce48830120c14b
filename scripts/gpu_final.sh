# GPU job: smoke, the -m gpu parity suite, the default bench line, and the rocprofv3 passes of the headline kernel
export TMPDIR=/tmp
TAG=${1:-r1final2}
cd $GRAFT_REPO_ROOT
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3
timeout 900 python -m pytest tests -m gpu -q --maxfail=10 2>&1 | tail -8
timeout 600 python bench.py > gpurun_out/bench_default_$TAG.json 2> gpurun_out/bench_default_$TAG.err; tail -c 3000 gpurun_out/bench_default_$TAG.json
timeout 200 python bench.py --dtype c128 --steps 10 --warmup 3 --no-cpu-baseline --no-extra > gpurun_out/bench_c128_$TAG.json 2>/dev/null
export WITH_MEM=1 CMD="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra"
bash scripts/gpu_profile_cmd.sh $TAG > gpurun_out/prof_$TAG.log 2>&1
grep -E "k_chain|k_direct" gpurun_out/prof_$TAG/summary.txt | cut -c1-30,60-140 | head -40
