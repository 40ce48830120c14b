# GPU job: try two ranks sharing the one GPU over RCCL (validates the real N > 1 collectives if RCCL permits it)
set -x
export TMPDIR=/tmp
export LS_AMD_BENCH_SHARE_DEVICE=1
export NCCL_DEBUG=WARN
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --model heisenberg_chain_24 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/two_ranks.log 2>&1; grep -vE "^W0|^\*\*\*" gpurun_out/two_ranks.log | grep -E "Error|error|NCCL|Duplicate|Traceback|File|^\{" | head -30
