#!/bin/bash
# ONE parametrised runner for every `gpurun` job of the round (replaces the per-job scripts of rounds 3/4):
#     gpurun --timeout T -- 'bash scripts/gpu_job.sh <tag> <step> [<step> ...]'
# Steps write under gpurun_out/<tag>/ (merged back by gpurun); what is kept as evidence is copied into profiles/ by hand.
#   smoke            __graft_entry__.smoke()
#   tests            the whole -m gpu suite            tests:<expr> = pytest -k <expr>      tests3 = the suite as three concurrent processes
#   bench            default bench line                bench1 = --force-distributed --distributed-extras (one RCCL rank)
#   bench_c128       chain_32 c128 line                bench1p = one RCCL rank, packets only
#   packets[:v ..]   packet path A/B: sorted streams | pre-indexed + atomics | state-carrying + atomics, timing trees (scripts/tile_bench.py)
#   packets_prof     rocprofv3 kernel trace + SQ counters of chain_28 x 8 partitions
#   stream_cost      k_chain_t with one more 8-byte stream per row (profiling build): what a byte per row costs
#   lattice          heisenberg_square_6x6 / 4x4: K4 mode 5 vs mode 4 (LS_AMD_K4=cosets)      ablate:<model>  scripts/ablate_pull.py
#   loopback:<L>[s]  scripts/loopback_bench.py, 8 loop-back ranks (s = _symm)
#   pmc:<model>:<dtype>   kernel trace + FETCH/WRITE/VALU/TCC passes -> pmc_traffic entry (scripts/gpu_pmc_traffic.sh)
#   prof_bench       rocprofv3 --kernel-trace --stats of the default bench command (no extras)
#   ab:<ENV>=<a>,<b>:<cmd...>   run <cmd> once per value of ENV (A/B inside one job, one box)
export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p "$OUT"
for step in "$@"; do
  echo "=== [$TAG] $step ($(date +%T))"
  case "$step" in
    smoke) timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 ;;
    tests) ( time timeout 1500 python -m pytest tests -m gpu -q --maxfail=15 --timeout 420 --timeout-method signal --durations=40 > "$OUT/pytest_gpu.log" 2>&1 ) 2>&1 | grep real; grep -A45 "slowest" "$OUT/pytest_gpu.log" | cut -c1-150 | head -48; tail -6 "$OUT/pytest_gpu.log" ;;
    tests3) # the -m gpu suite as three concurrent pytest processes (partition by file; the ports of the multi-process tests are pid-derived): for a short slot
      ( timeout 150 python -m pytest tests/test_gpu_parity_configs.py -m gpu -q --timeout 140 --timeout-method signal > "$OUT/pytest_A.log" 2>&1; echo "A rc=$? $(tail -1 "$OUT/pytest_A.log")" ) &
      ( timeout 150 python -m pytest tests/test_gpu_matvec.py tests/test_gpu_loopback.py -m gpu -q --timeout 140 --timeout-method signal > "$OUT/pytest_B.log" 2>&1; echo "B rc=$? $(tail -1 "$OUT/pytest_B.log")" ) &
      ( timeout 150 python -m pytest tests -m gpu -q --timeout 140 --timeout-method signal --ignore=tests/test_gpu_parity_configs.py --ignore=tests/test_gpu_matvec.py --ignore=tests/test_gpu_loopback.py > "$OUT/pytest_C.log" 2>&1; echo "C rc=$? $(tail -1 "$OUT/pytest_C.log")" ) &
      wait
      grep -h "FAILED\|ERROR" "$OUT"/pytest_[ABC].log | head -20 ;;
    hang:*) # a test suspected of hanging: per-test timeout with a dump of every thread's stack (faulthandler), then exit
      timeout 400 python -X faulthandler -m pytest tests -m gpu -q -x --timeout 150 --timeout-method thread -k "${step#hang:}" > "$OUT/pytest_hang.log" 2>&1; tail -120 "$OUT/pytest_hang.log" | cut -c1-220 ;;
    tests:*) ( time timeout 1200 python -m pytest tests -m gpu -q --maxfail=15 --timeout 420 --timeout-method signal -k "${step#tests:}" >> "$OUT/pytest_focus.log" 2>&1 ) 2>&1 | grep real; tail -15 "$OUT/pytest_focus.log" ;;
    bench) ( time timeout 900 python bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err" ) 2>&1 | grep real; tail -c 6000 "$OUT/bench_default.json"; tail -5 "$OUT/bench_default.err" ;;
    bench_c128) timeout 300 python bench.py --dtype c128 --steps 10 --warmup 3 --no-cpu-baseline --no-extra > "$OUT/bench_c128.json" 2>/dev/null; cut -c1-400 "$OUT/bench_c128.json" ;;
    bench1) ( time timeout 900 python bench.py --force-distributed --distributed-extras --no-cpu-baseline --kDisplayTimings > "$OUT/bench_one_rank.json" 2> "$OUT/bench_one_rank.err" ) 2>&1 | grep real; echo "rc=$?"; tail -c 5000 "$OUT/bench_one_rank.json"; grep -v "^$" "$OUT/bench_one_rank.err" | tail -40 ;;
    packets|packets:*) # packet path A/B on one device, P logical partitions: sorted streams (default) | pre-indexed + atomics | state-carrying + atomics
      variants="${step#packets}"; variants="${variants#:}"; [ -z "$variants" ] && variants="streams indexed states"
      for v in $variants; do
        case $v in streams) ENVV="LS_AMD_PACKET_STREAMS=1";; indexed) ENVV="LS_AMD_PACKET_INDEX=1 LS_AMD_PACKET_STREAMS=0";; states) ENVV="LS_AMD_PACKET_INDEX=0";; *) ENVV="$v";; esac
        for args in "--L 28 --P 8" "--L 28 --P 8 --dtype c128" "--L 28 --P 2" "--L 30 --P 8"; do
          echo -n "$v $args: "; env $ENVV timeout 300 python scripts/tile_bench.py $args --steps 5 --tree 2>&1 | grep -E "matvec=|producers|consumers" | tr '\n' ' ' | sed 's/  */ /g' | cut -c1-420; echo
        done
      done | tee -a "$OUT/packets_ab.txt" ;;
    packets_prof)
      WITH_MEM=1 CMD="python $GRAFT_REPO_ROOT/scripts/tile_bench.py --L 28 --P 8 --steps 3" bash scripts/gpu_profile_cmd.sh "${TAG}_packets" > "$OUT/packets_prof.log" 2>&1
      grep -E "k_tile|k_scatter|k_window|k_diag" "gpurun_out/prof_${TAG}_packets/summary.txt" | cut -c1-200 | head -30 ;;
    bench1p) # one RCCL rank through the C host's packet path only (chain_32: every packet is an own-partition packet)
      ( time timeout 600 python bench.py --force-distributed --exchange packets --no-cpu-baseline --no-extra --steps 5 --warmup 2 --kDisplayTimings > "$OUT/bench_one_rank_packets.json" 2> "$OUT/bench_one_rank_packets.err" ) 2>&1 | grep real; echo "rc=$?"; tail -c 3000 "$OUT/bench_one_rank_packets.json"; grep -v "^$" "$OUT/bench_one_rank_packets.err" | tail -25 ;;
    lattice) # the reference's benchmark model: K4 mode 5 (factorised point group) against mode 4 (one network per coset), same box
      for k4 in default cosets; do
        for m in heisenberg_square_6x6 heisenberg_square_4x4; do
          echo -n "K4=$k4 $m: "; if [ $k4 = default ]; then timeout 300 python scripts/lattice_bench.py $m 5 2>&1 | tail -1 | cut -c1-600; else LS_AMD_K4=$k4 timeout 300 python scripts/lattice_bench.py $m 5 2>&1 | tail -1 | cut -c1-600; fi
        done
      done | tee "$OUT/lattice_k4_ab.txt" ;;
    stream_cost) timeout 400 python scripts/chain_stream_cost.py 32 10 2>&1 | grep -v amdgpu.ids | tee "$OUT/chain_stream_cost.txt" ;;
    ablate:*) # where the time of the projected pull kernel goes (profiling build, results wrong by construction)
      timeout 600 python scripts/ablate_pull.py "${step#ablate:}" 2>&1 | tee "$OUT/ablate_${step#ablate:}.txt" | tail -12 | cut -c1-300 ;;
    loopbackp:*) a=${step#loopbackp:} # packets only, sorted streams | atomics
      for st in 1 0; do echo "--- LS_AMD_PACKET_STREAMS=$st"; LS_AMD_PACKET_STREAMS=$st timeout 600 python scripts/loopback_bench.py --L "$a" --P 8 --steps 3 --mode packets 2>&1 | grep -E "ranks sharing|aggregate|rank 0:|producers|localeIdxOf|consumers" | cut -c1-300; done | tee "$OUT/loopback_packets_$a.txt" ;;
    loopback:*) a=${step#loopback:}; L=${a%s}; S=""; [ "$a" != "$L" ] && S="--symm"
      timeout 900 python scripts/loopback_bench.py --L "$L" $S --P 8 --steps 3 > "$OUT/loopback_$a.txt" 2>&1; grep -E "ranks sharing|x received|aggregate" "$OUT/loopback_$a.txt" | cut -c1-300 ;;
    pmc:*) IFS=: read -r _ model dtype mode <<< "$step"   # pmc:<model>:<dtype>[:push]
      MODEL=$model DTYPE=$dtype MODE=${mode:-auto} TAG="${TAG}_pmc_${model}_${dtype}${mode:+_$mode}" bash scripts/gpu_pmc_traffic.sh 2>&1 | tail -14 ;;
    prof_bench)
      cd /tmp && timeout -k 5 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/$OUT/trace" -o trace -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-extra > "$GRAFT_REPO_ROOT/$OUT/prof_bench.log" 2>&1
      cd "$GRAFT_REPO_ROOT" && python3 scripts/rocpd_summary.py "$OUT" > "$OUT/prof_bench_summary.txt" 2>&1; rm -rf "$OUT"/trace/*.db "$OUT"/trace/*/*.db; head -12 "$OUT/prof_bench_summary.txt" | cut -c1-170 ;;
    ab:*) spec=${step#ab:}; var=${spec%%=*}; rest=${spec#*=}; vals=${rest%%:*}; cmd=${rest#*:}
      for v in ${vals//,/ }; do echo "--- $var=$v"; env "$var=$v" timeout 600 bash -c "$cmd" 2>&1 | tail -6 | cut -c1-400; done | tee -a "$OUT/ab_$var.txt" ;;
    *) echo "unknown step $step" ;;
  esac
done
