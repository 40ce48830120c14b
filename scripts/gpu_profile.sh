# GPU job: rocprofv3 kernel trace + PMC passes of the default bench command (chain_32, f64)
set -x
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_${1:-r1}
mkdir -p $OUT
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-extra --no-cpu-baseline ${BENCH_ARGS:-}"
cd /tmp
timeout -k 5 240 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
timeout -k 5 240 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout -k 5 240 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o pmc -- $CMD > $OUT/pmc_write.log 2>&1
timeout -k 5 240 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD -d $OUT/pmc_sq -o pmc -- $CMD > $OUT/pmc_sq.log 2>&1
timeout -k 5 240 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $OUT/pmc_tcc -o pmc -- $CMD > $OUT/pmc_tcc.log 2>&1
cd $OUT
find . -name "*.csv" | head -30
for f in $(find . -name "*kernel_stats.csv"); do echo == $f; head -12 $f; done
# summarise the PMC csvs per kernel: mean counter value per dispatch
python3 - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(f)):
        k = (row.get('Kernel_Name','')[:60], row.get('Counter_Name'))
        acc[k][0] += float(row.get('Counter_Value', 0)); acc[k][1] += 1
    print('==', f)
    for (kn, cn), (s, n) in sorted(acc.items()):
        if 'k_direct' in kn or 'k_diag' in kn or 'k_tile' in kn or 'k_scatter' in kn:
            print(f'{kn:60s} {cn:22s} mean/dispatch {s/n:.6g}  n={n}')
PY
