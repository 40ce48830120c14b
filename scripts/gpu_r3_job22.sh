# GPU job 22: k_chain_t with counted far gathers (no zero fill) against the previous build; VALU / wave counters of the new one
export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r3; mkdir -p $OUT
cd $ROOT
P=distributed-matvec_amd
timeout 900 python -m pytest tests/test_gpu_matvec.py tests/test_gpu_parity_configs.py -m gpu -q -x -k "chain or staged or edge or sibling or c128" > $OUT/pytest_job22.log 2>&1; grep -E "passed|failed|error" $OUT/pytest_job22.log | tail -2
cp $P/libls_amd.so /tmp/new.so
for v in new prev new prev; do
  [ $v = prev ] && cp $P/libls_amd_prev.so $P/libls_amd.so || cp /tmp/new.so $P/libls_amd.so
  timeout 600 python bench.py --no-cpu-baseline --no-extra --steps 10 > $OUT/var3_f64_$v.json 2>/dev/null
  timeout 600 python bench.py --dtype c128 --no-cpu-baseline --no-extra --steps 8 > $OUT/var3_c128_$v.json 2>/dev/null
  echo "$v: f64 $(grep -o '"kernel_ms_avg": [0-9.]*' $OUT/var3_f64_$v.json | head -1) c128 $(grep -o '"kernel_ms_avg": [0-9.]*' $OUT/var3_c128_$v.json | head -1)"
done | tee $OUT/chain_variants3.txt
cp /tmp/new.so $P/libls_amd.so
cd /tmp
timeout -k 5 150 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/sq22 -o pmc -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/sq22.log 2>&1
python3 $ROOT/scripts/rocpd_summary.py $OUT/sq22 > $OUT/sq22_summary.txt 2>&1; rm -rf $OUT/sq22
grep -E "k_chain_t" $OUT/sq22_summary.txt | grep -E "SQ_|GRBM" | cut -c1-30,60-140
