# GPU job: FETCH_SIZE and TCC hit/miss of $CMD under the current environment; tag = $1.
# One counter group per pass (FETCH_SIZE with WRITE_SIZE in one pass exceeds the hardware and makes
# rocprofv3 abort and then hang), every pass under its own timeout.
export TMPDIR=/tmp
TAG=${1:-q}
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
ROOT=$GRAFT_REPO_ROOT
cd /tmp
timeout -k 5 150 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o pmc -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout -k 5 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $OUT/pmc_tcc -o pmc -- $CMD > $OUT/pmc_tcc.log 2>&1
python3 $ROOT/scripts/rocpd_summary.py $OUT > $OUT/summary.txt 2>&1
rm -rf $OUT/*/*.db
grep -E "FETCH|WRITE|TCC" $OUT/summary.txt | cut -c1-40,60-140
