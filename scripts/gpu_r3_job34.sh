# GPU job 34: HDF5 tests on the device (eigenvectors written block by block, three hash partitions)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3; mkdir -p $OUT
timeout 600 python -m pytest tests/test_hdf5_io.py -q -x > $OUT/pytest_job34.log 2>&1; tail -5 $OUT/pytest_job34.log
