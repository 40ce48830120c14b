export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for i in $(seq 1 30); do
  timeout 600 python -X faulthandler -m pytest tests/test_gpu_loopback.py -m gpu -q -x > /tmp/lb_$i.log 2>&1; rc=$?
  echo "run $i rc=$rc $(tail -1 /tmp/lb_$i.log | cut -c1-80)"
  if [ $rc -ne 0 ]; then grep -v "^$" /tmp/lb_$i.log | grep -A40 -E "Fatal|Segmentation|Aborted|Error" | head -70; break; fi
done
