#!/bin/bash
# round 5, fourth GPU call: which interaction of the stream-ordered allocator corrupts a running kernel's scratch
# (tools/probes/mallocasync_probe.hip modes), and the FETCH_SIZE calibration
TAG=${1:-r05_d}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
/opt/rocm/bin/hipconfig --version > $OUT/rocm_version.txt 2>&1; cat /opt/rocm/.info/version >> $OUT/rocm_version.txt 2>/dev/null
for m in 0 0 2 4 8 6 14 1; do
  timeout 300 tools/probes/bin/mallocasync_probe $m > $OUT/probe_mode${m}_$RANDOM.txt 2>&1; rc=$?
  f=$(ls -t $OUT/probe_mode${m}_*.txt | head -1)
  echo "mode $m rc=$rc: $(tail -1 $f) | calls with corruption: $(grep -c -v 'rounds 0, inside the rounds 0' $f | head -1)" | tee -a $OUT/summary.txt
done
bash tools/dev/fetch_calib.sh $TAG/fetch_calib
cat $OUT/fetch_calib/summary.txt >> $OUT/summary.txt
echo done
