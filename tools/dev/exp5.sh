export TMPDIR=/tmp
mkdir -p gpurun_out/exp5
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for leg in "-" "TSF_QUAD_PREBUILD=0"; do
  echo "-- leg $leg"
  ( if [ "$leg" != "-" ]; then export $leg; fi; python tools/bench_ragged.py 2>/dev/null | head -1 | cut -c1-330 )
done ) 2>&1 | tee gpurun_out/exp5/summary.txt
