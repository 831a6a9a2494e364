#!/bin/bash
OUT=$PWD/gpurun_out/r06_o; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "reference_contract or permissive or holidays_through or jobs_as_pipelines or job or udf or frame or panel" > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -n 3 $OUT/pytest_gpu.log
python - <<'PY'
import sys, time, os
sys.path.insert(0, os.getcwd())
import bench, json
b = bench.boundary_legs()
print(json.dumps({k: {kk: v[kk] for kk in v if kk != 'workload'} for k, v in b.items()}, indent=1))
PY
