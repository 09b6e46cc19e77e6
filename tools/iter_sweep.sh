#!/bin/bash
# tuning sweep of the device-resident iterative scan (C5 shape): size of the visited table kept across resumes
for sh in 0 -1 -2 -3; do
  echo "== iter vis shift $sh"
  HX_ITER_VIS_SHIFT=$sh HX_ITER_QUERIES=6000 HX_C5_QUERIES=6000 timeout -k 10 300 python tools/bench_configs.py c5 1000000 clustered 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['iterative_relaxed']['qps'], d['iterative_relaxed']['recall_at_10'], d['fused']['redone'])" || exit 1
done
