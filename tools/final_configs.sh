#!/bin/bash
# the BASELINE shapes on one MI355X (tools/bench_configs.py), one JSON line each -> gpurun_out/final_configs.jsonl
set -e
out=gpurun_out/final_configs.jsonl; : > $out
run() { timeout -k 10 400 python tools/bench_configs.py "$@" 2>gpurun_out/final_cfg.err | tail -1 >> $out; tail -1 $out | cut -c1-200; }
run c1
run c3
run c3 1000000 clustered
run c4
HX_ITER_QUERIES=6000 HX_C5_QUERIES=10000 run c5
HX_ITER_QUERIES=6000 HX_C5_QUERIES=10000 run c5 2000000 clustered
