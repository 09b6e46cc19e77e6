# rocprofv3 passes of the default bench command (run from the repo root on the GPU box); summaries go to gpurun_out/
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
R=${1:-r03}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${R}_kt -- python3 bench.py --no-cpu --no-k1-1536 --no-query-sweep --steps 5 > gpurun_out/${R}_bench_under_rocprof_kt.json 2> gpurun_out/${R}_prof_kt.err
python3 tools/rocprof_summary.py gpurun_out/prof_${R}_kt "rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu --no-k1-1536 --no-query-sweep --steps 5" > gpurun_out/${R}_rocprofv3_kernel_trace_1m.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_${R}_fetch -- python3 bench.py --no-cpu --no-k1-1536 --no-query-sweep --steps 2 > gpurun_out/${R}_bench_under_rocprof_fetch.json 2> gpurun_out/${R}_prof_fetch.err
python3 tools/rocprof_summary.py gpurun_out/prof_${R}_fetch "rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 bench.py --no-cpu --no-k1-1536 --no-query-sweep --steps 2" > gpurun_out/${R}_rocprofv3_pmc_fetch_size_1m.txt
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES --output-format csv -d gpurun_out/prof_${R}_sq -- python3 bench.py --no-cpu --no-k1-1536 --no-query-sweep --steps 2 > gpurun_out/${R}_bench_under_rocprof_sq.json 2> gpurun_out/${R}_prof_sq.err
python3 tools/rocprof_summary.py gpurun_out/prof_${R}_sq "rocprofv3 --kernel-trace --pmc SQ_* -- python3 bench.py --no-cpu --no-k1-1536 --no-query-sweep --steps 2" > gpurun_out/${R}_rocprofv3_pmc_sq_1m.txt
head -2 $(ls gpurun_out/prof_${R}_kt/*/*kernel_trace.csv | head -1) > gpurun_out/${R}_kernel_trace_csv_header.txt 2>/dev/null
rm -rf gpurun_out/prof_${R}_kt gpurun_out/prof_${R}_fetch gpurun_out/prof_${R}_sq
head -40 gpurun_out/${R}_rocprofv3_kernel_trace_1m.txt
