# rocprofv3 passes of the iterative-scan kernel on the C5 shape (run from the repo root on the GPU box); summaries go to gpurun_out/
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
R=${1:-r03}
# round 3: the iterative-scan kernel (k_fused MODE 2) on the C5 shape at 2M rows: kernel trace, SQ counters, FETCH_SIZE
export HX_C5_QUERIES=6000 HX_ITER_QUERIES=6000
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${R}_c5_kt -- python3 tools/bench_configs.py c5 2000000 clustered > gpurun_out/${R}_c5_under_rocprof_kt.json 2> gpurun_out/${R}_prof_c5.err
python3 tools/rocprof_summary.py gpurun_out/prof_${R}_c5_kt "rocprofv3 --kernel-trace --stats -- python3 tools/bench_configs.py c5 2000000 clustered (HX_C5_QUERIES=6000 HX_ITER_QUERIES=6000)" > gpurun_out/${R}_rocprofv3_kernel_trace_c5.txt
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES --output-format csv -d gpurun_out/prof_${R}_c5_sq -- python3 tools/bench_configs.py c5 2000000 clustered > /dev/null 2>> gpurun_out/${R}_prof_c5.err
python3 tools/rocprof_summary.py gpurun_out/prof_${R}_c5_sq "rocprofv3 --kernel-trace --pmc SQ_* -- python3 tools/bench_configs.py c5 2000000 clustered" > gpurun_out/${R}_rocprofv3_pmc_sq_c5.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_${R}_c5_fetch -- python3 tools/bench_configs.py c5 2000000 clustered > /dev/null 2>> gpurun_out/${R}_prof_c5.err
python3 tools/rocprof_summary.py gpurun_out/prof_${R}_c5_fetch "rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 tools/bench_configs.py c5 2000000 clustered" > gpurun_out/${R}_rocprofv3_pmc_fetch_c5.txt
rm -rf gpurun_out/prof_${R}_c5_kt gpurun_out/prof_${R}_c5_sq gpurun_out/prof_${R}_c5_fetch
grep 'k_fused<OpHamming, 2' gpurun_out/${R}_rocprofv3_kernel_trace_c5.txt gpurun_out/${R}_rocprofv3_pmc_sq_c5.txt gpurun_out/${R}_rocprofv3_pmc_fetch_c5.txt | head -20
