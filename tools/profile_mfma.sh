# MFMA counters of the pair-block kernel (tools/kbench_mfma.py) in their own --pmc pass -> gpurun_out/${R}_rocprofv3_pmc_mfma.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
R=${1:-r03}
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d gpurun_out/prof_${R}_mfma -- python3 tools/kbench_mfma.py 4000 300000 > gpurun_out/${R}_kbench_mfma_under_rocprof.jsonl 2> gpurun_out/${R}_prof_mfma.err
python3 tools/rocprof_summary.py gpurun_out/prof_${R}_mfma "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU -- python3 tools/kbench_mfma.py 4000 300000" > gpurun_out/${R}_rocprofv3_pmc_mfma.txt
rm -rf gpurun_out/prof_${R}_mfma
grep -i 'mfma\|k_pair' gpurun_out/${R}_rocprofv3_pmc_mfma.txt | head -30
tail -3 gpurun_out/${R}_prof_mfma.err
# round 3: the build's own GEMM (k_wgemm_f16: select_neighbors of the device-resident build on the matrix cores) under the same counters
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d gpurun_out/prof_${R}_wgemm -- python3 tools/bench_configs.py c4 300000 > gpurun_out/${R}_c4_under_rocprof_mfma.json 2> gpurun_out/${R}_prof_wgemm.err
python3 tools/rocprof_summary.py gpurun_out/prof_${R}_wgemm "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU -- python3 tools/bench_configs.py c4 300000" > gpurun_out/${R}_rocprofv3_pmc_mfma_build_gemm.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_${R}_wgemm_f -- python3 tools/bench_configs.py c4 300000 > /dev/null 2>> gpurun_out/${R}_prof_wgemm.err
python3 tools/rocprof_summary.py gpurun_out/prof_${R}_wgemm_f "rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 tools/bench_configs.py c4 300000" > gpurun_out/${R}_rocprofv3_pmc_fetch_build_gemm.txt
rm -rf gpurun_out/prof_${R}_wgemm gpurun_out/prof_${R}_wgemm_f
grep -i 'k_wgemm\|k_wselect' gpurun_out/${R}_rocprofv3_pmc_mfma_build_gemm.txt gpurun_out/${R}_rocprofv3_pmc_fetch_build_gemm.txt | head -30
