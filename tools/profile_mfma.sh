# MFMA counters of the pair-block kernel (tools/kbench_mfma.py) in their own --pmc pass -> gpurun_out/${R}_rocprofv3_pmc_mfma.txt
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
R=${1:-r02}
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --output-format csv -d gpurun_out/prof_${R}_mfma -- python3 tools/kbench_mfma.py 4000 300000 > gpurun_out/${R}_kbench_mfma_under_rocprof.jsonl 2> gpurun_out/${R}_prof_mfma.err
python3 tools/rocprof_summary.py gpurun_out/prof_${R}_mfma "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU -- python3 tools/kbench_mfma.py 4000 300000" > gpurun_out/${R}_rocprofv3_pmc_mfma.txt
rm -rf gpurun_out/prof_${R}_mfma
grep -i 'mfma\|k_pair' gpurun_out/${R}_rocprofv3_pmc_mfma.txt | head -30
tail -3 gpurun_out/${R}_prof_mfma.err
