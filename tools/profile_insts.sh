# dynamic instruction counts of the bench's kernels (a separate --pmc pass: SQ_INSTS_*), for the heap (HX_SORTED_ARRAY=0) and sorted-array query kernels
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
R=${1:-r03}
for sa in 0; do     # (1: the sorted-array kernel, only in -DHX_EXPERIMENTS builds)
export HX_SORTED_ARRAY=$sa
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d gpurun_out/prof_${R}_insts_$sa -- python3 bench.py --no-cpu --no-k1-1536 --no-query-sweep --steps 2 > gpurun_out/${R}_bench_under_rocprof_insts_$sa.json 2> gpurun_out/${R}_prof_insts_$sa.err
python3 tools/rocprof_summary.py gpurun_out/prof_${R}_insts_$sa "HX_SORTED_ARRAY=$sa rocprofv3 --kernel-trace --pmc SQ_INSTS_* -- python3 bench.py --no-cpu --no-k1-1536 --no-query-sweep --steps 2" > gpurun_out/${R}_rocprofv3_pmc_insts_sa$sa.txt
rm -rf gpurun_out/prof_${R}_insts_$sa
grep 'k_fused<OpF32<0>, 0' gpurun_out/${R}_rocprofv3_pmc_insts_sa$sa.txt
done
