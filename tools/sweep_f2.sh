for nc in 0 13; do
  if [ $nc = 0 ]; then export HX_FUSED2=0; else export HX_FUSED2=1; export HX_FUSED2_NC=$nc; fi
  python bench.py --no-cpu --no-k1-1536 --steps 5 > gpurun_out/r2_f2_$nc.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/r2_f2_$nc.json"))
print("nc $nc build", d["build_sec"], d["build_kernels"]["k_fused<insert>"]["GBps"], d["build_kernels"]["k_fused<insert>"]["ms"], "links", d["build_kernels"]["k_links"]["ms"], "qps", d["value"], d["roofline"]["frac"], d["recall_at_10"], d["fused"])
PY
done
