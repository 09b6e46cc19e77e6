for c in 600 768 1024; do
  HX_CLDS_INSERT=$c python bench.py --no-cpu --no-k1-1536 --steps 3 > gpurun_out/r2_sw_clds$c.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/r2_sw_clds$c.json"))
print("clds_insert $c build", d["build_sec"], d["build_kernels"]["k_fused<insert>"], "qps", d["value"], d["roofline"]["frac"])
PY
done
for c in 256 384; do
  HX_CLDS_QUERY=$c python bench.py --no-cpu --no-k1-1536 --steps 5 > gpurun_out/r2_sw_cldsq$c.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/r2_sw_cldsq$c.json"))
print("clds_query $c qps", d["value"], d["roofline"]["frac"], d["fused"])
PY
done
