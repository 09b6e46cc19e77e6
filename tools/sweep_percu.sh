for pc in 15 14 13 12 10; do
  HX_QUERY_PER_CU=$pc python bench.py --no-cpu --no-k1-1536 --steps 5 > gpurun_out/r2_pc_$pc.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/r2_pc_$pc.json"))
print("query per CU $pc qps", d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
PY
done
for q in 11520 15360 30720; do
  python bench.py --no-cpu --no-k1-1536 --steps 5 --queries $q > gpurun_out/r2_q_$q.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/r2_q_$q.json"))
print("queries $q qps", d["value"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
PY
done
