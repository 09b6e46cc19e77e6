# builds kernel variants on the GPU box and benches each (tuning only; the shipped defaults are in hx_fused_core.h)
cp pgvector-rx_amd/libhnswrx.so /tmp/libhnswrx_default.so
for v in "4" "3"; do
  HX_CFLAGS="-DFUSED_MINW_INS=$v" python pgvector-rx_amd/build.py --force > /dev/null 2>&1
  for clds in 600 700; do
  HX_CLDS_INSERT=$clds python bench.py --no-cpu --no-k1-1536 --steps 3 > gpurun_out/r2_ins_$v.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/r2_ins_$v.json"))
print("MINW_INS $v clds $clds build", d["build_sec"], d["build_kernels"]["k_fused<insert>"], d["build_kernels"]["k_links"]["ms"], "qps", d["value"])
PY
  done
done
cp /tmp/libhnswrx_default.so pgvector-rx_amd/libhnswrx.so
