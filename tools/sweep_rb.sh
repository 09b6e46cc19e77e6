# builds kernel variants on the GPU box and benches each (tuning only; the shipped defaults are in hx_fused_core.h)
cp pgvector-rx_amd/libhnswrx.so /tmp/libhnswrx_default.so
for v in "4 4" "6 3" "8 2" "5 4" "6 4"; do
  set -- $v
  HX_CFLAGS="-DFUSED_RB=$1 -DFUSED_MINW=$2 -DLC_RB=4" python pgvector-rx_amd/build.py --force > /dev/null 2>&1
  python bench.py --no-cpu --no-k1-1536 --steps 5 > gpurun_out/r2_rb_$1_$2.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/r2_rb_$1_$2.json"))
print("RB $1 MINW $2 build", d["build_sec"], d["build_kernels"]["k_fused<insert>"]["GBps"], "qps", d["value"], d["roofline"]["frac"], d["recall_at_10"])
PY
done
cp /tmp/libhnswrx_default.so pgvector-rx_amd/libhnswrx.so
