# query kernel at 5 waves/SIMD (20 searches per CU = 5120 slots: 10 000 queries in two rounds instead of three); builds variants on the GPU box
cp pgvector-rx_amd/libhnswrx.so /tmp/libhnswrx_default.so
python bench.py --no-cpu --no-k1-1536 --no-query-sweep --steps 5 > gpurun_out/occ.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/occ.json')); print('default qps', d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'])"
for v in "5 3 256" "5 3 320" "5 4 256" "6 2 192"; do set -- $v
  HX_CFLAGS="-DFUSED_MINW=$1 -DFUSED_RB=$2" python pgvector-rx_amd/build.py --force > /dev/null 2>&1
  HX_CLDS_QUERY=$3 HX_DEBUG=1 python bench.py --no-cpu --no-k1-1536 --no-query-sweep --steps 5 > gpurun_out/occ.json 2>gpurun_out/occ.err
  grep 'mode 0' gpurun_out/occ.err | head -1
  python -c "
import json; d=json.load(open('gpurun_out/occ.json')); print('MINW $1 RB $2 clds $3 qps', d['value'], d['roofline']['frac'], d['roofline']['avg_launch_ms'], d['recall_at_10'])"
done
cp /tmp/libhnswrx_default.so pgvector-rx_amd/libhnswrx.so
