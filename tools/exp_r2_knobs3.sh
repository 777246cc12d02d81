for kv in "HC_ROW_STAGES=2" "HC_ROW_STAGES=3" "HC_ROW_STAGES=4" "HC_SPARSE_MAX=48" "HC_SPARSE_MAX=64" "HC_SPARSE_MAX=24"; do
  env $kv python bench.py --steps 20 --cpu-sample 0 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$kv',d['value'],d['roofline']['all_kernels_ms_per_launch']['k_cascade'],d['parity']['bit_exact'])"
done
