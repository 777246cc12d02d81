for cfg in "HC_PAIR=0" "HC_PAIR=1 HC_BAND_ROWS1=8" "HC_PAIR=1 HC_BAND_ROWS1=16" "HC_PAIR=1 HC_BAND_ROWS1=8 HC_SPARSE_MAX=48" "HC_PAIR=1 HC_BAND_ROWS1=16 HC_SPARSE_MAX=48" "HC_PAIR=1 HC_BAND_ROWS1=8 HC_SPARSE_MAX=16"; do
  echo -n "$cfg: "
  env $cfg python bench.py --steps 10 --warmup 2 --cpu-sample 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['all_kernels_ms_per_launch']['k_cascade'], d['parity']['bit_exact'])"
done
