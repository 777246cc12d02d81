for r2 in 8 16; do for r1 in 8 16 32; do for sp in 24 48; do
  echo -n "rows1=$r1 rows2=$r2 sparse=$sp: "
  HC_BAND_ROWS1=$r1 HC_BAND_ROWS2=$r2 HC_SPARSE_MAX=$sp python bench.py --steps 5 --warmup 1 --cpu-sample 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['all_kernels_ms_per_launch']['k_cascade'], d['parity']['bit_exact'])"
done; done; done
