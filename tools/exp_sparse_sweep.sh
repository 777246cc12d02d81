for s in 0 16 32 64 128 256 512; do
  echo -n "sparse_max=$s: "
  HC_SPARSE_MAX=$s python bench.py --steps 5 --warmup 1 --cpu-sample 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['all_kernels_ms_per_launch']['k_cascade'], d['parity']['bit_exact'])"
done
