for rs in 1 2 3; do for sp in 24 40; do
  echo -n "row_stages=$rs sparse=$sp: "
  HC_ROW_STAGES=$rs HC_SPARSE_MAX=$sp python bench.py --steps 5 --warmup 1 --cpu-sample 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['all_kernels_ms_per_launch']['k_cascade'], d['parity']['bit_exact'])"
done; done
