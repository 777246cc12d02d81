for s in 1 2 3 4 5 6 8 10 12 16 22; do
  echo -n "stages<=$s: "
  HC_DEBUG_MAX_STAGES=$s python bench.py --steps 5 --warmup 1 --cpu-sample 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['all_kernels_ms_per_launch']['k_cascade'])"
done
