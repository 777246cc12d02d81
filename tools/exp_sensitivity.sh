# usage: exp_sensitivity.sh  -- times k_cascade for each ugproj-detector_amd/exp_*.so variant (sensitivity runs)
cd $GRAFT_REPO_ROOT
cp ugproj-detector_amd/libhip_cascade.so /tmp/keep.so
for v in ugproj-detector_amd/exp_*.so; do
  cp $v ugproj-detector_amd/libhip_cascade.so
  echo -n "$v: "
  timeout -k 10 200 python bench.py --steps 10 --warmup 2 --cpu-sample 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['all_kernels_ms_per_launch']['k_cascade'], d['parity']['bit_exact'])" || exit 1
done
cp /tmp/keep.so ugproj-detector_amd/libhip_cascade.so
