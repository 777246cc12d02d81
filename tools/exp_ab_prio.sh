for i in 1 2 3; do for v in "" _DP; do
  HC_LIB_PATH=$PWD/ugproj-detector_amd/libhip_cascade$v.so python bench.py --steps 40 --cpu-sample 0 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('lib${v:-=base}',d['value'],d['roofline']['all_kernels_ms_per_launch']['k_cascade'])"
done; done
