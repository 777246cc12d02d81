for v in "" _D4 _D8; do
  HC_LIB_PATH=$PWD/ugproj-detector_amd/libhip_cascade$v.so python bench.py --config 3 --cascade default --steps 10 --cpu-sample 0 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('default N${v:-=6}',d['value'],d['roofline']['all_kernels_ms_per_launch']['k_cascade'],d['parity']['bit_exact'])"
done
