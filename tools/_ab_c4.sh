for rep in 1 2 3; do for lib in 3d-reconstruction-from-point-cloud_amd/libpt_hip.so tools/_ab/libpt_sw_old.so; do
  PT_HIP_LIB=$lib timeout -k 10 200 python bench.py --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys; a=json.loads(sys.stdin.read()); print('$lib', round(a['ms_per_step'],2), a['kernels_ms']['knn_query'])"
done; done
