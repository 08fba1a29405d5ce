#!/bin/bash
# temporary job: fused scene preparation
cd /root/repo
mkdir -p gpurun_out/r05s
timeout 600 python -m pytest tests/test_gpu_sceneprep.py tests/test_golden_loader.py tests/test_gpu_model.py -x -q -m gpu > gpurun_out/r05s/pytest_s.txt 2>&1
tail -15 gpurun_out/r05s/pytest_s.txt
for i in 1 2; do
  for v in 0 1; do
    GPN_SCENE_PREPARE=$v timeout 600 python tools/pth_loader_bench.py --modes packed_cache,device_pipeline 2>gpurun_out/r05s/pth_$v.err | tail -1 > gpurun_out/r05s/pth_${v}_$i.json
    echo "fused=$v: $(python -c "import json;d=json.load(open('gpurun_out/r05s/pth_${v}_$i.json'));print({k:round(v,2) for k,v in d.items() if 'with_loader' in k})")"
  done
done
