cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -s -k "training_mode_backbone or step_switches or training_step_on_the_gpu_matches" 2>&1 | grep -v "^$" | tail -25 > gpurun_out/r04_new_tests.txt
python bench.py --steps 30 --warmup 10 > gpurun_out/r04_bench_a.json 2> gpurun_out/r04_bench_a.err
python tools/pth_loader_bench.py > gpurun_out/r04_pth_loader.json 2> gpurun_out/r04_pth_loader.err
python - <<'PY' > gpurun_out/r04_ckpt_tool.txt 2>&1
import json, os, sys, tempfile, subprocess, torch
sys.path.insert(0, os.getcwd())
from gapartnet_amd.network.model import GAPartNet
from tests.golden.recipe import scene_arrays
d = tempfile.mkdtemp()
cfg = json.load(open("tests/golden/yaml_init_args.json"))["model"]["init_args"]; cfg["ckpt"] = ""
torch.manual_seed(0)
m = GAPartNet(**cfg)
torch.save({"state_dict": m.state_dict()}, os.path.join(d, "release.ckpt"))
for i in range(2):
    torch.save(scene_arrays(40 + i, 20000), os.path.join(d, f"Box_{i:05d}_00_000.pth"))
print(subprocess.run([sys.executable, "tools/check_ckpt_orientation.py", "--ckpt", os.path.join(d, "release.ckpt"), "--scenes", os.path.join(d, "*.pth")], capture_output=True, text=True))
PY
tail -5 gpurun_out/r04_new_tests.txt; tail -c 600 gpurun_out/r04_bench_a.json; cat gpurun_out/r04_pth_loader.json; tail -3 gpurun_out/r04_pth_loader.err; tail -c 1500 gpurun_out/r04_ckpt_tool.txt
