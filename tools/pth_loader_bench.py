#!/usr/bin/env python
"""The on-disk format end to end: ``.pth`` scene files -> GAPartNetInst (gapartnet.yaml's data module) -> DevicePrefetcher ->
training steps at 20k points per scene, bs 8.  (VERDICT r3 "missing" 6: the bench feeds HBM-resident synthetic scenes only;
the dataset itself is absent, so the files are written here in the reference's format - the 6-tuple of
dataset/process_tools/convert_rendered_into_input.py:156-158 that dataset/gapartnet.py:208-229 reads back with torch.load.)

Measures, for the same model / batch size as bench.py:
  * steps/s with the loader in the loop (worker processes read + prepare scenes, the prefetcher voxelises on the GPU), for
    both loader modes: per-scene CPU preparation as in the reference (device_pipeline=False) and raw hand-over with the
    per-batch GPU pipeline (device_pipeline=True);
  * the loader alone (no model), scenes/s - the rate the host side can sustain;
  * bench.py's resident-batch figure for comparison comes from its own line.

    python tools/pth_loader_bench.py [--scenes 64] [--workers 8] [--epochs 3]
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch


def write_dataset(root, n_train, n_eval, n_points):
    from tests.golden.recipe import scene_arrays
    for split, n, seed0 in (("train", n_train, 100), ("val", n_eval, 5000), ("test_intra", n_eval, 6000), ("test_inter", n_eval, 7000)):
        d = os.path.join(root, split, "pth")
        os.makedirs(d, exist_ok=True)
        for i in range(n):
            xyz, rgb, sem, ins, npcs, pix = scene_arrays(seed0 + i, n_points)
            torch.save((xyz, rgb, sem, ins, npcs, pix), os.path.join(d, f"StorageFurniture_{seed0 + i:05d}_00_{i % 32:03d}.pth"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=384)
    ap.add_argument("--points", type=int, default=20000)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--modes", default="per_scene_cpu,device_pipeline,packed_cache")
    ap.add_argument("--root", default=None, help="(internal) dataset already written here: run the modes in THIS process")
    args = ap.parse_args()
    from gapartnet_amd.dataset.gapartnet import GAPartNetInst
    from gapartnet_amd.dataset.prefetch import DevicePrefetcher
    from gapartnet_amd.smoke import make_model
    from gapartnet_amd.trainer import move_batch
    device = torch.device("cuda:0")
    out = {"scenes": args.scenes, "points": args.points, "batch": args.batch, "workers": args.workers}
    if args.root is None:
        # the dataset is written once; every mode then runs in a process of its own (a mode measured after another one in the same
        # process came out 6 - 25 % slower than alone, round 5: allocator / worker-pool state of the earlier mode)
        import subprocess
        root = tempfile.mkdtemp(prefix="gpn_pth_")
        try:
            t0 = time.perf_counter()
            write_dataset(root, args.scenes, 8, args.points)
            out["write_s"] = time.perf_counter() - t0
            size = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(root) for f in fs)
            out["mb_per_scene"] = size / (args.scenes + 24) / 1e6
            for mode in args.modes.split(","):
                cmd = [sys.executable, os.path.abspath(__file__), "--root", root, "--modes", mode, "--scenes", str(args.scenes),
                       "--points", str(args.points), "--batch", str(args.batch), "--workers", str(args.workers), "--epochs", str(args.epochs)]
                res = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
                if res.returncode != 0:
                    sys.stderr.write(res.stderr[-2000:])
                    raise SystemExit(f"mode {mode} failed")
                out.update({k: v for k, v in json.loads(res.stdout.strip().splitlines()[-1]).items() if "/" in k})
        finally:
            shutil.rmtree(root, ignore_errors=True)
        print(json.dumps(out))
        return
    root = args.root
    try:
        for mode in args.modes.split(","):
            device_pipeline = mode != "per_scene_cpu"
            dm = GAPartNetInst(root, max_points=args.points, train_batch_size=args.batch, val_batch_size=args.batch,
                               test_batch_size=args.batch, num_workers=args.workers, pos_jitter=0.1, color_jitter=0.3,
                               flip_prob=0.3, rotate_prob=0.3, device_pipeline=device_pipeline, packed_cache=mode == "packed_cache")
            dm.setup("fit")
            key = mode
            if mode == "packed_cache":
                t0 = time.perf_counter()
                dm.train_dataloader()  # first use: the cache file is written
                out["packed_cache/build_s"] = time.perf_counter() - t0
            # loader alone
            loader = dm.train_dataloader()
            n = 0
            t0 = time.perf_counter()
            for _ in range(2):
                for batch in loader:
                    n += len(batch)
            out[f"{key}/loader_only_scenes_per_s"] = n / (time.perf_counter() - t0)
            # loader + prefetcher + training steps
            torch.manual_seed(0)
            model = make_model((0, 0)).to(device).train()
            opt = model.configure_optimizers()
            steps, times = 0, []
            for epoch in range(args.epochs + 1):  # epoch 0 = warm-up (allocator, worker start)
                loader = dm.train_dataloader()
                feed = DevicePrefetcher(loader, model, device, augmentation=dm.aug if device_pipeline else None)
                torch.cuda.synchronize()
                t_start, n = None, 0
                for i, batch in enumerate(feed):
                    batch = move_batch(batch, device)
                    opt.zero_grad(set_to_none=True)
                    loss = model.training_step(batch, i)
                    loss.backward()
                    opt.step()
                    n += 1
                    if i == 3:  # steady state of the epoch: the worker start-up at its head (the reference re-creates its
                        torch.cuda.synchronize()  # workers every epoch too) is left out
                        t_start = time.perf_counter()
                torch.cuda.synchronize()
                if epoch > 0 and t_start is not None and n > 8:
                    times.append((time.perf_counter() - t_start) / (n - 4))
                    steps += n - 4
            assert bool(torch.isfinite(loss)), "loss is not finite"
            out[f"{key}/ms_per_step_with_loader"] = float(np.median(times)) * 1e3
            out[f"{key}/point_clouds_per_s_with_loader"] = args.batch / float(np.median(times))
            out[f"{key}/steps_timed"] = steps
    finally:
        pass
    print(json.dumps(out))


if __name__ == "__main__":
    main()
