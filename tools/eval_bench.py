#!/usr/bin/env python
"""Throughput of the evaluation path (BASELINE.json config 2: full pipeline, 20k-point scenes, bs 4, eval mode): validation steps
(backbone -> point heads -> dual-set clustering -> ScoreNet / NPCS-Net -> score filter -> NMS) over the three loaders of the data
module, and the epoch end (AP at ten IoU thresholds + mIoU on the device).  Reference: network/model.py:667-692, 694-857.
release.ckpt is absent: name-keyed seeded weights with non-trivial BatchNorm statistics (tests/golden/recipe.py), so that the
network predicts more than one class and proposals exist.

    python tools/eval_bench.py [--batch 4] [--points 20000] [--steps 12]        # one JSON line
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--points", type=int, default=20000)
    ap.add_argument("--steps", type=int, default=12, help="validation steps per loader and epoch")
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--lib-knobs", type=str, default="", help="see bench.py (e.g. bn_fusion=0: BatchNorm launches instead of conv epilogues)")
    ap.add_argument("--ab-bn-fusion", action="store_true",
                    help="alternate epochs with the BatchNorm in the conv epilogues (inference passes, round 6) and as launches of its "
                         "own; reports the median step time of each")
    args = ap.parse_args()
    import importlib.util
    spec = importlib.util.spec_from_file_location("gpn_bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    bench.apply_lib_knobs(args.lib_knobs)
    from gapartnet_amd.dataset.prefetch import DevicePrefetcher
    from gapartnet_amd.smoke import make_batch, make_model
    from tests.golden import recipe
    dev = torch.device("cuda:0")
    model = make_model((0, 0)).eval()
    model.load_state_dict(recipe.name_keyed_state(model))
    model = model.to(dev)
    logged = {}
    model._log_sink = lambda name, value, bs, sync: logged.__setitem__(name, value)
    model.defer_validation_outputs = True  # (what gapartnet_amd.trainer.Trainer's evaluation loop sets)
    pools = [[[pc.to(dev) for pc in make_batch(args.batch, args.points, seed0=2000 + 1000 * l + 10 * j)] for j in range(2)]
             for l in range(3)]
    step_ms, end_ms, kept = [], [], 0
    by_mode = {0: [], 1: []}
    from gapartnet_amd import _C
    with torch.no_grad():
        for epoch in range((2 * args.epochs if args.ab_bn_fusion else args.epochs) + 1):  # epoch 0 warms up
            if args.ab_bn_fusion:
                _C.lib().gpn_net_bn_fusion(epoch % 2)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for l in range(3):
                # (as the Trainer's evaluation loop does: the next batch is prepared on the side stream while this one runs)
                feed = DevicePrefetcher((pools[l][i % 2] for i in range(args.steps)), model, dev)
                for i, batch in enumerate(feed):
                    model.validation_step(batch, i, l)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            model._resolve_pending_outputs()
            last = model.validation_step_outputs[-1][-1][2]
            kept = int(last.score_preds.shape[0]) if last is not None else 0
            model.on_validation_epoch_end()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            if epoch > 0:
                step_ms.append((t1 - t0) / (3 * args.steps) * 1e3)
                end_ms.append((t2 - t1) * 1e3)
                by_mode[epoch % 2].append(step_ms[-1])
    step_ms.sort(); end_ms.sort()
    ms = step_ms[len(step_ms) // 2]
    if args.ab_bn_fusion:
        _C.lib().gpn_net_bn_fusion(1)
        med = lambda v: sorted(v)[len(v) // 2]
        print(json.dumps({"ms_per_validation_step": {"batchnorm_in_conv_epilogues": med(by_mode[1]), "batchnorm_launches": med(by_mode[0])},
                          "epochs_each": args.epochs, "all": by_mode, "batch": args.batch, "points": args.points}))
        return
    print(json.dumps({"metric": "point-clouds/sec (20k pts, validation step, eval mode)", "value": args.batch / ms * 1e3,
                      "ms_per_validation_step": ms, "epoch_end_ms": end_ms[len(end_ms) // 2],
                      "steps_per_epoch": 3 * args.steps, "batch": args.batch, "points": args.points,
                      "proposals_kept_last_step": kept, "mAP_logged": float(logged.get("val/mAP", float("nan"))),
                      "config": "BASELINE config 2 shape (full pipeline eval, bs 4 x 20k), seeded weights (release.ckpt absent)"}))


if __name__ == "__main__":
    main()
