#!/usr/bin/env python
"""Settle the spconv conventions that no test in this repo can settle (DESIGN.md "Oracle", README "What is not verified").

spconv and epic_ops are absent from the reference tree, so three conventions of the sparse convolutions are CHOSEN here
 and pass every self-consistency test either way:
  1. tap orientation of the k=3 submanifold convs: cross-correlation (tap index = delta + 1 per axis) or its mirror;
  2. the tap of a k=2 s=2 conv (and of its inverse): tap = c mod 2 per axis, or its mirror;
  3. the weight layout of the checkpoint tensors: [Cout, kD, kH, kW, Cin] (spconv 2.x) or [kD, kH, kW, Cin, Cout] (1.x) -
     detected by shape on load (spconv/pytorch/__init__.py), listed here for completeness.
Only the released checkpoint decides: a wrong orientation destroys its accuracy.  Given ``release.ckpt`` and one or more
``.pth`` scenes of the dataset, this script evaluates the semantic segmentation of the backbone under the four
(orientation x stride-2 tap) combinations and prints the per-point accuracy of each; the right combination is the one far
above the others.  Mirroring is applied to the CHECKPOINT TENSORS (flip of the kernel axes), so the library is untouched.

    python tools/check_ckpt_orientation.py --ckpt ckpt/release.ckpt --scenes data/GAPartNet_All/test_intra/*.pth [--device cuda:0]
"""
import argparse
import glob
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch


def mirrored_state(state, mirror_k3: bool, mirror_k2: bool):
    """checkpoint tensors with the kernel axes of the 5-d conv weights flipped (both layouts handled by shape)"""
    out = {}
    for key, w in state.items():
        if w.dim() == 5:
            spatial = (1, 2, 3) if w.shape[1] == w.shape[2] == w.shape[3] else (0, 1, 2)
            k = w.shape[spatial[0]]
            if (k == 3 and mirror_k3) or (k == 2 and mirror_k2):
                w = torch.flip(w, dims=spatial)
        out[key] = w
    return out


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--ckpt", required=True)
    ap.add_argument("--scenes", nargs="+", required=True, help=".pth scene files (globs allowed)")
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--yaml-args", default=os.path.join(ROOT, "tests", "golden", "yaml_init_args.json"),
                    help="parsed init args of gapartnet.yaml (model section)")
    args = ap.parse_args()
    import json

    from gapartnet_amd.dataset.gapartnet import compact_instance_labels, generate_inst_info, load_data
    from gapartnet_amd.network.model import GAPartNet
    from gapartnet_amd.structure.point_cloud import PointCloud

    device = torch.device(args.device)
    with open(args.yaml_args) as fh:
        cfg = json.load(fh)["model"]["init_args"]
    cfg["ckpt"] = ""
    paths = sorted(p for pat in args.scenes for p in glob.glob(pat))
    assert paths, "no scene file matched"
    scenes = [generate_inst_info(compact_instance_labels(load_data(p))).to_tensor().to(device) for p in paths]
    state = torch.load(args.ckpt, map_location="cpu", weights_only=False)["state_dict"]
    rows = []
    for mirror_k3, mirror_k2 in itertools.product((False, True), (False, True)):
        model = GAPartNet(**cfg)
        missing, unexpected = model.load_state_dict(mirrored_state(state, mirror_k3, mirror_k2), strict=False)
        assert not missing, f"checkpoint lacks {missing[:5]}"
        model = model.to(device).eval()
        correct = on_part_correct = total = on_part = 0
        with torch.no_grad():
            for i in range(0, len(scenes), 4):
                batch = PointCloud.collate(scenes[i:i + 4], voxel_size=model.voxel_size)
                preds = model.forward_sem_seg(model.forward_backbone(batch)).argmax(-1)
                labels = batch.sem_labels
                hit = preds == labels
                correct += int(hit.sum()); total += labels.numel()
                on_part_correct += int((hit & (labels > 0)).sum()); on_part += int((labels > 0).sum())
        rows.append((mirror_k3, mirror_k2, correct / max(total, 1), on_part_correct / max(on_part, 1)))
    print(f"{len(scenes)} scene(s), checkpoint {args.ckpt}")
    print("k3 taps mirrored | k2s2 taps mirrored | all-point accuracy | part-point accuracy")
    for mk3, mk2, acc, pacc in rows:
        tag = "   <- this repo's convention" if not mk3 and not mk2 else ""
        print(f"{str(mk3):>16} | {str(mk2):>18} | {acc:18.4f} | {pacc:19.4f}{tag}")
    best = max(rows, key=lambda r: r[3])
    if best[0] or best[1]:
        print("\nThe checkpoint prefers a MIRRORED convention: flip the corresponding tap order in csrc/rulebook.hip "
              "(subm3_lookup_kernel: tap k <-> 26 - k; down_mark_kernel: tap <-> 7 - tap) or load checkpoints through mirrored_state().")
    else:
        print("\nThe repo's conventions (cross-correlation, tap = c mod 2) are the checkpoint's.")


if __name__ == "__main__":
    main()
