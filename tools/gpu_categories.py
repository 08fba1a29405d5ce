"""GPU time per step by kernel category from a rocprofv3 kernel_stats CSV (argv[1]); argv[2] = number of steps traced"""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
cat, cnt = defaultdict(float), defaultdict(int)


def c(n):
    if 'spconv_fwd' in n or 'spconv_tiles' in n or 'spconv_msplit' in n: return 'conv fwd/dgrad'
    if 'tile_order' in n: return 'voxelize/rulebook'
    if 'wgrad' in n: return 'wgrad (+reduce)'
    if 'bn_' in n: return 'batchnorm'
    if any(k in n for k in ('reduce_partials', 'accumulate_kernel', 'concat_kernel', 'split_kernel', 'pack_')): return 'executor misc'
    if 'linear_' in n: return 'dense heads'
    if 'adam_kernel' in n: return 'optimizer'
    if any(k in n for k in ('prop_', 'point_losses', 'npcs_loss', 'score_loss', 'gather_rows', 'scatter_rows', 'segmented_', 'instance_iou', 'nms_')): return 'proposal stage / losses / gather'
    if 'ball_query' in n or 'bq_' in n: return 'ball query'
    if 'ccl_' in n: return 'ccl'
    if 'rocprim' in n or 'radix' in n.lower() or 'sort' in n.lower(): return 'sort/scan (rocprim, torch)'
    if 'Cijk' in n: return 'GEMM (hipBLASLt)'
    if any(k in n for k in ('vox_', 'hash', 'subm', 'down_', 'compact', 'lists_', 'bitmap_', 'level_counts', 'identity_rulebook', 'fill_two')): return 'voxelize/rulebook'
    if 'multi_tensor' in n or 'fused_adam' in n.lower(): return 'optimizer'
    if 'copyBuffer' in n or 'fillBuffer' in n: return 'memcpy/memset'
    if 'index' in n: return 'torch indexing'
    if 'at::native' in n: return 'torch elementwise/reduce'
    return 'other'


for r in rows:
    k = c(r['Name'])
    cat[k] += float(r['TotalDurationNs'])
    cnt[k] += int(r['Calls'])
for k, v in sorted(cat.items(), key=lambda kv: -kv[1]):
    print(f"{k:34s} {v / 1e6 / steps:7.3f} ms/step  {cnt[k] / steps:7.1f} launches/step")
print(f"{'total':34s} {sum(cat.values()) / 1e6 / steps:7.3f} ms/step  {sum(cnt.values()) / steps:7.1f} launches/step")
