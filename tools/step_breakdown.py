#!/usr/bin/env python3
"""Per-launch breakdown of the last training step in a rocprofv3 kernel trace CSV."""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
names = [r['Kernel_Name'] for r in rows]
sgd = [i for i, n in enumerate(names) if 'sgd_kernel' in n]
a, b = sgd[-2] + 1, sgd[-1] + 1
step = rows[a:b]
t0 = int(step[0]['Start_Timestamp']); t1 = int(step[-1]['End_Timestamp'])
tot = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step) / 1e3
print(f"step wall {(t1 - t0) / 1e3:.1f} us, kernels {len(step)}, sum of kernel time {tot:.1f} us")
def short(n):
    m = re.search(r'(conv3x3_wch_kernel|conv3x3_glds_w4_kernel|conv3x3_c16_kernel|conv3x3_p64_kernel|conv3x3_kernel|upconv_wch_kernel|pack_layers_kernel|unpack_layers_kernel|sum_replicas_kernel|igemm_kernel|wgrad_pp_group_kernel|wgrad_group_kernel|wgrad_up_pp_kernel|wgrad_pp_kernel|wgrad_kernel|bn_act_pool_kernel|bn_act_kernel|unpool_bn_bwd_apply_kernel|bn_bwd_reduce_kernel|bn_bwd_apply_stream_kernel|bn_bwd_apply_kernel|head_fwd64_kernel|grad_overflow_kernel|colstats_kernel|unpool_add_kernel|head_fwd_kernel|head_bwd_kernel|pack_conv3x3_kernel|pack_upconv_kernel|unpack_wgrad\w+|sgd_kernel|bn_finalize_kernel|nchw_to_nhwc_kernel|wce_\w+_kernel|copyBuffer|Fill|vectorized|unrolled|manual)', n)
    s = m.group(1) if m else n[:30]
    m2 = re.search(r'(?:igemm|conv3x3)_kernelI(DF16b|f)Li(\d+)ELi(\d+)E', n)
    if m2: s += f"<{m2.group(2)},{m2.group(3)}>"
    return s
agg = collections.OrderedDict()
for r in step:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    k = short(r['Kernel_Name'])
    agg.setdefault(k, [0, 0.0]); agg[k][0] += 1; agg[k][1] += d
    if len(sys.argv) > 2 and d > float(sys.argv[2]):
        print(f"{k:30s} grid {r['Grid_Size_X']:>9s} x{r['Grid_Size_Y']:>5s} vgpr {r['VGPR_Count']:>4s}+{r['Accum_VGPR_Count']:>3s} {d:8.1f} us")
print("---- aggregate over the step")
for k, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:30s} x{n:3d} {d:9.1f} us  {100 * d / tot:5.1f} %")
