#!/usr/bin/env python3
"""Fold the rocprofv3 --pmc passes (gpurun_out/pmc_*/.../*counter_collection.csv, one pass per counter set, produced by
tools/pmc_pass.sh on tools/run_hot_kernels.py) into profiles/<name>_pmc.json + a readable summary.
HBM bytes per launch = 2 * FETCH_SIZE (gfx950: the counter tallies 128-B requests of wide reads at 64 B, see
/opt/skills/guides/MI355X_MICROARCH.md "HBM") + WRITE_SIZE, both in KiB, averaged over the launches of the pass.
The library build the passes ran on (`orp_version`, written by tools/pmc_all.sh next to the passes) is stored under "_build":
bench.py only quotes these counters when the running library reports the same build.
usage: tools/make_pmc_json.py profiles/r01 gpurun_out/pmc_a gpurun_out/pmc_b ..."""
import collections, csv, glob, json, os, sys

dst, dirs = sys.argv[1], sys.argv[2:]
# the other hot-path kernels (tools/run_hot_kernels.py ops), keyed by kernel name without the suffix
OPS = ('convex_iou_kernel', 'convex_giou_kernel', 'minarearect_kernel', 'chamfer_nn_kernel', 'points_in_quad_aligned_kernel',
       'focal_fwd_kernel', 'point_assign_gt_kernel', 'point_assign_finish_kernel', 'max_iou_assign_kernel', 'gt_max_kernel',
       'gt_argmax_assign_kernel', 'box_iou_rotated_kernel', 'apaa_select_kernel')
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in glob.glob(d + '/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            key = ('dcn_bwd_scatter' if 'dcn_bwd_scatter' in k else
                   'dcn_bwd_input' if 'dcn_bwd_input' in k else 'dcn_bwd_weight' if 'dcn_bwd_weight' in k else
                   'dcn_fwd_half' if 'dcn_fwd_half' in k else
                   # dcn_fwd_split_kernel<MT, products, nchw, PLAIN>: PLAIN = the tower / FPN convolutions (no offsets)
                   'conv_split_pair' if ('dcn_fwd_split' in k and ', true>' in k) else 'dcn_fwd_split' if 'dcn_fwd_split' in k else
                   'dcn_fwd_pair' if 'dcn_fwd_mfma2' in k else 'nms_mask' if 'nms_mask' in k
                   else 'nms_sweep' if 'nms_sweep' in k else 'nms_rankprep' if 'nms_rankprep' in k else None)
            if key is None:
                for name in OPS:
                    if name in k:
                        key = name.replace('_kernel', '')
                        break
            if key:
                agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    e = dict(counters=m, launches=max(len(v) for v in c.values()))
    if 'FETCH_SIZE' in m and 'WRITE_SIZE' in m:
        e['fetch_bytes_per_launch'] = 2.0 * m['FETCH_SIZE'] * 1024
        e['write_bytes_per_launch'] = m['WRITE_SIZE'] * 1024
        e['hbm_bytes_per_launch'] = e['fetch_bytes_per_launch'] + e['write_bytes_per_launch']
    if 'SQ_VALU_MFMA_BUSY_CYCLES' in m and 'GRBM_GUI_ACTIVE' in m and m['GRBM_GUI_ACTIVE'] > 0:
        # MFMA busy is summed over 1024 SIMDs, GRBM_GUI_ACTIVE over 8 XCDs
        e['mfma_busy_frac'] = (m['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0) / (m['GRBM_GUI_ACTIVE'] / 8.0)
    if m.get('GRBM_GUI_ACTIVE', 0) > 0:
        simd_cycles = m['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0          # SIMD-cycles of the launch (SQ_* count quad-cycles)
        e['kernel_us_at_2p4ghz'] = m['GRBM_GUI_ACTIVE'] / 8.0 / 2400.0
        if 'SQ_ACTIVE_INST_VALU' in m:
            e['valu_busy_frac'] = 4.0 * m['SQ_ACTIVE_INST_VALU'] / simd_cycles
        if 'SQ_WAVE_CYCLES' in m:
            e['waves_per_simd'] = 4.0 * m['SQ_WAVE_CYCLES'] / simd_cycles
        if 'SQ_INSTS_VALU' in m:
            # one VALU instruction of a 64-wide wave occupies its SIMD's 16 lanes for 4 cycles
            e['valu_issue_frac'] = 4.0 * m['SQ_INSTS_VALU'] / simd_cycles
    if k.startswith('dcn_bwd'):
        e['batch'] = 2
        e['img'] = 1024
    if k.startswith('dcn_fwd') or k.startswith('conv_split'):
        e['batch'] = 1
        e['img'] = 1024
    out[k] = e
vers = sorted({open(f).read().strip() for d in dirs for f in glob.glob(os.path.join(os.path.dirname(d.rstrip('/')) or '.', 'pmc_version.txt'))})
out['_build'] = dict(orp_version=vers[0] if len(vers) == 1 else None, note='library build of the passes (tools/pmc_all.sh)')
json.dump(out, open(dst + '_pmc.json', 'w'), indent=1, sort_keys=True)
with open(dst + '_pmc.txt', 'w') as f:
    f.write('build: %s\n' % out['_build']['orp_version'])
    for k, e in sorted(out.items()):
        if k.startswith('_'):
            continue
        f.write('%s  launches %d\n' % (k, e['launches']))
        for n, v in sorted(e['counters'].items()):
            f.write('    %-28s %.4g\n' % (n, v))
        for n in ('fetch_bytes_per_launch', 'write_bytes_per_launch', 'hbm_bytes_per_launch', 'mfma_busy_frac', 'valu_busy_frac',
                  'valu_issue_frac', 'waves_per_simd', 'kernel_us_at_2p4ghz'):
            if n in e:
                f.write('    %-28s %.4g\n' % (n, e[n]))
print(open(dst + '_pmc.txt').read())
