#!/bin/bash
# Round 6: SQ / TCC counters of the bench step, one pair at a time (rocprofv3 --pmc serialises kernels anyway), plans preloaded, in
# separate --pmc passes (tools/pmc_passes.sh); per-step sums by kernel family + derived utilisation lines.
#   usage (inside gpurun): bash tools/pmc_r06.sh
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_r06; mkdir -p $O
cd $R
rm -f /tmp/plans_pmc.json
python bench.py --no-pmc --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-sustained --no-f32-leg --no-3d-leg --no-mix-layers --streams 1 --plans /tmp/plans_pmc.json > /dev/null 2>&1
bash tools/pmc_passes.sh $O -- python $R/bench.py --no-pmc --steps 10 --warmup 3 --no-cpu-baseline --no-parity --no-sustained --no-f32-leg --no-3d-leg --no-mix-layers --streams 1 --plans /tmp/plans_pmc.json
cd $R
python tools/pmc_sum.py $O 6 > $O/pmc_r06_f16x3_bench_sums.txt 2>&1
python - <<PY >> $O/pmc_r06_f16x3_bench_sums.txt
import re
v = {}
for ln in open("$O/pmc_r06_f16x3_bench_sums.txt"):
    m = re.match(r'(\S+)\s+\((\d+) steps\): (.*)', ln)
    if m:
        v[m.group(1)] = dict((kv.split('=')[0], float(kv.split('=')[1])) for kv in m.group(3).split('  '))
print()
print('derived, per step of one pair (forward + decode + class NMS), conv engine launches unless noted:')
if 'SQ_VALU_MFMA_BUSY_CYCLES' in v:
    mf = v['SQ_VALU_MFMA_BUSY_CYCLES']['conv engine']
    n_mfma = mf / 32.0
    print('  v_mfma_f32_32x32x16_f16 executed: %.4g (= MFMA busy cycles / 32) = %.1f GFLOP issued = 3 x %.1f GFLOP algorithmic' % (n_mfma, n_mfma * 32768 / 1e9, n_mfma * 32768 / 3e9))
    if 'GRBM_GUI_ACTIVE' in v:
        ga = v['GRBM_GUI_ACTIVE']['conv engine']
        print('  MFMA pipe utilisation while conv kernels run: busy SIMD-cycles / (1024 SIMDs x GUI-active cycles): %.3f (GRBM_GUI_ACTIVE summed over the 8 XCDs: / 8 applied)' % (mf / (1024.0 * ga / 8.0)))
    if 'SQ_BUSY_CYCLES' in v:
        print('  MFMA busy / SQ busy cycles: %.3g' % (mf / v['SQ_BUSY_CYCLES']['conv engine']))
if 'SQ_WAVE_CYCLES' in v:
    wc = v['SQ_WAVE_CYCLES']['conv engine']
    for k in ('SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_WAIT_INST_LDS'):
        if k in v:
            print('  %-20s / SQ_WAVE_CYCLES = %.3f' % (k, v[k]['conv engine'] / wc))
if 'TCC_HIT_sum' in v and 'TCC_MISS_sum' in v:
    h, m_ = v['TCC_HIT_sum']['all kernels'], v['TCC_MISS_sum']['all kernels']
    print('  L2 hit rate, all kernels: %.3f' % (h / (h + m_)))
if 'SQ_LDS_BANK_CONFLICT' in v and 'SQ_LDS_IDX_ACTIVE' in v:
    print('  LDS bank-conflict cycles / LDS active cycles: %.4f' % (v['SQ_LDS_BANK_CONFLICT']['conv engine'] / max(v['SQ_LDS_IDX_ACTIVE']['conv engine'], 1)))
PY
rm -rf $O/pass*/          # the counter CSVs are hundreds of MB: gpurun merges at most 64 MiB back
cat $O/pmc_r06_f16x3_bench_sums.txt; cat $O/fail.log 2>/dev/null
