#!/bin/bash
# usage (GPU box): bash tools/pmc_hot.sh <tag>
# MFMA-busy / instruction-mix / wait-fraction counters of the hot kernels -> gpurun_out/<tag>_mfma_busy_pmc.json
# (three `rocprofv3 --pmc` passes per workload, --kernel-trace only: SQ has 8 slots per pass).  Workloads: the greedy B=64
# step on ONE stream (bench.py --pipeline 0) and one cross-entropy training step.
TAG=${1:-r03}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc_${TAG}
P1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES"
P3="SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_TRANS SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
for wl in greedy train; do
  if [ $wl = greedy ]; then ARGS="--steps 2 --warmup 1 --pipeline 0 --isolated 0 --no-cpu-baseline --single-region --power 0"; else ARGS="--mode train --steps 1 --warmup 1"; fi
  i=0
  for P in "$P1" "$P2" "$P3"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/gpurun_out/pmc_${TAG}/${wl}_p$i -o p -- python3 $R/bench.py $ARGS > $R/gpurun_out/pmc_${TAG}/${wl}_p$i.log 2>&1
  done
done
cd $R
python3 - $TAG <<'PY'
import csv, collections, json, os, sys
tag = sys.argv[1]
res = {}
for wl in ('greedy', 'train'):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for p in (1, 2, 3):
        d = 'gpurun_out/pmc_%s/%s_p%d' % (tag, wl, p)
        f = None
        for root, _, files in os.walk(d):
            for x in files:
                if x.endswith('counter_collection.csv'):
                    f = os.path.join(root, x)
        if not f:
            print('missing pass', wl, p); continue
        for r in csv.DictReader(open(f)):
            n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0]
            per[n][r['Counter_Name']].append(float(r['Counter_Value']))
    out = {}
    for n, c in per.items():
        a = {k: sum(v) / len(v) for k, v in c.items()}
        a['launches_sampled'] = max(len(v) for v in c.values())
        wc = a.get('SQ_WAVE_CYCLES', 0.0)
        busy = a.get('SQ_BUSY_CYCLES', 0.0)
        d = {'counters_avg_per_launch': {k: round(v, 1) for k, v in a.items()}}
        if wc:
            d['wave_time_fractions'] = {'wait_any(parked: waitcnt/barrier)': round(a.get('SQ_WAIT_ANY', 0) / wc, 4),
                                        'wait_inst_any(issue stall)': round(a.get('SQ_WAIT_INST_ANY', 0) / wc, 4),
                                        'active_inst_any': round(a.get('SQ_ACTIVE_INST_ANY', 0) / wc, 4),
                                        'wait_inst_lds': round(a.get('SQ_WAIT_INST_LDS', 0) / wc, 4)}
        mf = a.get('SQ_VALU_MFMA_BUSY_CYCLES')
        if mf and a.get('GRBM_GUI_ACTIVE'):
            # MFMA_BUSY: matrix-pipe busy cycles summed over the chip's 1024 SIMDs (= 16 x the bf16 16x16x32 MFMA count, 32 x the 32x32x16
            # count); GRBM_GUI_ACTIVE: the kernel's duration in cycles summed over the 8 XCDs -> fraction of the chip's matrix-pipe cycles
            d['mfma_busy_frac_of_chip'] = round(mf / (1024.0 * a['GRBM_GUI_ACTIVE'] / 8.0), 4)
        if mf and a.get('SQ_BUSY_CU_CYCLES'):
            # the same while a CU has any wave resident (SQ_BUSY_CU_CYCLES: busy cycles summed over the 256 CUs; 4 SIMDs each)
            d['mfma_busy_frac_while_cu_busy'] = round(mf / (4.0 * a['SQ_BUSY_CU_CYCLES']), 4)
            if a.get('GRBM_GUI_ACTIVE'):
                d['cu_busy_frac'] = round(a['SQ_BUSY_CU_CYCLES'] / (256.0 * a['GRBM_GUI_ACTIVE'] / 8.0), 4)
        if a.get('SQ_ACTIVE_INST_VALU') and a.get('GRBM_GUI_ACTIVE'):
            # ACTIVE_INST_VALU counts quad-cycles with a VALU-class instruction (MFMA included) active, summed over SIMDs
            d['valu_class_active_frac_of_chip'] = round(4.0 * a['SQ_ACTIVE_INST_VALU'] / (1024.0 * a['GRBM_GUI_ACTIVE'] / 8.0), 4)
        ins = {k: a.get(k, 0.0) for k in ('SQ_INSTS_VALU', 'SQ_INSTS_MFMA', 'SQ_INSTS_LDS', 'SQ_INSTS_VMEM', 'SQ_INSTS_SALU', 'SQ_INSTS_VALU_TRANS')}
        tot = ins['SQ_INSTS_VALU'] + ins['SQ_INSTS_LDS'] + ins['SQ_INSTS_VMEM'] + ins['SQ_INSTS_SALU']
        if tot:
            d['instruction_mix'] = {k[9:].lower(): round(v / tot, 4) for k, v in ins.items()}
        out[n] = d
    res[wl] = out
json.dump(res, open('gpurun_out/%s_mfma_busy_pmc.json' % tag, 'w'), indent=1, sort_keys=True)
for wl, out in res.items():
    print('==', wl)
    for n, d in sorted(out.items(), key=lambda x: -x[1]['counters_avg_per_launch'].get('SQ_BUSY_CYCLES', 0) * x[1]['counters_avg_per_launch']['launches_sampled'])[:14]:
        print('%-58s mfma busy %s (while CU busy %s, CU busy %s)  %s' % (n[:58], d.get('mfma_busy_frac_of_chip'), d.get('mfma_busy_frac_while_cu_busy'), d.get('cu_busy_frac'), d.get('wave_time_fractions')))
PY
