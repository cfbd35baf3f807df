#!/bin/bash
# usage (GPU box): bash tools/pmc_any.sh <kernel substring> <python script> [args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
KS=$1; shift
mkdir -p $R/gpurun_out/pmc
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmc/any_p1 -o p -- python3 $R/"$@" > $R/gpurun_out/pmc/any_p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d $R/gpurun_out/pmc/any_p2 -o p -- python3 $R/"$@" > $R/gpurun_out/pmc/any_p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_TRANS SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC --output-format csv -d $R/gpurun_out/pmc/any_p3 -o p -- python3 $R/"$@" > $R/gpurun_out/pmc/any_p3.log 2>&1
cd $R
python3 - "$KS" <<'PY'
import csv, collections, sys, os
for p in (1, 2, 3):
    f = 'gpurun_out/pmc/any_p%d/p_counter_collection.csv' % p
    if not os.path.exists(f):
        print('pass', p, 'missing'); continue
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(list)
    for r in rows:
        if sys.argv[1] in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    print('pass', p, {k: round(sum(v) / len(v)) for k, v in agg.items()})
PY
tail -2 gpurun_out/pmc/any_p1.log
