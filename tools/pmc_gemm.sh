#!/bin/bash
# usage (GPU box): bash tools/pmc_gemm.sh "<hints>" M N K act out_f32 res
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
HINTS=$1; shift
for h in $HINTS; do
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmc/h${h}_p1 -o p -- python3 $R/tools/gemm_one.py $h "$@" 3 > $R/gpurun_out/pmc/h${h}_p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVES --output-format csv -d $R/gpurun_out/pmc/h${h}_p2 -o p -- python3 $R/tools/gemm_one.py $h "$@" 3 > $R/gpurun_out/pmc/h${h}_p2.log 2>&1
done
cd $R
python3 - "$HINTS" <<'PY'
import csv, collections, sys
for h in sys.argv[1].split():
    for p in (1, 2):
        rows = list(csv.DictReader(open('gpurun_out/pmc/h%s_p%d/p_counter_collection.csv' % (h, p))))
        agg = collections.defaultdict(list)
        for r in rows:
            if 'gemm' in r['Kernel_Name']:
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
        print('hint', h, {k: round(sum(v) / len(v)) for k, v in agg.items()})
PY
