cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/pmc
rocprofv3 -L > $R/gpurun_out/pmc/counters.txt 2>&1
for h in 2 3; do
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmc/h${h}_p1 -o p -- python3 $R/tools/gemm_one.py $h 36928 2304 768 0 0 0 3 > $R/gpurun_out/pmc/h${h}_p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVES --output-format csv -d $R/gpurun_out/pmc/h${h}_p2 -o p -- python3 $R/tools/gemm_one.py $h 36928 2304 768 0 0 0 3 > $R/gpurun_out/pmc/h${h}_p2.log 2>&1
done
ls -R $R/gpurun_out/pmc | head -30
