timeout 600 python -m pytest tests/test_hip_ops.py -q -k "4wave" -x 2>&1 | tail -5 > gpurun_out/t4w.log
for f in 1 2; do for m in 36928 295424; do tools/probes/_bin/g4w_probe $m $f; done; done 2>&1 | cut -c1-225 > gpurun_out/g4w_probe.txt
for mi in 8 7 6; do echo MI $mi; VITCAP_GEMM4W_MI=$mi tools/probes/_bin/g4w_probe 36928 2; done  2>&1 | cut -c1-225 >> gpurun_out/g4w_probe.txt
