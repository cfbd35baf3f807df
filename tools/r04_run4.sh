timeout 600 python -m pytest tests/test_hip_ops.py -q -k "4wave" -x 2>&1 | tail -5 > gpurun_out/t4w.log
for m in 36928 295424; do tools/probes/_bin/g4w_probe $m 2; done > gpurun_out/g4w_probe.txt 2>&1
