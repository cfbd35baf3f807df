timeout 600 python -m pytest tests/test_hip_ops.py -q -k "4wave" -x 2>&1 | tail -15 > gpurun_out/t4w.log
for f in 0 1 2; do tools/probes/_bin/g4w_probe 36928 $f; done > gpurun_out/g4w_probe.txt 2>&1
for f in 1 2; do tools/probes/_bin/g4w_probe 295424 $f; done >> gpurun_out/g4w_probe.txt 2>&1
