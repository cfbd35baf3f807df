"""Which limiter holds the clock under the hot kernels?  (round 6)  Per arm: the kernel is launched back to back for a few seconds while a
side thread takes ONE `amd-smi metric --json` snapshot (power, clocks, voltage, throttle / violation status, temperatures) and rocm-smi
samples of sclk / socket power.
    python tools/throttle_probe.py [seconds per arm] > gpurun_out/throttle_probe.txt"""
import json
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vitcap_amd import ops
from tools.power_probe import smi_sample

HERE = os.path.dirname(os.path.abspath(__file__))


def amd_smi_snapshot():
    try:
        out = subprocess.run(['amd-smi', 'metric', '-g', '0', '--json'], capture_output=True, text=True, timeout=20).stdout
        d = json.loads(out)
        return d
    except Exception as e:      # noqa
        return {'error': repr(e)}


def slim(d):
    """keeps the power / clock / voltage / throttle parts of the snapshot"""
    if isinstance(d, dict) and 'gpu_data' in d:
        d = d['gpu_data']
    if isinstance(d, list) and d:
        d = d[0]
    if not isinstance(d, dict):
        return d
    keep = {}
    for k, v in d.items():
        kl = k.lower()
        if any(s in kl for s in ('power', 'clock', 'volt', 'throttle', 'violation', 'temperature', 'usage', 'energy')):
            keep[k] = v
    return keep or d


def run_arm(name, fn, seconds, gflop=0.0, sync=True):
    samples, snap = [], {}
    stop = threading.Event()

    def sampler():
        t0 = time.time()
        took = False
        while not stop.is_set():
            s = smi_sample()
            if s:
                samples.append(s[:2])
            if not took and time.time() - t0 > seconds * 0.4:
                snap['d'] = amd_smi_snapshot()
                took = True
            time.sleep(0.05)
    for _ in range(3):
        fn()
    if sync:
        torch.cuda.synchronize()
    t = threading.Thread(target=sampler, daemon=True)
    t.start()
    n = 0
    t0 = time.time()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < seconds:
        for _ in range(20):
            fn()
        n += 20
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop.set()
    t.join()
    us = e0.elapsed_time(e1) / max(n, 1) * 1e3
    sc = sorted(s[0] for s in samples if s[0] is not None)
    pw = sorted(s[1] for s in samples if s[1] is not None)
    med = lambda v: v[len(v) // 2] if v else float('nan')
    print('== %-50s %9.1f us %6.0f TF | sclk med %5.0f | power med %5.0f W' % (name, us, gflop / us * 1e3 if gflop else 0.0, med(sc), med(pw)), flush=True)
    print(json.dumps(slim(snap.get('d')), indent=None)[:6000], flush=True)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 6.0
    run_arm('idle', lambda: time.sleep(0.002), seconds)
    M, N, K = 36928, 2304, 768
    a = (torch.rand(M, K, device='cuda') * 2 - 1).to(torch.bfloat16)
    w = ((torch.rand(N, K, device='cuda') * 2 - 1) * 0.05).to(torch.bfloat16)
    bias = torch.rand(N, device='cuda')
    out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
    gf = 2.0 * M * N * K / 1e9
    run_arm('qkv GEMM M=36928 uniform, 4-wave persistent', lambda: ops.gemm_bias_act(a, w, bias, act=0, out=out, tile_hint=42), seconds, gf)
    az, wz = torch.zeros_like(a), torch.zeros_like(w)
    run_arm('qkv GEMM M=36928 ZERO operands, 4-wave persistent', lambda: ops.gemm_bias_act(az, wz, bias, act=0, out=out, tile_hint=42), seconds, gf)
    x = torch.randn(M, 768, device='cuda')
    g, b = torch.ones(768, device='cuda'), torch.zeros(768, device='cuda')
    run_arm('layernorm768 M=36928', lambda: ops.layernorm(x, g, b, 1e-6), seconds)
    qkv = torch.randn(64 * 577, 2304, device='cuda').to(torch.bfloat16)
    run_arm('dense attention B=64 S=577', lambda: ops.attn_dense(qkv, 64, 577), seconds, 4.0 * 64 * 12 * 577 * 577 * 64 / 1e9)
    # the matrix pipe alone (separate process)
    p = subprocess.Popen([os.path.join(HERE, 'probes', '_bin', 'mfma_power'), '1', str(seconds + 2), '2'], stdout=subprocess.PIPE, text=True)
    time.sleep(seconds * 0.5)
    print('== pure MFMA 32x32x16, random operands, 2 waves per SIMD (separate process)')
    print(json.dumps(slim(amd_smi_snapshot()), indent=None)[:6000], flush=True)
    print(p.communicate()[0].strip())


if __name__ == '__main__':
    main()
