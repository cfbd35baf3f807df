"""Direct evidence for (or against) "the large GEMMs are power-limited" (VERDICT r4 item 9): sclk / socket power sampled by rocm-smi
in a side thread while ONE GEMM shape is launched back to back for a few seconds, per kernel form and per operand fill.

    python tools/power_probe.py [seconds per arm] > profiles/r05_power_probe.txt

Arms: idle; qkv shape (N 2304, K 768) at M = 36 928 and 295 424 with the 4-wave persistent kernel (tile_hint 42), the 8-wave kernel
(32), the 8-wave ablation without LDS-DMA (7: same MFMA stream, no data movement; wrong results), each on uniform(-1, 1) operands and
on zero operands (the guide's DVFS note: zero operands toggle no multiplier bits).  Reports TFLOP/s from torch events over the whole
arm, and min / median / max of the sampled sclk and power."""
import json
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vitcap_amd import ops


def smi_sample():
    """One rocm-smi reading -> (sclk MHz, power W) or None."""
    try:
        out = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--json'], capture_output=True, text=True, timeout=5).stdout
        d = json.loads(out)
        card = d[sorted(d.keys())[0]]
        sclk = pw = None
        for k, v in card.items():
            kl = k.lower()
            if 'sclk' in kl and 'clock' in kl and sclk is None:
                sclk = float(str(v).strip('()').lower().replace('mhz', ''))
            if 'power' in kl and '(w)' in kl and pw is None:
                try:
                    pw = float(v)
                except ValueError:
                    pass
        return sclk, pw, card
    except Exception as e:          # noqa
        return None


def run_arm(name, fn, seconds, gflop):
    samples = []
    stop = threading.Event()

    def sampler():
        while not stop.is_set():
            s = smi_sample()
            if s:
                samples.append(s[:2])
            time.sleep(0.05)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = threading.Thread(target=sampler, daemon=True)
    t.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 0
    t0 = time.time()
    e0.record()
    while time.time() - t0 < seconds:
        for _ in range(50):
            fn()
        n += 50
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop.set()
    t.join()
    us = e0.elapsed_time(e1) / max(n, 1) * 1e3
    sc = sorted(s[0] for s in samples if s[0] is not None)
    pw = sorted(s[1] for s in samples if s[1] is not None)
    med = lambda v: v[len(v) // 2] if v else float('nan')
    print('%-58s %8.1f us %6.0f TF | sclk MHz min %5.0f med %5.0f max %5.0f | power W min %5.0f med %5.0f max %5.0f | %d samples' % (
        name, us, gflop / us * 1e3 if gflop else 0.0, sc[0] if sc else 0, med(sc), sc[-1] if sc else 0, pw[0] if pw else 0, med(pw),
        pw[-1] if pw else 0, len(samples)), flush=True)


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
    s = smi_sample()
    print('rocm-smi fields:', None if s is None else {k: v for k, v in s[2].items() if 'clk' in k.lower() or 'power' in k.lower()}, flush=True)
    run_arm('idle (no kernels)', lambda: time.sleep(0.001), seconds, 0)
    N, K = 2304, 768
    for M in (36928, 295424):
        for fill in ('uniform', 'zero'):
            if fill == 'uniform':
                a = (torch.rand(M, K, device='cuda') * 2 - 1).to(torch.bfloat16)
                w = ((torch.rand(N, K, device='cuda') * 2 - 1) * 0.05).to(torch.bfloat16)
            else:
                a = torch.zeros(M, K, device='cuda', dtype=torch.bfloat16)
                w = torch.zeros(N, K, device='cuda', dtype=torch.bfloat16)
            bias = torch.rand(N, device='cuda')
            out = torch.empty(M, N, device='cuda', dtype=torch.bfloat16)
            gf = 2.0 * M * N * K / 1e9
            for label, hint in (('4-wave persistent (42)', 42), ('8-wave 256-row tiles (32)', 32), ('8-wave, no LDS-DMA in the loop (7, ablation)', 7)):
                run_arm('qkv M=%d %s, %s' % (M, fill, label), lambda: ops.gemm_bias_act(a, w, bias, act=0, out=out, tile_hint=hint), seconds, gf)
            del a, w, out


if __name__ == '__main__':
    main()
