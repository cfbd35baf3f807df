"""Runs tools/probes/_bin/mfma_power per variant and samples rocm-smi (sclk, board power) beside it.
    python tools/mfma_power.py [seconds per arm]"""
import json
import os
import subprocess
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))


def smi():
    try:
        d = json.loads(subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--json'], capture_output=True, text=True, timeout=5).stdout)
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
        return sclk, pw
    except Exception:
        return None


def main():
    secs = sys.argv[1] if len(sys.argv) > 1 else '4'
    for wps in ('1', '2'):
        for v in ('0', '1', '2', '3'):
            p = subprocess.Popen([os.path.join(HERE, 'probes', '_bin', 'mfma_power'), v, secs, wps], stdout=subprocess.PIPE, text=True)
            time.sleep(1.0)
            samples = []
            while p.poll() is None:
                s = smi()
                if s:
                    samples.append(s)
                time.sleep(0.05)
            out = p.stdout.read().strip()
            sc = sorted(s[0] for s in samples if s[0]); pw = sorted(s[1] for s in samples if s[1])
            med = lambda x: x[len(x) // 2] if x else float('nan')
            print('%s | sclk med %.0f MHz | power med %.0f W max %.0f W | %d samples' % (out, med(sc), med(pw), pw[-1] if pw else 0, len(samples)), flush=True)


if __name__ == '__main__':
    main()
