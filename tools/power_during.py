"""Samples sclk / socket power (rocm-smi) every ~60 ms while a command runs as a CHILD process (this process never touches the GPU).

    python tools/power_during.py <skip seconds> -- python bench.py --steps 200 --no-cpu-baseline

Prints the child's stdout unchanged, then one summary line to stderr-like stdout: samples after the first <skip> seconds (start-up,
weights, warm-up), min / median / mean / max of sclk and power."""
import json
import subprocess
import sys
import threading
import time


def sample():
    try:
        out = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--json'], capture_output=True, text=True, timeout=5).stdout
        card = json.loads(out)
        card = card[sorted(card.keys())[0]]
        sclk = pw = None
        for k, v in card.items():
            kl = k.lower()
            if 'sclk' in kl and 'clock speed' in kl:
                sclk = float(str(v).strip('()').lower().replace('mhz', ''))
            if 'power' in kl and '(w)' in kl:
                pw = float(v)
        return sclk, pw
    except Exception:
        return None


def main():
    i = sys.argv.index('--')
    skip = float(sys.argv[1]) if i > 1 else 0.0
    cmd = sys.argv[i + 1:]
    samples = []
    stop = threading.Event()
    t0 = time.time()

    def loop():
        while not stop.is_set():
            s = sample()
            if s and s[0] is not None and s[1] is not None:
                samples.append((time.time() - t0,) + s)
            time.sleep(0.03)
    th = threading.Thread(target=loop, daemon=True)
    th.start()
    rc = subprocess.call(cmd)
    stop.set()
    th.join()
    end = time.time() - t0
    keep = [s for s in samples if s[0] >= skip and s[0] <= end - 0.3]
    # the busy part: samples above 60 % of the maximum power seen (drops the cpu-baseline / host phases of a bench run)
    if keep:
        pmax = max(s[2] for s in keep)
        busy = [s for s in keep if s[2] >= 0.6 * pmax]
        for name, grp in (('all', keep), ('busy (power >= 60 % of max)', busy)):
            sc = sorted(s[1] for s in grp)
            pw = sorted(s[2] for s in grp)
            print('power_during %-28s n %4d | sclk MHz min %5.0f med %5.0f mean %5.0f max %5.0f | power W min %5.0f med %5.0f mean %5.0f max %5.0f' % (
                name, len(grp), sc[0], sc[len(sc) // 2], sum(sc) / len(sc), sc[-1], pw[0], pw[len(pw) // 2], sum(pw) / len(pw), pw[-1]))
    sys.exit(rc)


if __name__ == '__main__':
    main()
