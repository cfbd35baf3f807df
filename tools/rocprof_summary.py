"""Turn a rocprofv3 `--kernel-trace --stats` result (rocpd sqlite .db) into a small markdown summary
for profiles/.  Usage: python tools/rocprof_summary.py gpurun_out/prof/r1_results.db "title" > profiles/x.md"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    title = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
    rows = list(db.execute('select name, total_calls, total_duration, average, percentage from top_kernels'))
    tot = sum(r[2] for r in rows)
    print('# %s\n' % title)
    print('Source: `rocprofv3 --kernel-trace --stats` (durations in microseconds; total kernel time %.1f us)\n' % tot)
    print('| kernel | calls | total us | avg us | % |')
    print('|---|---:|---:|---:|---:|')
    for n, c, t, a, p in rows:
        n = n.replace('(anonymous namespace)::', '').replace('|', '/')
        if len(n) > 110:
            n = n[:107] + '...'
        print('| `%s` | %d | %.1f | %.2f | %.2f |' % (n, c, t, a, p))


if __name__ == '__main__':
    main()
