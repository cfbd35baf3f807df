#!/bin/bash
# usage (GPU box): bash tools/pmc_traffic.sh <tag> [extra bench.py args, e.g. "--batch 512"]  -> HBM traffic per kernel (FETCH_SIZE / WRITE_SIZE in separate passes)
TAG=$1
EXTRA="$2"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/traffic_${TAG}_$c -o p -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --single-region --power 0 $EXTRA > $R/gpurun_out/traffic_${TAG}_$c.log 2>&1
done
cd $R
python3 - $TAG <<'PY'
import csv, collections, json, sys
tag = sys.argv[1]
out = {}
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    rows = list(csv.DictReader(open('gpurun_out/traffic_%s_%s/p_counter_collection.csv' % (tag, c))))
    agg = collections.defaultdict(list)
    for r in rows:
        if r['Counter_Name'] == c:
            n = r['Kernel_Name'].replace('(anonymous namespace)::', '')
            n = n.split('(')[0]
            agg[n].append(float(r['Counter_Value']))
    for n, v in agg.items():
        out.setdefault(n, {})[c] = {'launches': len(v), 'avg': sum(v) / len(v)}
res = {}
for n, d in out.items():
    f = d.get('FETCH_SIZE', {}).get('avg', 0.0)
    w = d.get('WRITE_SIZE', {}).get('avg', 0.0)
    # rocprofv3 reports KiB; gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide coalesced reads -> x2
    res[n] = {'launches': d.get('FETCH_SIZE', d.get('WRITE_SIZE'))['launches'], 'fetch_kib_raw': f, 'write_kib_raw': w,
              'hbm_bytes_per_launch_corrected': (2.0 * f + w) * 1024.0}
json.dump(res, open('gpurun_out/traffic_%s.json' % tag, 'w'), indent=1)
for n, d in sorted(res.items(), key=lambda x: -x[1]['hbm_bytes_per_launch_corrected'] * x[1]['launches'])[:12]:
    print('%-60s launches %4d  fetch %.1f MiB (raw)  write %.1f MiB  corrected %.1f MB' % (n[:60], d['launches'], d['fetch_kib_raw'] / 1024, d['write_kib_raw'] / 1024, d['hbm_bytes_per_launch_corrected'] / 1e6))
PY
