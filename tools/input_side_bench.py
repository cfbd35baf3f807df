"""Input side, measured once (GPU box): `run.py pipeline_eval_multi` from a TSV of base64 JPEGs on disk, next to the resident-HBM
bench figure.  Builds a synthetic test set (N JPEGs of mixed COCO-like sizes), then reports
  * decode-only images/s of the host (base64 + Pillow/libjpeg -> RGB) by number of threads, and the cost per image,
  * end-to-end images/s of the pipeline (TSV -> decode threads -> HIP resize/crop/normalise -> 2-slot caption pipeline -> predict TSV)
    by `num_workers`, batch size 64.
Usage: python tools/input_side_bench.py [N=4096] [out.json]"""
import base64
import io
import json
import os
import sys
import tempfile
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch
import yaml

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
SIZES = [(480, 640), (375, 500), (640, 480), (427, 640), (768, 1024), (333, 500), (500, 375), (600, 800)]


def synth_jpeg(i):
    from PIL import Image
    h, w = SIZES[i % len(SIZES)]
    g = np.random.default_rng(1000 + i)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.stack([127 + 100 * np.sin(xx / (17 + i % 13) + c) * np.cos(yy / (23 + i % 7) - c) for c in range(3)], -1)
    img += g.normal(0, 12, img.shape)                     # texture: keeps the entropy decoder honest (about 100-200 KB per image)
    buf = io.BytesIO()
    Image.fromarray(np.clip(img, 0, 255).astype(np.uint8), 'RGB').save(buf, format='JPEG', quality=90)
    return buf.getvalue()


def main():
    import run
    from vitcap_amd.imageio import decode_image
    from vitcap_amd.model import ImageCaptioning
    from vitcap_amd.tsv import TSVFile, tsv_writer
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    if os.environ.get('INPUT_SIDE_SWITCH') == '1':
        sys.setswitchinterval(0.0005)          # experiment: GIL hand-over between the launcher, the loader thread and the pool's threads
    out_path = sys.argv[2] if len(sys.argv) > 2 else None
    tmp = tempfile.mkdtemp(prefix='vitcap_input_')
    os.chdir(tmp)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(32) as pool:
        jpegs = list(pool.map(synth_jpeg, range(min(N, 256))))       # 256 distinct images, repeated: the decoder does not care
    rows = [('img%d' % i, base64.b64encode(jpegs[i % len(jpegs)])) for i in range(N)]
    tsv_writer(rows, os.path.join(tmp, 'data', 'toy', 'test.tsv'))
    quota = None
    try:      # cgroup v2 CPU bandwidth of the box: "<quota us> <period us>" (the GPU pool gives 16 cores of a 256-thread host)
        q, per = open('/sys/fs/cgroup/cpu.max').read().split()
        quota = None if q == 'max' else round(int(q) / int(per), 1)
    except Exception:
        pass
    import torch as _t
    res = {'torch_threads': _t.get_num_threads(), 'images': N, 'mean_jpeg_bytes': int(np.mean([len(j) for j in jpegs])), 'sizes': SIZES, 'host_cores': os.cpu_count(),
           'host_cpu_quota_cores': quota,
           'build_s': round(time.perf_counter() - t0, 1)}
    # ---- decode only
    t = TSVFile(os.path.join(tmp, 'data', 'toy', 'test.tsv'))
    recs = [t[i][1] for i in range(min(N, 1024))]
    dec = {}
    skip_dec = bool(os.environ.get('INPUT_SIDE_SKIP_DECODE'))       # the decode-only sweeps take a minute of box time
    for th in (() if skip_dec else (1, 8, 16, 32, 64)):
        with ThreadPoolExecutor(th) as pool:
            list(pool.map(decode_image, recs[:64]))
            t0 = time.perf_counter()
            list(pool.map(decode_image, recs))
            dec[th] = round(len(recs) / (time.perf_counter() - t0), 1)
    res['decode_only_images_per_s_by_threads'] = dec
    if dec:
        res['decode_ms_per_image_one_thread'] = round(1e3 / dec[1], 2)
    # ---- the pipeline from disk
    enc = os.path.join(tmp, 'enc')
    os.makedirs(enc)
    toks = ['[PAD]'] + ['w%d' % i for i in range(1, 30522)]
    toks[100], toks[101], toks[102], toks[103] = '[UNK]', '[CLS]', '[SEP]', '[MASK]'
    open(os.path.join(enc, 'vocab.txt'), 'w').write('\n'.join(toks) + '\n')
    sd = ImageCaptioning().load_recipe(0).state_dict()
    ck = os.path.join(tmp, 'base.pt')
    torch.save({'model': {'module.' + k: v for k, v in sd.items()}, 'iteration': 0}, ck)
    # decode only, worker processes (what the loader uses)
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    from vitcap_amd.jpegdec import decode_many
    decp = {}
    for th in (() if skip_dec else (8, 16, 32)):
        with ProcessPoolExecutor(th, mp_context=mp.get_context('spawn')) as pool:
            chunks = [recs[c:c + 8] for c in range(0, len(recs), 8)]
            list(pool.map(decode_many, chunks[:th]))
            t0 = time.perf_counter()
            list(pool.map(decode_many, chunks))
            decp[th] = round(len(recs) / (time.perf_counter() - t0), 1)
    res['decode_only_images_per_s_by_worker_processes'] = decp
    # the pipeline: run.py pipeline_eval_multi over the N rows; steady-state images/s as CaptionUniPipeline.predict measures it (from the
    # second batch's captions to the last row: model load, weight packing and worker start-up are outside)
    from vitcap_amd import pipeline as P
    e2e = {}

    def run_once(name, workers, threads, device_jpeg=True):
        cfg = {'type': 'pipeline_eval_multi', 'all_test_data': [{'test_data': 'toy', 'test_split': 'test'}],
               'param': {'full_expid': name, 'max_iter': 10, 'model_file': ck, 'text_encoder_type': enc, 'tagemb': 'cls',
                         'test_batch_size': 64, 'force_predict': True, 'crop_pct': 1.0, 'test_crop_size': 384, 'num_workers': workers,
                         'loader_threads': threads, 'device_jpeg': bool(device_jpeg), 'pipeline_type': {'from': 'vitcap_amd.pipeline', 'import': 'CaptionUniPipeline'}}}
        yf = os.path.join(tmp, name + '.yaml')
        open(yf, 'w').write(yaml.safe_dump(cfg))
        kw = run.parse_general_args(['-c', yf])
        P.LAST_PREDICT_STATS.clear()
        t0 = time.perf_counter()
        getattr(run, kw.pop('type'))(**kw)
        torch.cuda.synchronize()
        return dict(P.LAST_PREDICT_STATS), time.perf_counter() - t0
    def run_fed(name, make_batches):
        """The same predict loop (2-slot caption pipeline, detokeniser, predict TSV) fed by `make_batches(pipeline)` instead of the
        loader: what the GPU + this process sustain when JPEG decoding costs nothing."""
        param = {'full_expid': name, 'max_iter': 10, 'model_file': ck, 'text_encoder_type': enc, 'tagemb': 'cls', 'test_batch_size': 64,
                 'force_predict': True, 'crop_pct': 1.0, 'test_crop_size': 384, 'num_workers': 0, 'test_data': 'toy', 'test_split': 'test',
                 'pipeline_type': {'from': 'vitcap_amd.pipeline', 'import': 'CaptionUniPipeline'}}
        pipe = run._build(param)[0]
        pipe.cfg.test_batches = make_batches(pipe)
        P.LAST_PREDICT_STATS.clear()
        pipe.ensure_predict()
        torch.cuda.synchronize()
        return round(P.LAST_PREDICT_STATS['images_per_sec'], 1)

    if os.environ.get('INPUT_SIDE_CEILING'):
        from vitcap_amd.imageio import ImagePreprocessor
        nb = N // 64
        dev = torch.device('cuda', 0)
        decoded = [np.ascontiguousarray(decode_image(r)) for r in recs[:256]]
        pinned = [torch.from_numpy(a).pin_memory().numpy() for a in decoded]       # page-locked, as the loader's slabs are

        def resident(pipe):
            pre = ImagePreprocessor(dev, 384, 1.0)
            img = pre(pinned[:64])
            torch.cuda.synchronize()
            for b in range(nb):
                yield {'image': img, 'key': ['img%d' % (b * 64 + i) for i in range(64)]}

        def decoded_host(pipe):
            pre = ImagePreprocessor(dev, 384, 1.0)
            for b in range(nb):
                imgs = [pinned[(b * 64 + i) % 256] for i in range(64)]
                yield {'image': pre(imgs), 'key': ['img%d' % (b * 64 + i) for i in range(64)]}
        res['ceiling'] = {}
        for rep in range(2):
            res['ceiling']['resident_batches_through_predict_%d' % rep] = run_fed('res%d' % rep, resident)
            res['ceiling']['decoded_pinned_host_images_through_predict_%d' % rep] = run_fed('dech%d' % rep, decoded_host)
        print('ceiling', res['ceiling'], flush=True)
    sweep = ((8, True), (6, False), (8, False), (10, False), (12, False), (14, False), (16, False))
    if os.environ.get('INPUT_SIDE_WORKERS'):
        sweep = tuple((int(w), False) for w in os.environ['INPUT_SIDE_WORKERS'].split(','))
    # INPUT_SIDE_DEVICE_JPEG = comma list aligned with the sweep (round 6): 1 = the workers only entropy-decode, the GPU finishes the JPEG
    # (csrc/jpeg.hip; the default of the pipeline), 0 = Pillow decodes everything in the workers (rounds 1-5)
    dj = [int(x) for x in os.environ.get('INPUT_SIDE_DEVICE_JPEG', '').split(',') if x != '']
    sweep = tuple((w, t, (dj[i] if i < len(dj) else 1)) for i, (w, t) in enumerate(sweep))
    def cpu_stat():
        try:
            return {k: int(v) for k, v in (ln.split() for ln in open('/sys/fs/cgroup/cpu.stat').read().splitlines())}
        except Exception:
            return {}
    res['cgroup_cpu'] = {}
    import threading
    for run_i, (workers, threads, device_jpeg) in enumerate(sweep):
        c0 = cpu_stat()
        m0 = torch.cuda.memory_stats().get('num_device_alloc', 0)
        samples, stop = [], threading.Event()

        def sampler():        # cgroup CPU accounting every 50 ms: which part of the throttling falls into the steady-state window
            while not stop.is_set():
                c = cpu_stat()
                samples.append((time.perf_counter(), c.get('nr_throttled', 0), c.get('usage_usec', 0), c.get('nr_periods', 0)))
                stop.wait(0.05)
        th = threading.Thread(target=sampler, daemon=True)
        th.start()
        st, wall = run_once('w%d%d_%d' % (workers, threads, run_i), workers, threads, device_jpeg)
        stop.set()
        th.join()
        c1 = cpu_stat()
        if samples and st.get('t_end'):
            win = [x for x in samples if st['t_end'] - st['steady_seconds'] <= x[0] <= st['t_end']]
            if len(win) > 1:
                res.setdefault('cgroup_cpu_steady_window', {})['run %d: %d %s, device_jpeg %d' % (run_i, workers, 'threads' if threads else 'processes', device_jpeg)] = sw = {
                    'seconds': round(win[-1][0] - win[0][0], 2), 'periods': win[-1][3] - win[0][3], 'throttled_periods': win[-1][1] - win[0][1],
                    'avg_cores': round((win[-1][2] - win[0][2]) / 1e6 / max(1e-9, win[-1][0] - win[0][0]), 2)}
                print('steady window cgroup cpu', sw, flush=True)
        if c0 and c1:      # whole run (model load included): CPU seconds used by the cgroup, periods in which the quota throttled it
            res['cgroup_cpu']['run %d: %d %s, device_jpeg %d' % (run_i, workers, 'threads' if threads else 'processes', device_jpeg)] = {
                'wall_s': round(wall, 2), 'cpu_s': round((c1['usage_usec'] - c0['usage_usec']) / 1e6, 2),
                'avg_cores': round((c1['usage_usec'] - c0['usage_usec']) / 1e6 / wall, 2),
                'periods': c1.get('nr_periods', 0) - c0.get('nr_periods', 0), 'throttled_periods': c1.get('nr_throttled', 0) - c0.get('nr_throttled', 0),
                'throttled_s': round((c1.get('throttled_usec', 0) - c0.get('throttled_usec', 0)) / 1e6, 2)}
            print('cgroup cpu', res['cgroup_cpu']['run %d: %d %s, device_jpeg %d' % (run_i, workers, 'threads' if threads else 'processes', device_jpeg)], flush=True)
        print('device allocations (hipMalloc) during the run: %d; reserved %.1f GB' % (
            torch.cuda.memory_stats().get('num_device_alloc', 0) - m0, torch.cuda.memory_reserved() / 2**30), flush=True)
        e2e['run %d: %d %s, %s' % (run_i, workers, 'threads' if threads else 'processes', 'device JPEG back half' if device_jpeg else 'Pillow in the workers')] = round(st['images_per_sec'], 1)
        print('num_workers %d (%s, device_jpeg %d): %.1f images/s from disk in the steady state (%d rows in %.2f s; whole run %.1f s); loader seconds %s' % (
            workers, 'threads' if threads else 'processes', device_jpeg, st['images_per_sec'], st['steady_rows'], st['steady_seconds'], wall,
            st.get('loader_seconds')), flush=True)
        res.setdefault('loader_seconds', {})['run %d: %d %s, device_jpeg %d' % (run_i, workers, 'threads' if threads else 'processes', device_jpeg)] = st.get('loader_seconds')
    res['pipeline_from_tsv_images_per_s'] = e2e
    res['note'] = ('run.py pipeline_eval_multi, batch 64, 2-slot caption pipeline, predictions written as the reference\'s predict TSV; '
                   'steady state = from the second batch\'s captions to the last row')
    print(json.dumps(res))
    if out_path:
        json.dump(res, open(os.path.join(REPO, out_path) if not os.path.isabs(out_path) else out_path, 'w'), indent=1)


if __name__ == '__main__':
    main()
