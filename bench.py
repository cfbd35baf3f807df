"""bench.py -- images/sec, end-to-end greedy 20-token caption, ViT-B/16-384 (BASELINE.json metric).

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W
  python bench.py --gpus N ...      (no WORLD_SIZE in the environment: the process starts that torch.distributed.run
                                     command itself as a CHILD, before it has touched a GPU, and exits with its code)

One "step" = one pass of the captioning hot path (patch embed -> 16 ViT blocks -> tag head -> decoder
prefill -> 19 incremental decode steps with device-side greedy bookkeeping) over one batch of 64 synthetic
384x384 images per GPU, already resident in HBM as bf16 (BASELINE.json configs[1]).  Inference shards by
image with no data-path collective ("replicas only"): every rank runs its own batch;
torch.distributed (RCCL) is used only for the barriers and the max-over-ranks of the elapsed time.

Prints ONE JSON line (rank 0).  Extra objects:
  roofline     -- for the dominant kernel (the 256x256x64 bf16 MFMA GEMM): algorithmic FLOPs of its launches
                  / their summed duration, measured live with hipEvents recorded on the launch stream around
                  every such launch inside the timed region (vitcap_engine_timing_*).
  cpu_baseline -- the CPU oracle's restatement of the reference algorithm AS WRITTEN (full re-encode per
                  step, fp32 eager torch) timed on this host: a bounded sample of 1 image.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

TRAFFIC_FILE = os.path.join(REPO, 'profiles', 'r06_hbm_traffic_pmc.json')   # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/pmc_traffic.sh)
FLOP_PER_IMAGE = 187.95e9          # SURVEY.md section 8d / BASELINE.md section 3 (algorithmic)
# What the engine executes per greedy image: of the 4th tag block only the CLS row is ever read (pooler input and first
# visual token), so its Q / attention / proj / MLP run for that row alone: 9.19 GF -> K|V projections 1.36 + one 128-row
# attention block 0.23.  Reported next to the algorithmic figure; `value` (images/s) does not depend on either.
FLOP_EXECUTED_PER_IMAGE = FLOP_PER_IMAGE - (9.19e9 - 1.36e9 - 0.23e9)
PEAK_BF16_TFLOPS = 2500.0          # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
# kernel names in the PMC file: <activation, fp32 output, residual, schedule, short-tile height class (0 = 256-row tiles only)>
PMC_KERNEL = {3: 'void gemm_nt_256_kernel<0, 1, true, 4, 0>', 0: 'void gemm_nt_256_kernel<0, 0, false, 4, 0>',
              4: 'void gemm_nt_256_kernel<1, 0, false, 4, 0>'}
# the 4-wave persistent kernels of the same variants (one-stream runs, and the pipeline from 8 rounds of tiles on): <activation, fp32
# output, residual, m-tiles of 16 rows per wave> -- the B = 64 file holds the 224-row forms, the B = 512 file the 256-row ones
PMC_KERNEL_4W = {3: 'void gemm_nt_4wp_kernel<0, 1, true, ', 0: 'void gemm_nt_4wp_kernel<0, 0, false, ', 4: 'void gemm_nt_4wp_kernel<1, 0, false, '}      # name prefixes
VARIANT_NAMES = {0: 'gemm_nt_256<256x256x64,bias,bf16>', 1: 'gemm_nt_256<256x256x64,bias+res,bf16>',
                 2: 'gemm_nt_256<256x256x64,bias,f32>', 3: 'gemm_nt_256<256x256x64,bias+residual,f32>',
                 4: 'gemm_nt_256<256x256x64,bias+gelu,bf16>', 6: 'gemm_nt_256<256x256x64,bias+gelu,f32>'}


def kernel_form(lib, M, variant, tiles_mode):
    """Which kernel family ran the dominant large-GEMM variant (vitcap_gemm_large_form: the dispatch rule of vitcap_gemm_ex)."""
    N, K = {0: (2304, 768), 4: (3072, 768), 3: (768, 768)}.get(variant, (768, 768))
    # variants 0 / 4 (bias -> bf16, bias + GELU -> bf16) have neither fp32 output nor residual: flag 0x100 of the query
    code = lib.vitcap_gemm_large_form(M, N, K, (5 if tiles_mode else 0) | (0x100 if variant in (0, 4) else 0))
    if code < 0:
        return '8-wave 256x256x64 tiles, two waves per SIMD, one tile per workgroup (csrc/gemm.hip)'
    return '4-wave %dx256x64 tiles, one wave per SIMD / 512 registers, %s (csrc/gemm4w.hip)' % (
        32 * (code // 10), {1: 'one tile per workgroup', 2: 'persistent continuous pipeline', 0: 'one tile per workgroup, LDS epilogue'}[code % 10])


def cpu_baseline(budget_s=60.0):
    """Reference algorithm as written (a full re-encode of the ViT + joint sequence at every decode step, fp32 eager torch) on
    the host cores, by BASELINE.md section 4's protocol: ONE image (BASELINE configs[0]), the whole greedy 20-token caption (all 19
    decode steps -- random weights never emit [SEP]), 1 warm-up caption + 3 timed captions, median; no extrapolation.

    The thread count is chosen first on ONE decode step per candidate (8 = the figure comparable with BASELINE.md section 2, the
    cgroup CPU budget of this process, 32, all host cores; a count that is > 3x slower on a three-block proxy is not even probed:
    256 threads run this eager workload ~200x slower than 32 on the pool's hosts).  `budget_s` bounds the leg: if the warm-up caption
    shows that three timed captions would not fit, fewer are timed and `timed_captions` says how many."""
    from oracle import vitcap_oracle as O       # checker / baseline only
    from vitcap_amd import weights as W
    from vitcap_amd.dist_util import host_cpu_budget
    sd = O.to_torch(W.make_state_dict(0, True))
    img = torch.from_numpy(W.gen_image_batch(1, 1234))
    all_cores = os.cpu_count() or torch.get_num_threads()
    counts = []
    for n in (8, host_cpu_budget(), 32, all_cores):
        n = max(1, min(n, all_cores))
        if n not in counts:
            counts.append(n)
    default_threads = torch.get_num_threads()
    t_all = time.time()
    step_s, proxy, skipped = {}, {}, {}
    with torch.no_grad():
        for n in counts:
            torch.set_num_threads(n)
            xb = torch.randn(1, 577, 768)
            t0 = time.time()
            for i in range(3):
                xb = O.vit_block(sd, 'module.bert.encoder.blocks.%d' % i, xb)
            proxy[n] = time.time() - t0
            if step_s and proxy[n] > 3.0 * min(proxy[m] for m in step_s):
                skipped[n] = proxy[n] / min(proxy[m] for m in step_s)
                continue
            O.greedy_as_written(sd, img, max_steps=1)           # untimed: first touch of the weights at this thread count
            t0 = time.time()
            O.greedy_as_written(sd, img, max_steps=1)
            step_s[n] = time.time() - t0
        best_n = min(step_s, key=lambda n: step_s[n])
        torch.set_num_threads(best_n)
        t0 = time.time()
        ids_w, _ = O.greedy_as_written(sd, img)                 # warm-up: one whole caption
        t_warm = time.time() - t0
        runs = []
        for i in range(3):
            if runs and (time.time() - t_all) + t_warm > budget_s:
                break
            t0 = time.time()
            ids_c, _ = O.greedy_as_written(sd, img)
            runs.append(time.time() - t0)
            assert torch.equal(torch.as_tensor(ids_c), torch.as_tensor(ids_w))
    torch.set_num_threads(default_threads)
    runs.sort()
    s_per_image = runs[len(runs) // 2] if len(runs) % 2 else 0.5 * (runs[len(runs) // 2 - 1] + runs[len(runs) // 2])
    return {'value': 1.0 / s_per_image, 'unit': 'images/sec', 'cores': best_n, 'kind': 'port',
            'sample': '1 image (BASELINE configs[0]): the whole greedy 20-token caption = 19 decode steps of the reference algorithm as '
                      'written (ViT + joint sequence re-run at every step, fp32 eager torch); 1 warm-up caption + %d timed, median '
                      '%.2f s/image at %d threads' % (len(runs), s_per_image, best_n),
            'protocol': 'BASELINE.md section 4: thread count chosen on one decode step per candidate (8, cgroup CPU budget, 32, all cores; '
                        'a count > 3x slower on a 3-block proxy is skipped), then 1 warm-up + 3 timed whole captions at that count, median',
            'host_cores': all_cores, 'cpu_budget_cores': host_cpu_budget(), 'timed_captions': len(runs),
            's_per_image_runs': [round(t, 3) for t in runs], 'warmup_caption_s': round(t_warm, 3),
            'one_step_s_by_threads': {str(n): round(t, 3) for n, t in step_s.items()},
            'not_probed_threads_proxy_slowdown': {str(n): round(r, 1) for n, r in skipped.items()},
            'leg_seconds': round(time.time() - t_all, 1)}


class PowerSampler(object):
    """Board power / shader clock read by rocm-smi in a side thread (a child process per reading: this thread never touches the GPU).
    The pipeline runs at the board's power cap, so joules per step is the number a kernel change has to move."""

    def __init__(self, period=0.02):
        import threading
        self.samples, self._stop, self.period = [], threading.Event(), period
        self._t = threading.Thread(target=self._loop, daemon=True)

    @staticmethod
    def read():
        import subprocess
        try:
            out = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--json'], capture_output=True, text=True, timeout=5).stdout
            card = json.loads(out)
            card = card[sorted(card.keys())[0]]
            sclk = pw = None
            for k, v in card.items():
                kl = k.lower()
                if 'sclk' in kl and 'clock' in kl and sclk is None:
                    sclk = float(str(v).strip('()').lower().replace('mhz', ''))
                if 'power' in kl and '(w)' in kl and pw is None:
                    pw = float(v)
            return (sclk, pw) if (sclk is not None and pw is not None) else None
        except Exception:       # noqa  (no rocm-smi, no permission: the fields stay null)
            return None

    def _loop(self):
        while not self._stop.is_set():
            r = self.read()
            if r:
                self.samples.append(r)
            time.sleep(self.period)

    def __enter__(self):
        self._t.start()
        return self

    def __exit__(self, *a):
        self._stop.set()
        self._t.join()

    def summary(self, ms_per_step):
        if len(self.samples) < 3:
            return None
        sc = sorted(s[0] for s in self.samples)
        pw = sorted(s[1] for s in self.samples)
        med_p = pw[len(pw) // 2]
        return {'board_power_w_median': med_p, 'board_power_w_max': pw[-1], 'sclk_mhz_median': sc[len(sc) // 2],
                'joules_per_step': round(med_p * ms_per_step * 1e-3, 3), 'energy_pass_ms_per_step': round(ms_per_step, 3),
                'samples': len(self.samples),
                'note': 'rocm-smi readings during an UNTIMED pass of the same steps (>= 2 s) right after the timed regions; joules_per_step = '
                        'median board power x that pass\'s time per step'}


TRAIN_FLOP_PER_SAMPLE = 569.3e9     # fwd+bwd, SURVEY.md section 8d (algorithmic)
# executed: rows nobody reads are not computed -- rows 1..576 of the last tag block (7.6 GF forward) and the 578 visual rows
# of the last decoder layer's attention / output / MLP (6.9 GF forward), forward + backward = 3x
TRAIN_FLOP_EXECUTED_PER_SAMPLE = TRAIN_FLOP_PER_SAMPLE - 3 * (7.6e9 + 6.9e9)


def bench_train(args, rank, world, local, dist, D):
    """Cross-entropy training step, data parallel: forward + backward + RCCL gradient all-reduce + clip + AdamW.
    One step = one optimizer step on `batch` samples per GPU (weak scaling: global batch = batch x N)."""
    from vitcap_amd.synthetic import synthetic_train_inputs
    from vitcap_amd import weights as W
    from vitcap_amd.model import ImageCaptioning
    from vitcap_amd.train import TrainEngine
    B = args.batch
    model = ImageCaptioning().load_recipe(0)
    eng = TrainEngine(model, 'cuda:%d' % local, max_iter=10 ** 6, dist=dist)
    eng.use_graphs = bool(args.train_graph)          # the step as ~11 host calls (captured segments) instead of ~750
    batch = synthetic_train_inputs(B, seed=D.shard_seed(4321, rank))
    batch = {k: v.cuda() for k, v in batch.items()}
    batch['image'] = torch.from_numpy(W.gen_image_batch(B, D.shard_seed(1234, rank))).cuda().to(torch.bfloat16).contiguous()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    from vitcap_amd import ops
    for _ in range(max(args.warmup, 1)):
        out = eng.train_step(batch)
    barrier()
    ops.TIMING = None if eng.use_graphs else []      # events on the launch stream around every large GEMM of the timed region (eager only)
    t0 = time.perf_counter()
    host_s = 0.0
    for _ in range(args.steps):
        h0 = time.perf_counter()
        out = eng.train_step(batch)
        host_s += time.perf_counter() - h0          # time the HOST spends issuing a step (enqueue only: no synchronisation inside)
    barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, dist, device='cuda')
    timing, ops.TIMING = (ops.TIMING or []), None
    roof_note = None
    if eng.use_graphs:
        # launches replayed from a graph carry no per-launch events: the roofline of the dominant GEMM comes from an UNTIMED eager pass
        # of the same step right behind the timed region (same kernels, arguments and order as the captured ones)
        eng.use_graphs = False
        ops.TIMING = []
        for _ in range(min(args.steps, 5)):
            eng.train_step(batch)
        torch.cuda.synchronize()
        timing, ops.TIMING = ops.TIMING, None
        eng.use_graphs = True
        roof_note = 'per-launch events of an untimed eager pass (5 steps) behind the timed region: graph replays carry none'
    n_timed_steps = min(args.steps, 5) if roof_note else args.steps
    kinds = {}
    for kind, fl, e0, e1 in timing:
        k = kinds.setdefault(kind, [0.0, 0.0, 0])
        k[0] += e0.elapsed_time(e1)
        k[1] += fl
        k[2] += 1
    if rank == 0:
        value = D.whole_job_rate(B, args.steps, world, elapsed)
        dom = max(kinds, key=lambda k: kinds[k][0]) if kinds else None
        roof = None
        if dom:
            ms_, fl_, n_ = kinds[dom]
            roof = {'bound': 'mfma', 'kernel': dom, 'achieved': round(fl_ / (ms_ * 1e-3) / 1e12, 2), 'peak': PEAK_BF16_TFLOPS,
                    'unit': 'TFLOP/s', 'frac': round(fl_ / (ms_ * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4), 'traffic': None,
                    'launches': n_, 'avg_launch_ms': round(ms_ / n_, 4), 'avg_launch_gflop': round(fl_ / n_ / 1e9, 3),
                    'share_of_step_time': round(ms_ / n_timed_steps / (elapsed / args.steps * 1e3), 4), 'note': roof_note,
                    'per_kind': {k: {'launches': v[2], 'ms': round(v[0], 3), 'tflops': round(v[1] / (v[0] * 1e-3) / 1e12, 2)}
                                 for k, v in kinds.items()}}
        emit({
            'metric': 'images/sec cross-entropy training step, ViT-B/16-384 + 4-layer caption decoder',
            'value': round(value, 2), 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[3]: cross-entropy training step, %d samples per GPU (global %d), '
                                   'decoder attention dropout 0.1 on, fp32 master weights + AdamW, gradient exchange over RCCL (%s)'
                                   % (B, B * world, eng.reducer.algo),
                       'batch_per_gpu': B, 'global_batch': B * world, 'parallelism': 'dp%d' % world},
            'end_to_end_tflops_algorithmic': round(value / world * TRAIN_FLOP_PER_SAMPLE / 1e12, 2),
            'end_to_end_tflops_executed': round(value / world * TRAIN_FLOP_EXECUTED_PER_SAMPLE / 1e12, 2),
            'end_to_end_frac_of_bf16_peak': round(value / world * TRAIN_FLOP_PER_SAMPLE / 1e12 / PEAK_BF16_TFLOPS, 4),
            'roofline': roof,
            'launch': ('hipGraph segments replayed (TrainEngine.train_step_graph): copies + salt + %d replay(s) + optimizer' % len(next(iter(eng._graphs.values()))['segs'])
                       if eng.use_graphs else 'eager: every kernel launched from Python (ctypes)'),
            'host_issue_ms_per_step': round(host_s / args.steps * 1e3, 3),
            'masked_loss': float(out['masked_loss']), 'tag_loss': float(out['tag_loss'])})
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def bench_scst(args, rank, world, local, dist, D):
    """BASELINE configs[4]: SCST step, scst_num_return=5, 16 images per GPU (global 128 on 8 GPUs): greedy baseline decode,
    5 sampled decodes per image, CIDEr-D advantage on the host (synthetic reference captions), one-pass gradient, AdamW."""
    from vitcap_amd import weights as W
    from vitcap_amd.model import ImageCaptioning
    from vitcap_amd.scst import ScstTrainer
    from vitcap_amd.tokenizer import CaptionDetokenizer
    from vitcap_amd.train import TrainEngine
    B = 16 if args.batch == 64 else args.batch
    K = 5
    toks = ['[PAD]'] + ['w%d' % i for i in range(1, 30522)]
    toks[100], toks[101], toks[102], toks[103] = '[UNK]', '[CLS]', '[SEP]', '[MASK]'
    tok = CaptionDetokenizer(tokens=toks)
    model = ImageCaptioning().load_recipe(0)
    eng = TrainEngine(model, 'cuda:%d' % local, max_iter=10 ** 6, dist=dist)
    img = torch.from_numpy(W.gen_image_batch(B, D.shard_seed(999, rank))).cuda().to(torch.bfloat16).contiguous()
    model.eval()
    g_ids, _ = model.generate(img)
    gts = [[tok.decode(r.tolist(), skip_special_tokens=True), 'w11 w12 w13 w14 w15'] for r in g_ids[:, 0].cpu()]
    tr = ScstTrainer(model, eng, tok, num_return=K, seed=rank)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(max(args.warmup, 1)):
        out = tr.step(img, gts)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = tr.step(img, gts)
    barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, dist, device='cuda')
    if rank == 0:
        value = D.whole_job_rate(B, args.steps, world, elapsed)
        emit({
            'metric': 'images/sec SCST step (greedy baseline + 5 sampled captions/image + policy gradient), ViT-B/16-384',
            'value': round(value, 2), 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[4]: SCST, scst_num_return=5, %d images per GPU (global %d), synthetic '
                                   'reference captions, CIDEr-D advantage on the host' % (B, B * world),
                       'batch_per_gpu': B, 'global_batch': B * world, 'sampled_sequences_per_step': B * K * world,
                       'parallelism': 'dp%d' % world},
            'scst_loss': float(out['scst_loss']), 'score': out['score']})
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def bench_stub(args, rank, world, D):
    """The bench's launch / timing protocol on a CPU stand-in (gloo): warm-up, barrier, K timed steps, barrier, MAX over ranks,
    one JSON line from rank 0.  Exists so that the N > 1 entry (self-launch included) is covered by a test without a GPU; the
    line says so and carries no roofline."""
    dist = D.init('gloo') if world > 1 else None
    x = torch.ones(256, 256)

    def step():
        return float((x @ x).sum())
    for _ in range(max(args.warmup, 1)):
        step()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if dist is not None:
        dist.barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t0 + 1e-3 * rank, dist)
    if rank == 0:
        emit({'metric': 'stub steps/sec (plumbing test, not a measurement)', 'value': round(D.whole_job_rate(1, args.steps, world, elapsed), 2),
                          'unit': 'steps/sec', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                          'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
                          'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic', 'config': {'workload': 'stub (CPU, gloo)'}})
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


_JSON_FD = None


def emit(obj):
    """The ONE JSON line, on the process's ORIGINAL stdout (see main(): fd 1 itself is pointed at stderr)."""
    line = (json.dumps(obj) + '\n').encode()
    if _JSON_FD is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_JSON_FD, line)


def main():
    # RCCL prints a version banner to STDOUT when its first communicator is created (seen under torch.distributed.run on the GPU
    # box: five lines ahead of the JSON).  The contract is ONE JSON line on stdout: keep the original stdout for that line only
    # and point fd 1 (C stdio of every library included) at stderr for everything else.
    global _JSON_FD
    sys.stdout.flush()
    _JSON_FD = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200, help='timed steps (200 x ~18 ms keeps the timed region above 3 s)')
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=64, help='images per GPU per step (configs[1]: 64)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--power', type=int, default=1, help='1: an untimed pass of >= 2 s after the timed regions with rocm-smi sampled beside it (extra.board_power_w_median, sclk_mhz_median, joules_per_step); 0 skips it')
    ap.add_argument('--single-region', action='store_true', help='time ONE region of K steps even when it is shorter than 2 s (profiling runs)')
    ap.add_argument('--mode', default='caption', choices=['caption', 'train', 'scst'],
                    help="'caption' = the headline metric; 'train' = cross-entropy training step (BASELINE configs[3])")
    ap.add_argument('--beams', type=int, default=1, help='num_beams (1 = greedy, the headline metric; 5 = BASELINE configs[2])')
    ap.add_argument('--graph', type=int, default=0,
                    help='1: the decode loop is captured once into a hipGraph inside the engine (vitcap_gen_opts.use_graph) and replayed')
    ap.add_argument('--gemm-tiles', type=int, default=0, help='one-stream runs: large GEMMs one tile per workgroup (+ forked tag branch)')
    ap.add_argument('--isolated', type=int, default=1, help='1: after the timed region, an untimed one-stream pass measures the dominant kernel with the chip to itself (roofline.isolated); 0 skips it (profiling runs: keeps the rocprofv3 per-kernel averages those of the timed configuration)')
    ap.add_argument('--pipeline', type=int, default=1,
                    help='1: two-slot batch pipeline (encode+prefill of step i+1 overlaps the decode of step i on a second '
                         'stream; same results); 0: one stream, steps strictly back to back')
    ap.add_argument('--train-graph', type=int, default=1,
                    help='--mode train: 1 = the step replays hipGraph segments captured once per batch shape, 0 = eager launches from Python')
    ap.add_argument('--stub', action='store_true',
                    help='CPU stand-in for the workload (tests of the launch / barrier / max-over-ranks / JSON plumbing only: gloo, no '
                         'GPU, no kernels; never a measurement)')
    args = ap.parse_args()

    from vitcap_amd import dist_util as D
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # Started as plain `python bench.py --gpus N`: run one rank per GPU under torch.distributed.run as a CHILD process and
        # hand back its exit code.  Nothing in this process has initialised the GPU yet (importing torch does not), and it never
        # will: replacing a process that has touched the GPU (exec) takes the node down on this pool, a child does not.
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(('127.0.0.1', 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        sys.exit(subprocess.call(cmd, env=env, stdout=_JSON_FD))    # the ranks' stdout is this process's ORIGINAL stdout
    rank, world, local = D.env_rank_world()
    assert world == args.gpus, '--gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node == --gpus' % (args.gpus, world)
    if args.stub:
        return bench_stub(args, rank, world, D)
    torch.cuda.set_device(local)
    # a process group of ONE rank is legal: launched by torch.distributed.run with VITCAP_DP_FORCE=1 the launcher, init(device_id),
    # barrier, max-over-ranks and (training) the bucketed exchange all run on RCCL on a single-GPU box (scripts_gpu_round.sh)
    one_rank_group = world == 1 and os.environ.get('VITCAP_DP_FORCE', '0') == '1' and 'MASTER_ADDR' in os.environ
    dist = D.init('nccl', torch.device('cuda', local)) if (world > 1 or one_rank_group) else None

    from vitcap_amd import weights as W
    from vitcap_amd._lib import lib, check
    from vitcap_amd.model import ImageCaptioning

    B = args.batch
    if args.mode == 'train':
        return bench_train(args, rank, world, local, dist, D)
    if args.mode == 'scst':
        return bench_scst(args, rank, world, local, dist, D)
    model = ImageCaptioning().load_recipe(0).eval()
    model.pack('cuda:%d' % local)
    img = torch.from_numpy(W.gen_image_batch(B, D.shard_seed(1234, rank))).cuda().to(torch.bfloat16).contiguous()
    from vitcap_amd import _lib as L
    gen_kw = dict(num_beams=args.beams, num_keep_best=1, do_sample=False, num_return_sequences=1, use_graph=bool(args.graph))
    opts = model.gen_options(gemm_mode=L.GEMM_TILES if args.gemm_tiles else L.GEMM_AUTO, **gen_kw)   # one stream
    popts = model.gen_options(gemm_mode=L.GEMM_TILES, **gen_kw)      # batch pipeline: one tile per workgroup

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    stream = torch.cuda.Stream()
    ids = lp = None
    piped = bool(args.pipeline)
    with torch.cuda.stream(stream):
        if piped:
            model.prime_pipeline(img.shape[0], img.device, opts=popts)   # both slots' workspaces exist before any step runs
        for _ in range(max(args.warmup, 1)):
            ids, lp = model.generate_async(img, opts=popts).result() if piped else model.run(img, opts)
        stream.synchronize()

        # ---- timed region: exactly K steps, bracketed by barrier + synchronize on both sides ---------------
        TIMING_STRIDE = 7       # the large-GEMM launches of every 7th step carry events (whole steps: the busy-interval union stays meaningful)
        check(lib.vitcap_engine_timing_sample(model._engine, TIMING_STRIDE), 'timing_sample')

        def region(armed):
            """One timed region of exactly args.steps steps -> wall seconds.  armed: the engine records two HIP events around
            every large-GEMM launch inside it (the roofline's per-launch durations)."""
            if armed:
                check(lib.vitcap_engine_timing_begin(model._engine, args.steps * 160), 'timing_begin')
            barrier()
            t0 = time.perf_counter()
            pend = None
            out = (None, None)
            for _ in range(args.steps):
                if piped:
                    pend = model.generate_async(img, opts=popts)
                else:
                    out = model.run(img, opts)
            if pend is not None:
                out = pend.result()            # the last batch; earlier ones completed before it (in-order streams)
            stream.synchronize()
            barrier()
            return time.perf_counter() - t0, out

        # A region shorter than 2 s (the driver's K = 20 at B = 64 is 0.35 s) is repeated: three armed regions, the MEDIAN is
        # the reported one, the spread goes into `timed_regions_ms_per_step`; one more region with the per-launch events NOT armed
        # prices the instrumentation (`unarmed_ms_per_step`).  `steps` stays what was passed: every region is exactly K steps.
        elapsed, (ids, lp) = region(True)
        regions = [elapsed]
        unarmed = None
        events_region = 0
        # every rank must take the same branch (each region holds barriers): the decision uses the MAX over ranks of the first region
        first = D.max_over_ranks(elapsed, dist, device='cuda')
        if first < 2.0 and not args.single_region:
            ms_keep = None
            for _ in range(2):
                check(lib.vitcap_engine_timing_end_kernel(model._engine, (C.c_double * 12)(), (C.c_double * 12)(), (C.c_int * 12)(), None,
                                                          (C.c_double * 12)(), None), 'timing_end')
                e2, (ids, lp) = region(True)
                regions.append(e2)
            events_region = len(regions) - 1
            # median by the max-over-ranks times, so that every rank reports the SAME region
            rmax = [D.max_over_ranks(r, dist, device='cuda') for r in regions]
            elapsed = regions[sorted(range(len(regions)), key=lambda i: rmax[i])[1]]
            # the events of the LAST armed region stay in the engine for the roofline below; one unarmed region first would lose
            # them, so the unarmed one runs after they have been read (see below)
            unarmed = 'pending'

        # ---- live per-launch timing of the dominant kernel (hipEvents on the launch stream around every large GEMM)
        ms = (C.c_double * 12)()
        fl = (C.c_double * 12)()
        ln = (C.c_int * 12)()
        busy = (C.c_double * 12)()
        kms = (C.c_double * 12)()
        kbusy = (C.c_double * 12)()
        check(lib.vitcap_engine_timing_end_kernel(model._engine, ms, fl, ln, busy, kms, kbusy), 'timing_end')
        if unarmed == 'pending':
            unarmed, _ = region(False)
        # ---- energy pass (rank 0 samples; every rank runs the steps): >= 2 s of the same steps, untimed for `value`
        energy = None
        # (never under a profiler: its preloaded library would be inherited by the rocm-smi children of the sampler thread)
        profiled = any('rocprof' in os.environ.get(k, '').lower() for k in ('LD_PRELOAD', 'ROCP_TOOL_LIBRARIES', 'ROCPROFILER_LIBRARY_PATH'))
        if args.power and not profiled:
            n_e = max(args.steps, int(2.0 / max(elapsed / args.steps, 1e-4)) + 1)
            n_e = int(D.max_over_ranks(n_e, dist, device='cuda'))
            save_steps, args.steps = args.steps, n_e
            try:
                if rank == 0:
                    with PowerSampler() as ps:
                        e_s, _ = region(False)
                    energy = ps.summary(e_s / n_e * 1e3)
                else:
                    region(False)
            finally:
                args.steps = save_steps
        # With the batch pipeline the GEMMs of the timed region share the chip with the other slot's decode kernels, so
        # their launch durations are longer than the kernel alone needs.  A second, untimed pass of (at most 20 of) the same
        # steps on ONE stream gives the kernel's own rate (reported next to, not instead of, the timed-region figure).
        iso = None
        if piped and args.isolated:
            n_iso = min(args.steps, 20)
            ms2, fl2, ln2, km2 = (C.c_double * 12)(), (C.c_double * 12)(), (C.c_int * 12)(), (C.c_double * 12)()
            check(lib.vitcap_engine_timing_begin(model._engine, n_iso * 160), 'timing_begin')
            check(lib.vitcap_engine_timing_sample(model._engine, 1), 'timing_sample')
            iso_opts = model.gen_options(gemm_mode=L.GEMM_TILES if args.gemm_tiles else L.GEMM_AUTO, encode_parts=1, **gen_kw)
            for _ in range(n_iso):
                model.run(img, iso_opts)            # ONE chain: no other kernel shares the chip with the GEMM launches
            stream.synchronize()
            check(lib.vitcap_engine_timing_end_kernel(model._engine, ms2, fl2, ln2, None, km2, None), 'timing_end')
            iso = (list(km2), list(fl2), list(ln2))
        # decode phase alone (one stream, after an encode + prefill): GPU time of the step loop per batch
        dec_ms = None
        if args.beams >= 1:
            ws, need = model._workspace(B, img.device, 0, opts)
            o_ids, o_lp = model._out_buffers(B, opts, img.device)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            sp = C.c_void_p(stream.cuda_stream)
            n_dec = 5
            for i in range(n_dec + 1):
                if i == 1:
                    e0.record(stream)
                check(lib.vitcap_engine_decode(model._engine, B, C.byref(opts), C.c_void_p(ws.data_ptr()), need,
                                               C.c_void_p(o_ids.data_ptr()), C.c_void_p(o_lp.data_ptr()), None, sp), 'decode')
            e1.record(stream)
            stream.synchronize()
            dec_ms = e0.elapsed_time(e1) / n_dec

    elapsed = D.max_over_ranks(elapsed, dist, device='cuda')
    regions = [D.max_over_ranks(r, dist, device='cuda') for r in regions]
    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    value = D.whole_job_rate(B, args.steps, world, elapsed)
    # kms: durations from HIP events bound to the kernel dispatches (kernel begins executing -> complete: what rocprofv3
    # --kernel-trace reports); ms: stream-marker brackets around the launches (also hold the dispatch's wait behind the other stream)
    tot_ms = sum(kms)
    tot_fl = sum(fl)
    dom = max(range(12), key=lambda i: kms[i])
    gemm_all = (tot_fl / (tot_ms * 1e-3)) / 1e12 if tot_ms > 0 else 0.0
    dom_tf = (fl[dom] / (kms[dom] * 1e-3)) / 1e12 if kms[dom] > 0 else 0.0
    mark_tf = (fl[dom] / (ms[dom] * 1e-3)) / 1e12 if ms[dom] > 0 else 0.0
    traffic = None
    try:      # HBM bytes per launch of the dominant kernel from the committed PMC passes (tools/pmc_traffic.sh);
        # FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B), KiB -> bytes
        four_wave = kernel_form(lib, B * 577 // (2 if (piped and B >= 32) else 1), dom, piped or args.gemm_tiles).startswith('4-wave')
        tfile = TRAFFIC_FILE if B < 256 else TRAFFIC_FILE.replace('.json', '_b512.json')
        with open(tfile) as f:
            tab = json.load(f)
        names = [n for n in tab if n.startswith(PMC_KERNEL_4W[dom])] if four_wave else [PMC_KERNEL[dom]]
        traffic = max((tab[n] for n in names if n in tab), key=lambda e: e['launches'])['hbm_bytes_per_launch_corrected']
        if B not in (64, 512):
            traffic = None          # the committed passes are B = 64 and B = 512 runs
    except Exception:
        traffic = None
    out = {
        'metric': 'images/sec end-to-end greedy caption (20 tok), ViT-B/16-384' if args.beams == 1 else
                  'images/sec end-to-end beam=%d caption (20 tok), ViT-B/16-384' % args.beams,
        'value': round(value, 2), 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': round(elapsed / args.steps * 1e3, 3),
        'timed_regions_ms_per_step': [round(r / args.steps * 1e3, 3) for r in regions],
        'timed_regions_note': ('median of %d regions of exactly K = %d steps each (a single region is shorter than 2 s); value / ms_per_step are the median region\'s' % (len(regions), args.steps)) if len(regions) > 1 else 'one region of exactly K steps',
        'unarmed_ms_per_step': None if not isinstance(unarmed, float) else round(unarmed / args.steps * 1e3, 3),
        'unarmed_note': 'one more region of K steps WITHOUT the two HIP events the engine records around every large-GEMM launch for the roofline: the difference to ms_per_step is the cost of the instrumentation inside the timed region',
        'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16',
        'data': 'synthetic',
        'config': {'workload': ('BASELINE configs[1]: ViT-B/16-384 greedy decode (20 tok), batch %d bf16 per GPU, ' % B if args.beams == 1
                                else 'BASELINE configs[2]: ViT-B/16-384 beam=%d decode (20 tok), batch %d bf16 per GPU, ' % (args.beams, B)) +
                               'seeded random-init weights, uniform(-1,1) 384x384 images resident in HBM',
                   'batch_per_gpu': B, 'global_batch': B * world, 'decode': 'greedy' if args.beams == 1 else 'beam%d' % args.beams, 'max_length': 20,
                   'parallelism': 'replicas x%d (no data-path collective)' % world,
                   'launch': ('2-slot batch pipeline (encode of step i+1 || decode of step i)' if piped else 'one stream, steps back to back') +
                             (', decode loop replayed from an engine-owned hipGraph' if args.graph else ', eager launches'),
                   'streams_per_gpu': 2 if piped else 1},
        'decode_phase_ms_per_batch': None if dec_ms is None else round(dec_ms, 3),
        'extra': energy,
        'end_to_end_tflops_algorithmic': round(value / world * FLOP_PER_IMAGE / 1e12, 2),
        'end_to_end_tflops_executed': round(value / world * (FLOP_EXECUTED_PER_IMAGE if args.beams == 1 else FLOP_EXECUTED_PER_IMAGE + 13.35e9) / 1e12, 2),
        'end_to_end_frac_of_bf16_peak': round(value / world * FLOP_PER_IMAGE / 1e12 / PEAK_BF16_TFLOPS, 4),
        'roofline': {
            'bound': 'mfma', 'kernel': VARIANT_NAMES.get(dom, 'gemm_nt variant %d' % dom),
            'kernel_form': kernel_form(lib, B * 577 // (2 if (piped and B >= 32) else 1), dom, piped or args.gemm_tiles),
            'encode_parts': int(os.environ['VITCAP_ENCODE_SPLIT']) if os.environ.get('VITCAP_ENCODE_SPLIT') else (2 if (piped and B >= 32) else 1),
            'achieved': round(dom_tf, 2), 'peak': PEAK_BF16_TFLOPS, 'unit': 'TFLOP/s',
            'frac': round(dom_tf / PEAK_BF16_TFLOPS, 4) if ln[dom] else None, 'traffic': traffic,
            'note': None if ln[dom] else 'no large-GEMM launch (M >= 2048 rows) in this configuration: the per-launch events cover those only',
            'traffic_note': ('HBM bytes per launch from a COMMITTED rocprofv3 PMC pass, not measured in this run: (2*FETCH_SIZE+WRITE_SIZE)*1024 '
                             'in %s (tools/pmc_traffic.sh regenerates it)' % os.path.relpath(tfile, REPO)) if traffic else None,
            'launches': int(ln[dom]), 'avg_launch_ms': round(kms[dom] / max(1, ln[dom]), 4),
            'avg_launch_gflop': round(fl[dom] / max(1, ln[dom]) / 1e9, 3),
            'timing': 'HIP events bound to each kernel dispatch (hipExtLaunchKernelGGL start / stop events on the launch stream: kernel '
                      'begins executing -> kernel complete, the interval rocprofv3 --kernel-trace reports), every large-GEMM launch of '
                      'the timed region',
            'all_large_gemm_tflops': round(gemm_all, 2),
            'stream_markers': {
                'note': 'the same launches bracketed by hipEventRecord markers on the launch stream: the interval also holds the time '
                        'the dispatch waited behind the other stream\'s (higher-priority) decode kernels, so with the 2-slot pipeline '
                        'it is longer than the kernel ran',
                'achieved': round(mark_tf, 2), 'frac': round(mark_tf / PEAK_BF16_TFLOPS, 4),
                'avg_launch_ms': round(ms[dom] / max(1, ln[dom]), 4)},
            'achieved_busy': round(fl[dom] / (kbusy[dom] * 1e-3) / 1e12, 2) if kbusy[dom] > 0 else None,
            'frac_busy': round(fl[dom] / (kbusy[dom] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4) if kbusy[dom] > 0 else None,
            'busy_note': 'flops / length of the UNION of this kernel\'s execution intervals; differs from `achieved` (flops / summed '
                         'durations) because launches of the kernel overlap each other: the 4 tag blocks run on an engine side stream '
                         'next to caption blocks 8-11, and with encode_parts > 1 the batch parts run as separate chains',
            'isolated': None if iso is None or iso[0][dom] <= 0 else {
                'note': 'up to 20 of the same steps as ONE chain on one stream (no co-running kernels), untimed second pass',
                'achieved': round(iso[1][dom] / (iso[0][dom] * 1e-3) / 1e12, 2),
                'frac': round(iso[1][dom] / (iso[0][dom] * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                'all_large_gemm_tflops': round(sum(iso[1]) / (sum(iso[0]) * 1e-3) / 1e12, 2)},
            # the sampled steps are 0, STRIDE, 2 STRIDE, ... of the region
            'large_gemm_share_of_step_time': round(tot_ms / len(range(0, args.steps, TIMING_STRIDE)) / (elapsed / args.steps * 1e3), 4),
            'timing_sample': 'the large-GEMM launches of every %dth step of the timed region carry the events (launches / ms are the sampled ones); on every step they would cost 2.6 %% of it' % TIMING_STRIDE,
            'events_region': 'the per-launch events are those of timed region #%d of %d (the last armed one); value / ms_per_step are the median region\'s' % (events_region + 1, len(regions)),
            'per_variant': {VARIANT_NAMES.get(i, str(i)): {'launches': int(ln[i]), 'ms': round(kms[i], 3),
                                                            'tflops': round(fl[i] / (kms[i] * 1e-3) / 1e12, 2)}
                            for i in range(12) if ln[i] > 0},
        },
    }
    if not args.no_cpu_baseline and world == 1:
        out['cpu_baseline'] = cpu_baseline()
    emit(out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
