"""How much of the decode chain does the 2-slot pipeline hide?  Times, on the same box and with the pipeline's options:
  (a) encode + prefill only, back to back on one stream (2 encoder parts as the pipeline runs them) -- the MFMA-bound part alone,
  (b) decode only (isolated chain),
  (c) the full 2-slot pipeline (bench.py's timed loop).
(c) - (a) = what the decode chain still costs per batch although it overlaps the next batch's encoder.
Usage: python tools/encode_only_bench.py [B=64] [steps=60] [phase]
phase = enc | dec | pipe: only that phase, `steps` times (for tools/power_during.py: board power of each phase on its own -> the
energy balance of the pipeline, profiles/r05_energy_balance.txt)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vitcap_amd import _lib as L
from vitcap_amd import weights as W
from vitcap_amd._lib import check, lib
from vitcap_amd.model import ImageCaptioning


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    phase = sys.argv[3] if len(sys.argv) > 3 else None
    max_length = int(sys.argv[4]) if len(sys.argv) > 4 else 20          # decode steps = max_length - 1 (marginal cost of a step inside the pipeline)
    model = ImageCaptioning().load_recipe(0).eval()
    model.pack('cuda:0')
    img = torch.from_numpy(W.gen_image_batch(B, 1234)).cuda().to(torch.bfloat16).contiguous()
    popts = model.gen_options(gemm_mode=L.GEMM_TILES, num_beams=1, num_keep_best=1, do_sample=False, num_return_sequences=1, max_length=max_length)
    dev = img.device
    ws, need = model._workspace(B, dev, 'eo', popts)
    wp = C.c_void_p(ws.data_ptr())
    ids, lp = model._out_buffers(B, popts, dev)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        sp = C.c_void_p(stream.cuda_stream)

        def enc():
            check(lib.vitcap_engine_encode(model._engine, C.c_void_p(img.data_ptr()), 1, B, C.byref(popts), wp, need, sp), 'encode')
            check(lib.vitcap_engine_prefill(model._engine, B, C.byref(popts), wp, need, sp), 'prefill')

        def dec():
            check(lib.vitcap_engine_decode(model._engine, B, C.byref(popts), wp, need, C.c_void_p(ids.data_ptr()), C.c_void_p(lp.data_ptr()),
                                           None, sp), 'decode')

        def timed(fn, n):
            for _ in range(3):
                fn()
            stream.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            stream.synchronize()
            return (time.perf_counter() - t0) / n * 1e3
        if phase in ('enc', 'dec'):
            if phase == 'dec':
                enc()                      # the decode chain needs a prefilled workspace
            t = timed(enc if phase == 'enc' else dec, steps)
            print('B=%d | %s alone %.3f ms x %d' % (B, 'encode+prefill' if phase == 'enc' else 'decode', t, steps))
            return
        t_enc = t_dec = 0.0
        if phase is None:
            t_enc = timed(enc, steps)
            t_dec = timed(dec, steps)
        model.prime_pipeline(B, dev, opts=popts)
        for _ in range(3):
            model.generate_async(img, opts=popts).result()
        stream.synchronize()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pend = None
        for _ in range(steps):
            pend = model.generate_async(img, opts=popts)
        pend.result()
        torch.cuda.synchronize()
        t_pipe = (time.perf_counter() - t0) / steps * 1e3
    if phase == 'pipe':
        print('B=%d | max_length %d | 2-slot pipeline %.3f ms/batch x %d' % (B, max_length, t_pipe, steps))
        return
    print('B=%d | encode+prefill alone %.3f ms | decode alone %.3f ms | sum %.3f | 2-slot pipeline %.3f ms/batch (%.0f img/s) | '
          'decode cost not hidden %.3f ms (%.1f %% of the pipeline step)' % (
              B, t_enc, t_dec, t_enc + t_dec, t_pipe, B / t_pipe * 1e3, t_pipe - t_enc, (t_pipe - t_enc) / t_pipe * 100))


if __name__ == '__main__':
    main()
