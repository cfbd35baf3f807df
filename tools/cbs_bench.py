"""Constrained beam search throughput: python tools/cbs_bench.py [images] [beams]   (8 FSM states, two one-word constraints per image)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from vitcap_amd import cbs, weights as W
from vitcap_amd.model import ImageCaptioning


class Tok(object):
    vocab_size = 30522

    @staticmethod
    def convert_tokens_to_ids(tokens):
        return [{'dog': 3899, 'dogs': 6077, 'cat': 4937, 'tree': 3392}[t] for t in tokens]


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    m = ImageCaptioning(tie_weights=True, tagemb='cls').load_recipe(0).eval()
    m.pack('cuda')
    builder = cbs.FiniteStateMachineBuilder(Tok, {'dog': ['dog'], 'cat': ['cat'], 'tree': ['tree']}, {'dog': ['dog', 'dogs']}, 3)
    fsm, ncons = cbs.batch_fsm(builder, [['dog', 'tree'] if b % 2 else ['cat', 'dog'] for b in range(B)], device='cuda')
    img = torch.from_numpy(W.gen_image_batch(B, 5)).cuda().to(torch.bfloat16)
    for _ in range(2):
        m.generate_cbs(img, fsm, ncons, num_beams=K)
    torch.cuda.synchronize()
    t0 = time.time()
    n = 5
    for _ in range(n):
        ids, lp = m.generate_cbs(img, fsm, ncons, num_beams=K)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / n
    print('constrained beam search: %d images x %d states x %d beams = %d sequences: %.1f ms per batch, %.0f images/s' % (
        B, fsm.shape[1], K, B * fsm.shape[1] * K, dt * 1e3, B / dt))
    dtb = None
    if K > 1:
        for _ in range(2):
            m.generate_beam(img, K)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(n):
            m.generate_beam(img, K)
        torch.cuda.synchronize()
        dtb = (time.time() - t0) / n
        print('plain beam search, %d beams: %.1f ms per batch, %.0f images/s' % (K, dtb * 1e3, B / dtb))


if __name__ == '__main__':
    main()
