import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vitcap_amd import weights as W
from vitcap_amd.model import ImageCaptioning
from vitcap_amd.synthetic import synthetic_train_inputs
from vitcap_amd.train import TrainEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
eng = TrainEngine(ImageCaptioning().load_recipe(0), 'cuda', max_iter=100)
print('engine resident: %.2f GB' % (torch.cuda.memory_allocated() / 1e9))
b = {k: v.cuda() for k, v in synthetic_train_inputs(B).items()}
b['image'] = torch.from_numpy(W.gen_image_batch(B, 1)).cuda().to(torch.bfloat16)
torch.cuda.reset_peak_memory_stats()
eng.train_step(b); eng.train_step(b)
torch.cuda.synchronize()
print('B=%d peak allocated: %.2f GB' % (B, torch.cuda.max_memory_allocated() / 1e9))
