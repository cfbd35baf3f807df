"""Golden vectors for the cross-entropy training step, produced by the REFERENCE itself (container only).
Run:  python tests/golden/make_golden_train.py      (writes tests/golden/reference_train.npz)
The reference's ViTCAP.encode_forward(is_training=True), autograd, torch clip_grad_norm_ and its own solver.AdamW are
run on the seeded weights / synthetic batch of oracle.synthetic_train_inputs with attention dropout off (eval mode)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402


def main():
    MG.install_shims()
    from vitcap_amd import weights as W
    from oracle import vitcap_oracle as O
    torch.set_num_threads(8)
    sd_np = W.make_state_dict(seed=0, tie_weights=True)
    model, enc = MG.build_reference('cls', True)
    MG.load_recipe(model, enc, sd_np)
    model.eval()          # dropout off; the training branch is selected by is_training=True
    B = 2
    img = torch.from_numpy(W.gen_image_batch(B, 1234))
    batch = O.synthetic_train_inputs(B)
    for p in list(model.parameters()) + list(enc.parameters()):
        p.requires_grad_(True)
    img_feats = enc(img)
    full = O.construct_attn_mask(batch['attention_mask'], img_feats.shape[1])
    res = model(input_ids=batch['input_ids'].clone(), img_feats=img_feats, attention_mask=full,
                masked_pos=batch['masked_pos'].clone(), masked_ids=batch['masked_ids'].clone(),
                token_type_ids=batch['token_type_ids'], label=batch['label'], is_training=True, return_dict=True,
                gen_tag_ratio=None)
    loss = res['masked_loss']
    loss.backward()
    named = {('module.' + k): p for k, p in model.named_parameters()}
    named.update({('image_encoder.module.' + k): p for k, p in enc.named_parameters()})
    params = list(model.parameters()) + list(enc.parameters())
    total = torch.nn.utils.clip_grad_norm_(params, 1.0)
    out = {'masked_loss': np.array(float(loss)), 'tag_loss': np.array(float(res['tag_loss'])),
           'grad_norm': np.array(float(total)), 'class_logits_head': res['class_logits'][:, :64].detach().numpy()}
    nograd = sorted(k for k, p in named.items() if p.grad is None)
    out['no_grad_keys'] = np.array(nograd)
    probe = ['module.bert.decoder.layer.3.output.dense.weight', 'module.bert.decoder.layer.0.attention.self.key.weight',
             'module.bert.encoder.blocks.11.mlp.fc1.weight', 'module.bert.encoder.blocks.0.attn.qkv.weight',
             'module.bert.encoder.tag_blocks.3.attn.proj.bias', 'module.bert.encoder.blocks.5.norm1.weight',
             'module.bert.embeddings.word_embeddings.weight', 'module.bert.embeddings.position_embeddings.weight',
             'module.cls.predictions.transform.dense.weight', 'module.cls.predictions.bias',
             'image_encoder.module.patch_embed.proj.weight', 'image_encoder.module.pos_embed',
             'image_encoder.module.cls_token', 'module.bert.decoder.layer.2.attention.output.LayerNorm.weight']
    for k in probe:
        g = named[k].grad
        out['grad_norm__' + k] = np.array(float(g.norm()))
        out['grad_head__' + k] = g.reshape(-1)[:32].numpy().copy()
    # optimizer: the reference's own AdamW on the oracle's restatement of the 10 module groups
    from src.solver import AdamW
    pg = O.param_groups(named.keys())
    groups = {}
    for k, p in named.items():
        if pg[k] is None or p.grad is None:
            continue
        groups.setdefault(pg[k], []).append(p)
    opt = AdamW([{'params': ps, 'lr': lr, 'weight_decay': wd} for (lr, wd), ps in groups.items()], lr=1e-4, eps=1e-8)
    before = {k: named[k].detach().clone() for k in probe}
    opt.step()
    for k in probe:
        out['delta_head__' + k] = (named[k].detach() - before[k]).reshape(-1)[:32].numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'reference_train.npz'), **out)
    print('masked_loss', float(loss), 'tag_loss', float(res['tag_loss']), 'grad_norm', float(total))
    print('no grad:', len(nograd))


if __name__ == '__main__':
    main()
