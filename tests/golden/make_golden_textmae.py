"""Golden vectors for the text-MAE branch (modules/modeling.py:226-236, modules/module_clip.py:113-124,
modules/module_mae.py:332-355), produced by the REAL reference on CPU.

Run in the build container only:   python tests/golden/make_golden_textmae.py
Output (committed): tests/golden/textmae_tiny.npz - tiny model, flags use_text_mae_recon (+ contrastive), B=3, seed 31:
loss terms, the text masking (mask, ids_restore), the masked text hidden states, logits, every parameter-gradient norm
and a few full gradient tensors.  RNG draws are injected (Gumbel of the center stage, rand(N, L) of the text masking).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref_harness as rh  # noqa: E402
from segclip_amd import synth  # noqa: E402

B, SEED = 3, 31
FLAGS = dict(use_text_mae_recon=True)

if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(4)
    spec = synth.SPECS["tiny"]
    model, _ = rh.build_reference_model(spec, FLAGS, rank=0, world_size=1, cross_mode="t18")
    synth.apply_closed_form_weights(model)
    model.train()
    batch = synth.synthetic_batch(spec, B, seed=SEED, with_seg=False)
    noise = synth.synthetic_noise(spec, B, seed=SEED)
    cap = {}
    orig_seq = model.seq_mae_decoder.forward_seq

    def fseq(input_ids, seq_hidden, mae_mask, ids_restore, attention_mask):
        r = orig_seq(input_ids, seq_hidden, mae_mask, ids_restore, attention_mask)
        cap.update(loss_text_mae=r.detach().clone(), text_mae_hidden=seq_hidden.detach().clone(),
                   text_final_mask=mae_mask.detach().clone(), text_ids_restore=ids_restore.detach().clone())
        return r

    model.seq_mae_decoder.forward_seq = fseq
    orig_enc = model.clip.encode_text

    def enc(text, **kw):
        r = orig_enc(text, **kw)
        if kw.get("mask_ratio", 0.) > 0:
            cap["text_mae_mask"] = r[2].detach().clone()
        return r

    model.clip.encode_text = enc
    orig_sim = model._loose_similarity

    def sim(*a, **k):
        r = orig_sim(*a, **k)
        cap["t2v"], cap["v2t"] = r[0].detach().clone(), r[1].detach().clone()
        return r

    model._loose_similarity = sim
    with rh.NoiseTap(inject=[("gumbel", noise["gumbel_main"]), ("rand", noise["text_mask_noise"])]):
        loss = model(batch["input_ids"], batch["segment_ids"], batch["input_mask"], batch["image"].double())
    loss.backward()
    out = {"B": np.asarray(B), "seed": np.asarray(SEED), "loss": loss.detach().numpy()}
    for k, v in cap.items():
        out[k] = v.numpy()
    names, norms, none_grad = [], [], []
    for n, p in model.named_parameters():
        if p.grad is None:
            none_grad.append(n)
        else:
            names.append(n)
            norms.append(float(p.grad.double().norm()))
    out["grad_names"], out["grad_norms"], out["none_grad"] = np.array(names), np.array(norms), np.array(none_grad)
    pd = dict(model.named_parameters())
    for n in ("seq_mae_decoder.mask_token", "seq_mae_decoder.decoder_blocks.0.attn.in_proj_bias",
              "seq_mae_decoder.decoder_pred.bias", "clip.ln_final.weight", "clip.transformer.resblocks.0.ln_1.weight"):
        out["grad::" + n] = pd[n].grad.detach().numpy()
    out["decoder_pos_embed"] = model.seq_mae_decoder.decoder_pos_embed.detach().numpy()
    np.savez_compressed(os.path.join(HERE, "textmae_tiny.npz"), **out)
    print({k: (v.shape if hasattr(v, "shape") and v.shape else v) for k, v in out.items() if not k.startswith("grad_n")})
