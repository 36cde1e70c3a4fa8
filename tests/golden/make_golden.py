"""Generate the golden vectors in this directory by running the REAL reference
(/root/reference, imported through oracle/ref_harness.py) on CPU.

Run in the build container only:   python tests/golden/make_golden.py
Outputs (committed):  tests/golden/tiny_{t18,intended}.npz, tests/golden/vitb16_b4_t18.npz,
                      tests/golden/tiny_w2_t18.npz, tests/golden/tiny_w4_t18.npz
Every value stored here was produced by the reference's own modules.modeling.SegCLIP; weights and
inputs are the closed-form / seeded generators of segclip_amd/synth.py (so they are reproducible
on the GPU box without any file), and the RNG draws are injected (oracle.ref_harness.NoiseTap).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref_harness as rh  # noqa: E402
from segclip_amd import synth  # noqa: E402

FLAGS = dict(use_seglabel=True, use_vision_mae_recon=True)


def run_reference(spec, B, seed, mode, rank=0, world=1, full_tensors=True, flags=FLAGS):
    model, args = rh.build_reference_model(spec, flags, rank=rank, world_size=world, cross_mode=mode)
    synth.apply_closed_form_weights(model)
    # every rank draws the same global batch and takes its own slice (rank-ordered)
    gbatch = synth.synthetic_batch(spec, B * world, seed=seed)
    gnoise = synth.synthetic_noise(spec, B * world, seed=seed)
    sl = slice(rank * B, (rank + 1) * B)
    batch = {k: v[sl] for k, v in gbatch.items()}
    noise = {k: v[sl] for k, v in gnoise.items()}
    cap = {}
    vt = model.clip.visual.transformer
    sem_calls, l0_calls = [], []
    h1 = vt.semantic_layer2.register_forward_hook(lambda m, i, o: sem_calls.append([t.detach().clone() for t in o]))
    h2 = vt.layers0.register_forward_hook(lambda m, i, o: l0_calls.append(o.detach().clone()))
    orig_sim = model._loose_similarity

    def sim(*a, **k):
        r = orig_sim(*a, **k)
        cap["t2v"], cap["v2t"] = r[0].detach().clone(), r[1].detach().clone()
        return r

    model._loose_similarity = sim
    orig_txt, orig_img = model.clip.encode_text, model.clip.encode_image
    txt_calls, img_calls = [], []

    def enc_t(*a, **k):
        r = orig_txt(*a, **k)
        txt_calls.append(r)
        return r

    def enc_i(*a, **k):
        r = orig_img(*a, **k)
        img_calls.append(r)
        return r

    model.clip.encode_text, model.clip.encode_image = enc_t, enc_i
    if flags.get("use_vision_mae_recon"):
        orig_mae = model.vis_mae_decoder.forward_vis

        def mae(image, vis_hidden, mask, ids_restore, **k):
            r = orig_mae(image, vis_hidden, mask, ids_restore, **k)
            cap["loss_mae"] = r.detach().clone()
            cap["mae_mask"] = mask.detach().clone()
            cap["ids_restore"] = ids_restore.detach().clone()
            return r

        model.vis_mae_decoder.forward_vis = mae
    inject = [("gumbel", noise["gumbel_main"])]
    if flags.get("use_vision_mae_recon"):
        inject += [("rand", noise["mask_noise"]), ("gumbel", noise["gumbel_mae"])]
    with rh.NoiseTap(inject=inject):
        loss = model(batch["input_ids"], batch["segment_ids"], batch["input_mask"], batch["image"].double(),
                     image_seg=batch["image_seg"])
    loss.backward()
    h1.remove()
    h2.remove()
    out = {"loss": loss.detach()}
    Bv = batch["image"].shape[0]
    labels = torch.arange(Bv) + Bv * rank
    ce = torch.nn.functional.cross_entropy
    out["loss_contrastive"] = (ce(cap["t2v"], labels) + ce(cap["v2t"], labels)) / 2
    out["t2v"], out["v2t"] = cap["t2v"], cap["v2t"]
    if "loss_mae" in cap:
        out["loss_mae"] = cap["loss_mae"]
        out["mae_mask"] = cap["mae_mask"]
        out["ids_restore"] = cap["ids_restore"]
        out["mae_hard_idx"] = sem_calls[1][1].argmax(dim=1)
    if flags.get("use_seglabel"):
        out["loss_kl"] = out["loss"] - out["loss_contrastive"] - out.get("loss_mae", 0.0)
    t_feat, t_hidden = txt_calls[0]
    v_feat, v_hidden, _ = img_calls[0]
    out["text_feat"], out["image_feat"] = t_feat.detach(), v_feat.detach()
    out["eot"] = batch["input_ids"][:, 0].argmax(dim=-1)
    hard = sem_calls[0][1]
    assert set(hard.unique().tolist()) <= {0.0, 1.0} and bool((hard.sum(1) == 1).all())
    out["hard_idx"] = hard.argmax(dim=1)
    soft = sem_calls[0][2]
    if full_tensors:
        out["text_hidden"], out["image_hidden"] = t_hidden.detach(), v_hidden.detach()
        out["layers0_out"] = l0_calls[0]
        out["soft"] = soft
        out["center_q"] = sem_calls[0][3]
        if len(l0_calls) > 1:
            out["mae_layers0_out"] = l0_calls[1]
    else:
        for k, t in (("text_hidden", t_hidden), ("image_hidden", v_hidden), ("layers0_out", l0_calls[0]), ("soft", soft)):
            out["sum_" + k] = t.detach().double().sum().float()
            out["abssum_" + k] = t.detach().double().abs().sum().float()
    names, norms, none_grad = [], [], []
    for n, p in model.named_parameters():
        if p.grad is None:
            none_grad.append(n)
            continue
        names.append(n)
        norms.append(p.grad.double().norm().item())
    out["grad_norms"] = torch.tensor(norms, dtype=torch.float64)
    keep_full = ["clip.logit_scale", "clip.visual.transformer.semantic_layer2.semantic_center",
                 "clip.visual.ln_pre.weight", "clip.ln_final.bias", "clip.visual.proj",
                 "clip.visual.transformer.layers0.0.attn.in_proj_bias",
                 "clip.transformer.resblocks.0.ln_1.weight",
                 "clip.visual.transformer.reconstruct_layer2.rec_proj_a.a_fc.weight"]
    pd = dict(model.named_parameters())
    for n in keep_full:
        if n in pd and pd[n].grad is not None and (full_tensors or pd[n].numel() <= 4096):
            out["grad::" + n] = pd[n].grad.detach()
    meta = dict(grad_names=names, none_grad=none_grad)
    return out, meta


def save(path, out, meta, extra=None):
    arrs = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in out.items()}
    arrs["grad_names"] = np.array(meta["grad_names"])
    arrs["none_grad"] = np.array(meta["none_grad"])
    for k, v in (extra or {}).items():
        arrs[k] = np.asarray(v)
    np.savez_compressed(path, **arrs)
    print("wrote", path, "%.1f KiB" % (os.path.getsize(path) / 1024), "loss", float(out["loss"]))


def _rank_main(rank, world, spec_name, B, seed, mode, port, path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.manual_seed(0)
    torch.set_num_threads(2)
    rh.ensure_process_group(rank, world, port)
    out, meta = run_reference(synth.SPECS[spec_name], B, seed, mode, rank=rank, world=world, full_tensors=False)
    keep = {k: out[k] for k in ("loss", "loss_contrastive", "loss_kl", "loss_mae", "t2v", "v2t", "hard_idx",
                                "ids_restore", "grad_norms")}
    torch.save((keep, meta), f"{path}.rank{rank}.pt")


def multi_rank(spec_name, world, B, seed, mode, port, path):
    import torch.multiprocessing as mp
    mp.spawn(_rank_main, args=(world, spec_name, B, seed, mode, port, path), nprocs=world, join=True)
    arrs = {}
    for r in range(world):
        keep, meta = torch.load(f"{path}.rank{r}.pt")
        os.remove(f"{path}.rank{r}.pt")
        for k, v in keep.items():
            arrs[f"r{r}_{k}"] = v.detach().numpy()
        arrs["grad_names"] = np.array(meta["grad_names"])
    arrs["world"], arrs["B"], arrs["seed"] = np.asarray(world), np.asarray(B), np.asarray(seed)
    np.savez_compressed(path, **arrs)
    print("wrote", path, [float(arrs[f"r{r}_loss"]) for r in range(world)])


if __name__ == "__main__":
    which = sys.argv[1:] or ["tiny", "vitb16", "dist"]
    torch.manual_seed(0)
    if "dist" in which:
        multi_rank("tiny", 2, 2, 3, "t18", 29531, os.path.join(HERE, "tiny_w2_t18.npz"))
        multi_rank("tiny", 4, 2, 3, "t18", 29532, os.path.join(HERE, "tiny_w4_t18.npz"))
    if "tiny" in which:
        for mode in ("t18", "intended"):
            out, meta = run_reference(synth.SPECS["tiny"], 3, 1, mode)
            save(os.path.join(HERE, f"tiny_{mode}.npz"), out, meta, dict(B=3, seed=1))
    if "vitb16" in which:
        out, meta = run_reference(synth.SPECS["vitb16"], 4, 2, "t18", full_tensors=False)
        save(os.path.join(HERE, "vitb16_b4_t18.npz"), out, meta, dict(B=4, seed=2))
