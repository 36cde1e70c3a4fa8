"""Golden vectors for the training-step tail, produced by the REAL reference on CPU.

Run in the build container only:   python tests/golden/make_golden_train.py
Outputs (committed):
  tests/golden/adamw_steps.npz   the reference's own AdaptAdamW (modules/optimization_adamw.py) driven for 4
                                 steps over seeded tensors in 3 param groups (cosine / linear / constant lr)
  tests/golden/train_tiny_t18.npz  3 iterations of the reference's own train_epoch + prep_optimizer + freeze
                                 block (main_task_align.py) on the `tiny` spec, all three losses, injected noise

The reference driver is imported as a module (its import-time init_process_group("nccl") is neutralised, the
tokenizer's `ftfy` and the `dataloaders` package are stubbed: nothing on the measured path touches them); the
freeze block lives inside main(), so it is cut out of main()'s source at run time and executed on the model.
"""
import inspect
import logging
import os
import sys
import textwrap
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from oracle import ref_harness as rh  # noqa: E402
from segclip_amd import synth  # noqa: E402

FLAGS = dict(use_seglabel=True, use_vision_mae_recon=True)
TRAIN_ARGS = dict(lr=4e-3, lower_lr=4e-6, lower_text_lr=0.0, coef_lr=1.0, weight_decay=0.05, opt_b1=0.9, opt_b2=0.98,
                  eps=1e-6, warmup_proportion=0.2, lr_start=0.0, lr_end=0.0, clip_grad=1.0,
                  gradient_accumulation_steps=1, freeze_layer_num=0, freeze_text_layer_num=0, first_stage_layer=10,
                  pretrained_clip_name="ViT-B/16", n_display=100, disable_amp=True, local_rank=0, epochs=1)
T_TOTAL = 10
STEPS = 3
BATCH = 4


def import_driver():
    import torch.distributed as dist

    rh.import_reference("t18")
    rh.ensure_process_group(0, 1)
    rh._stub("ftfy")
    rh._stub("dataloaders")
    rh._stub("dataloaders.data_dataloaders", DATALOADER_DICT={})
    orig = dist.init_process_group
    dist.init_process_group = lambda *a, **k: None
    try:
        import main_task_align as mta
    finally:
        dist.init_process_group = orig
    mta.logger = logging.getLogger("golden")
    return mta


def run_freeze_block(mta, args, model):
    src = inspect.getsource(mta.main).splitlines()
    lo = next(i for i, l in enumerate(src) if "assert args.freeze_layer_num" in l)
    hi = next(i for i, l in enumerate(src) if "dataloader loading" in l) - 1
    code = textwrap.dedent("\n".join(src[lo:hi]))
    exec(compile(code, "<main_task_align.main freeze block>", "exec"), dict(args=args, model=model, logger=mta.logger))


def golden_adamw(mta):
    g = torch.Generator().manual_seed(7)
    shapes = [(5, 7), (33,), (4, 3, 2), (1,), (257,)]
    params = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
    groups = [dict(params=params[:2], lr=1e-2, weight_decay=0.05),
              dict(params=params[2:4], lr=3e-3, weight_decay=0.0, schedule="warmup_linear"),
              dict(params=params[4:], weight_decay=0.2, schedule="warmup_constant", t_total=-1)]
    opt = mta.AdaptAdamW(groups, lr=5e-2, warmup=0.25, schedule="warmup_cosine", b1=0.9, b2=0.98, e=1e-6, t_total=8,
                         weight_decay=0.05, max_grad_norm=1.0, lr_start=0.1, lr_end=0.05)
    out = {"n_params": len(params)}
    for i, p in enumerate(params):
        out[f"p{i}_init"] = p.detach().clone()
    lrs = []
    for step in range(4):
        for i, p in enumerate(params):
            if i == 3 and step == 1:
                p.grad = None  # a parameter without a gradient is skipped and its step does not advance
                continue
            gr = torch.randn(p.shape, generator=g) * (10.0 ** (-(i % 3) * 2))
            p.grad = gr
            out[f"g{i}_s{step}"] = gr.clone()
        opt.step()
        lrs.append(sorted(set(opt.get_lr())))
        for i, p in enumerate(params):
            out[f"p{i}_s{step}"] = p.detach().clone()
    for i, p in enumerate(params):
        out[f"m{i}"] = opt.state[p]["exp_avg"].clone()
        out[f"v{i}"] = opt.state[p]["exp_avg_sq"].clone()
        out[f"step{i}"] = opt.state[p]["step"]
    out["lrs"] = np.array([l + [np.nan] * (8 - len(l)) for l in lrs])
    np.savez_compressed(os.path.join(HERE, "adamw_steps.npz"),
                        **{k: (v.numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in out.items()})
    print("adamw_steps: lrs", lrs)


def golden_train(mta):
    spec = synth.SPECS["tiny"]
    model, _ = rh.build_reference_model(spec, FLAGS, rank=0, world_size=1, cross_mode="t18")
    synth.apply_closed_form_weights(model)
    args = types.SimpleNamespace(**TRAIN_ARGS)
    run_freeze_block(mta, args, model)
    frozen = [n for n, p in model.named_parameters() if not p.requires_grad]
    ddp = torch.nn.parallel.DistributedDataParallel
    torch.nn.parallel.DistributedDataParallel = lambda m, **k: m  # CPU run: no device_ids wrapper
    try:
        optimizer, scheduler, model, scaler = mta.prep_optimizer(args, model, T_TOTAL, torch.device("cpu"), 1, 0)
    finally:
        torch.nn.parallel.DistributedDataParallel = ddp
    pname = {id(p): n for n, p in model.named_parameters()}
    group_names = [[pname[id(p)] for p in g["params"]] for g in optimizer.param_groups]

    loader, inject = [], []
    for s in range(STEPS):
        b = synth.synthetic_batch(spec, BATCH, seed=100 + s)
        nz = synth.synthetic_noise(spec, BATCH, seed=100 + s)
        loader.append((b["input_ids"], b["input_mask"], b["segment_ids"], b["image"].double(), torch.zeros(BATCH, 4),
                       b["image_seg"]))
        inject += [("gumbel", nz["gumbel_main"]), ("rand", nz["mask_noise"]), ("gumbel", nz["gumbel_mae"])]

    losses, lrs, gnorms = [], [], []
    orig_clip = torch.nn.utils.clip_grad_norm_

    def tap_clip(params, max_norm, *a, **k):
        r = orig_clip(params, max_norm, *a, **k)
        gnorms.append(float(r))
        return r

    orig_step = optimizer.step

    def tap_step(*a, **k):
        r = orig_step(*a, **k)
        lrs.append(sorted(set(optimizer.get_lr())))
        return r

    orig_fwd = model.forward

    def tap_fwd(*a, **k):
        r = orig_fwd(*a, **k)
        losses.append(float(r))
        return r

    torch.nn.utils.clip_grad_norm_ = tap_clip
    optimizer.step = tap_step
    model.forward = tap_fwd
    try:
        with rh.NoiseTap(inject=inject):
            total_loss, global_step = mta.train_epoch(0, args, model, loader, torch.device("cpu"), 1, optimizer,
                                                      scheduler, 0, scaler, local_rank=0)
    finally:
        torch.nn.utils.clip_grad_norm_ = orig_clip
    out = dict(losses=np.array(losses), total_loss=total_loss, global_step=global_step, grad_norms=np.array(gnorms),
               lrs=np.array([l + [np.nan] * (8 - len(l)) for l in lrs]), frozen=np.array(frozen),
               t_total=T_TOTAL, steps=STEPS, batch=BATCH)
    for gi, names in enumerate(group_names):
        out[f"group{gi}"] = np.array(names if names else [""])
    names, sums, abssums = [], [], []
    for n, p in model.named_parameters():
        names.append(n)
        sums.append(p.detach().double().sum().item())
        abssums.append(p.detach().double().abs().sum().item())
    out["param_names"], out["param_sum"], out["param_abssum"] = np.array(names), np.array(sums), np.array(abssums)
    pd = dict(model.named_parameters())
    for n in ["clip.logit_scale", "clip.visual.transformer.semantic_layer2.semantic_center", "clip.ln_final.bias",
              "clip.visual.transformer.layers0.0.attn.in_proj_bias", "clip.visual.proj",
              "vis_mae_decoder.mask_token"]:
        if n in pd:
            out["final::" + n] = pd[n].detach().numpy()
    for k, v in TRAIN_ARGS.items():
        if isinstance(v, (int, float)) and not isinstance(v, bool):
            out["arg::" + k] = v
    np.savez_compressed(os.path.join(HERE, "train_tiny_t18.npz"), **out)
    print("train_tiny_t18: losses", losses, "lrs", lrs, "grad_norms", gnorms, "frozen", len(frozen),
          [len(g) for g in group_names])


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    mta = import_driver()
    golden_adamw(mta)
    golden_train(mta)
