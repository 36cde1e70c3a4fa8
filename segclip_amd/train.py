"""Training-step driver counterpart of the reference's main_task_align.py: the caller contract of
the hot path, with the per-iteration host round-trips removed.

  freeze_parameters(args, model)                   main_task_align.py:388-441
  prep_optimizer(args, model, t_total, ...)        main_task_align.py:175-256   (8 name-routed param groups)
  train_epoch(epoch, args, model, loader, ...)     main_task_align.py:292-359
  TrainTail                                        :326-347 fused: clip_grad_norm_ -> AdaptAdamW.step (skipped on the
                                                   device when the loss is NaN) -> zero_grad -> clamp(logit_scale)

The reference synchronises with the host three times per iteration (`float(loss)`, `int(torch.isnan(loss))`
twice); here the loss, the NaN decision, the gradient norm and the clip coefficient stay in a 32-byte device
control block (include/segclip_hip.h: segclip_train_ctrl) and are read back only when something is logged.
"""
import ctypes as C
import logging
import math
import time

import torch

from . import _lib as L
from .modules.optimization_adamw import AdaptAdamW

logger = logging.getLogger(__name__)
LN100 = math.log(100.0)

NO_DECAY = ('bias', 'LayerNorm.bias', 'LayerNorm.weight')
_CLIP_INIT_HEADS = ("clip.visual.class_embedding", "clip.visual.positional_embedding", "clip.visual.conv1.",
                    "clip.visual.ln_pre.", "clip.logit_scale", "clip.ln_final.", "clip.text_projection")
_TEXT_EMBED_HEADS = ("clip.positional_embedding", "clip.token_embedding.")
_CLIP_INIT_LAYERS = ("clip.visual.transformer.layers0.", "clip.transformer.resblocks.")


def _unwrap(model):
    return model.module if hasattr(model, 'module') else model


def _layer_no(name, key):
    return int(name.split(key)[1].split(".")[0])


def freeze_parameters(args, model):
    """Switch requires_grad off exactly where main() does (main_task_align.py:388-441).  Returns the names."""
    model = _unwrap(model)
    if not hasattr(model, "clip"):
        return []
    fl = getattr(args, "freeze_layer_num", 0)
    ftl = getattr(args, "freeze_text_layer_num", 0)
    first = getattr(args, "first_stage_layer", 10)
    assert -1 <= fl <= 12
    always = ("ln_final.", "text_projection", "logit_scale", "visual.ln_post.", "visual.proj")
    new_parts = ("visual.transformer.semantic_layer1", "visual.transformer.semantic_layer2",
                 "visual.transformer.layers_mae", "visual.transformer.reconstruct_layer")
    frozen = []
    for name, param in model.clip.named_parameters():
        off = False
        if fl > -1:
            if name.startswith(always) or name.startswith(new_parts):
                pass
            elif name.startswith("visual.transformer.layers0."):
                off = _layer_no(name, ".layers0.") < fl
            elif name.startswith("visual.transformer.layers2."):
                off = _layer_no(name, ".layers2.") < fl - first
            elif name.startswith("transformer.resblocks."):
                off = _layer_no(name, ".resblocks.") < fl
            else:
                off = True
        if ftl > 0:
            if name.startswith("positional_embedding") or name.startswith("token_embedding.weight"):
                off = True
            elif name.startswith("transformer.resblocks.") and _layer_no(name, ".resblocks.") < ftl:
                off = True
        if getattr(args, "pretrained_clip_name", "ViT-B/16") in ("ViT-B/32", "ViT-B/16", "ViT-L/14"):
            if name.startswith("visual.positional_embedding") or name.startswith("visual.conv1.weight"):
                off = True
        if off:
            param.requires_grad = False
            frozen.append("clip." + name)
    return frozen


def param_group_index(name):
    """Routing of prep_optimizer (main_task_align.py:183-220): even = decayed, odd = no_decay (a substring test, so
    only tensors named *bias* escape the decay: the LayerNorm modules here are called ln_1/ln_2/norm...)."""
    nd = int(any(s in name for s in NO_DECAY))
    if "clip." not in name:
        return 6 + nd
    if name.startswith(_CLIP_INIT_HEADS) or name.startswith(_CLIP_INIT_LAYERS):
        return nd
    if name.startswith(_TEXT_EMBED_HEADS):
        return 2 + nd
    return 4 + nd


def prep_optimizer(args, model, num_train_optimization_steps, device=None, n_gpu=1, local_rank=0, coef_lr=1.,
                   shadow_bf16=None):
    """Same return tuple as the reference: (optimizer, scheduler, model, scaler).  The model is wrapped in
    segclip_amd.dist.GradSync (DDP's contract) when a process group with more than one rank exists;
    AMP is permanently disabled in the reference (main_task_align.py:78), so the scaler is a disabled GradScaler."""
    model = _unwrap(model)
    buckets = [[] for _ in range(8)]
    for n, p in model.named_parameters():
        buckets[param_group_index(n)].append(p)
    lower_lr = args.lower_lr if getattr(args, "lower_lr", 0.) != 0. else args.lr * coef_lr
    lower_text_lr = args.lower_text_lr if getattr(args, "lower_text_lr", 0.) != 0. else lower_lr
    lrs = [lower_lr, lower_lr, lower_text_lr, lower_text_lr, args.lr, args.lr, None, None]
    groups = []
    for gi, params in enumerate(buckets):
        g = {'params': params, 'weight_decay': args.weight_decay if gi % 2 == 0 else 0.0}
        if lrs[gi] is not None:
            g['lr'] = lrs[gi]
        groups.append(g)
    from . import config
    if shadow_bf16 is None:
        shadow_bf16 = config.compute_dtype == torch.bfloat16
    if shadow_bf16 and hasattr(model, "segclip_config"):
        # the optimizer kernel rewrites the bf16 copies together with the weights: THIS model's forward may trust them
        # (scoped to the model, not the process - ADVICE r1)
        model.segclip_config["trust_weight_shadows"] = True
    optimizer = AdaptAdamW(groups, lr=args.lr, warmup=args.warmup_proportion, schedule='warmup_cosine',
                           b1=args.opt_b1, b2=args.opt_b2, e=args.eps, t_total=num_train_optimization_steps,
                           weight_decay=args.weight_decay, max_grad_norm=1.0,
                           lr_start=getattr(args, "lr_start", 0.), lr_end=getattr(args, "lr_end", 0.),
                           shadow_bf16=shadow_bf16)
    if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
        # The reference wraps the model in DistributedDataParallel(find_unused_parameters=True)
        # (main_task_align.py:251-252).  GradSync (segclip_amd/dist.py) has the same contract - `.module`, gradients
        # averaged over ranks after every backward, unused parameters keep grad None - without DDP's per-parameter
        # reducer: flat, 256-byte aligned gradient buckets the weight-gradient kernels write into directly.
        from .dist import GradSync
        model = GradSync(model)
    scaler = torch.amp.GradScaler("cuda", enabled=False)
    return optimizer, None, model, scaler


def save_model(epoch, args, model, optimizer, tr_loss, scaler=None, type_name=""):
    """main_task_align.py:258-273: bare model state_dict -> pytorch_model.bin.<epoch>, optimizer/epoch/loss ->
    pytorch_opt.bin.<epoch> (same file names and dictionary keys, so either implementation can resume the other)."""
    import os
    if hasattr(model, "drain"):
        model.drain()   # a checkpoint is a host decision: no unread GradSync verdict behind it
    tag = "{}{}".format("" if type_name == "" else type_name + ".", epoch)
    model_file = os.path.join(args.output_dir, "pytorch_model.bin." + tag)
    opt_file = os.path.join(args.output_dir, "pytorch_opt.bin." + tag)
    torch.save(_unwrap(model).state_dict(), model_file)
    torch.save({'epoch': epoch, 'optimizer_state_dict': optimizer.state_dict(), 'loss': tr_loss,
                'scaler': scaler.state_dict() if scaler is not None else {}}, opt_file)
    logger.info("Model saved to %s", model_file)
    logger.info("Optimizer saved to %s", opt_file)
    return model_file


def load_model(epoch, args, n_gpu, device, model_file=None):
    """main_task_align.py:275-290: state_dict file -> SegCLIP.from_pretrained(state_dict=...) -> device; None if absent."""
    import os
    from .modules.modeling import SegCLIP
    if model_file is None or len(model_file) == 0:
        model_file = os.path.join(args.output_dir, "pytorch_model.bin.{}".format(epoch))
    if not os.path.exists(model_file):
        return None
    state = torch.load(model_file, map_location='cpu')
    if getattr(args, "local_rank", 0) == 0:
        logger.info("Model loaded from %s", model_file)
    model = SegCLIP.from_pretrained(cache_dir=getattr(args, "cache_dir", None), state_dict=state, task_config=args)
    model.to(device)
    return model


class TrainTail:
    """clip_grad_norm_ + optimizer.step + zero_grad + logit_scale clamp with no host synchronisation."""

    def __init__(self, model, optimizer, clip_grad=1.0):
        self.model, self.optimizer, self.clip_grad = model, optimizer, float(clip_grad)
        dev = next(_unwrap(model).parameters()).device
        L.require_cuda(next(_unwrap(model).parameters()))
        self.ctrl = torch.zeros(8, dtype=torch.int32, device=dev)  # segclip_train_ctrl
        self._ws = None
        self._lib = L.load()

    def run(self, loss):
        lib = self._lib
        params = [p for p in self.model.parameters() if p.grad is not None]
        grads = []
        for p in params:
            g = p.grad
            if g.dtype != torch.float32 or not g.is_contiguous():
                g = g.float().contiguous()
                p.grad = g
            grads.append(g)
        n = len(grads)
        ptrs = (C.c_void_p * n)(*[g.data_ptr() for g in grads])
        sizes = (C.c_int64 * n)(*[g.numel() for g in grads])
        need = lib.segclip_grad_sqnorm_ws_bytes(C.cast(sizes, C.c_void_p), n)
        if self._ws is None or self._ws.numel() * 4 < need:
            self._ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=self.ctrl.device)
        loss = loss.detach()
        if loss.dtype != torch.float32:
            loss = loss.float()
        L.check(lib.segclip_grad_sqnorm(C.cast(ptrs, C.c_void_p), C.cast(sizes, C.c_void_p), n, L.ptr(self._ws),
                                        L.ptr(self.ctrl), self.clip_grad, L.stream()), "grad_sqnorm")
        self.optimizer.step(loss=loss, ctrl=self.ctrl)
        m = _unwrap(self.model)
        ls = m.clip.logit_scale if hasattr(m, "clip") else None
        L.check(lib.segclip_train_step_finish(L.ptr(self.ctrl), L.ptr(loss), L.ptr(ls.data) if ls is not None else None,
                                              LN100, L.stream()), "train_step_finish")
        self.optimizer.zero_grad(set_to_none=True)

    def read(self):
        """Host copy of the control block (synchronises): dict(grad_norm, clip_coef, nan_skips, steps, loss_sum, last_loss)."""
        raw = self.ctrl.cpu()
        f = raw.view(torch.float32)
        return dict(grad_norm=math.sqrt(max(float(f[0]), 0.0)), clip_coef=float(f[1]), nan_skips=int(raw[2]),
                    steps=int(raw[3]), loss_sum=float(f[4]), last_loss=float(f[5]))


def train_epoch(epoch, args, model, train_dataloader, device, n_gpu, optimizer, scheduler, global_step, scaler=None,
                local_rank=0, tail=None):
    """Signature and return value of the reference's train_epoch.  `tail` may be passed to keep one TrainTail (and
    its NaN-skip counter) across epochs."""
    model.train()
    log_step = getattr(args, "n_display", 100)
    acc = max(int(getattr(args, "gradient_accumulation_steps", 1)), 1)
    if tail is None:
        # one TrainTail (device control block with the NaN-skip counter) per optimizer, kept across epochs: the
        # optimizer's host-side step counts are relative to that counter (AdaptAdamW.effective_step)
        tail = getattr(optimizer, "_segclip_tail", None)
        if tail is None or tail.model is not model:
            tail = TrainTail(model, optimizer, getattr(args, "clip_grad", 1.0))
            optimizer._segclip_tail = tail
    start_sum = tail.read()["loss_sum"]
    partial = None  # losses of the micro-steps that do not end in an optimizer step
    start_time = time.time()
    n_batches = 0
    for step, batch in enumerate(train_dataloader):
        n_batches += 1
        batch = tuple(t.to(device=device, non_blocking=True) for t in batch)
        image_seg = None
        if len(batch) == 6:
            input_ids, input_mask, segment_ids, image, coord, image_seg = batch
        else:
            input_ids, input_mask, segment_ids, image, coord = batch
        loss = model(input_ids, segment_ids, input_mask, image, image_seg=image_seg)
        if n_gpu > 1:
            loss = loss.mean()
        if acc > 1:
            loss = loss / acc
        loss.backward()
        if (step + 1) % acc != 0:
            d = torch.nan_to_num(loss.detach().float(), nan=0.0)
            partial = d if partial is None else partial + d
            continue
        if scheduler is not None:
            scheduler.step()
        tail.run(loss)
        global_step += 1
        if global_step % log_step == 0 and local_rank == 0:
            st = tail.read()
            lrs = "-".join('%.9f' % v for v in sorted(set(optimizer.get_lr(with_grad_only=False))))
            logger.info("Epoch: %d/%s, Step: %d/%d, Lr: %s, Loss: %f, GradNorm: %f, Time/step: %f", epoch + 1,
                        getattr(args, "epochs", "?"), step + 1, len(train_dataloader), lrs, st["last_loss"],
                        st["grad_norm"], (time.time() - start_time) / (log_step * acc))
            start_time = time.time()
    if hasattr(model, "drain"):
        model.drain()   # GradSync: read the cross-rank agreements posted in the last passes (tail.read() below waits for the device anyway)
    total = tail.read()["loss_sum"] - start_sum + (float(partial) if partial is not None else 0.0)
    return total / max(n_batches, 1), global_step
