"""Mirror of modules/modeling.py: SegCLIP(nn.Module) - CLIP towers + learnable-center vision body,
contrastive (all-gather InfoNCE) + superpixel-KL + MAE-reconstruction losses.

Same constructor / from_pretrained / forward signatures, attribute names and state_dict keys as the
reference, so main_task_align.py-style drivers work unchanged."""
from functools import partial

import torch
from torch import nn

from .. import _lib as L
from .. import config, ops
from .module_clip import CLIP, available_models
from .module_mae import MAEDecoder
from .util_module import CrossEn, PreTrainedModel, dist_collect, get_attr, get_logger, show_log


def _detached(obj):
    """Copy of a nest of dicts / lists / tuples with every tensor detached.  The `last_*` attributes are for inspection
    (tests, logging): holding graph-attached tensors on the module would keep the previous step's autograd graph - and
    with it the AccumulateGrad nodes of the stream that step ran on - alive, which breaks hipGraph capture of the step."""
    if isinstance(obj, torch.Tensor):
        return obj.detach()
    if isinstance(obj, dict):
        return {k: _detached(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_detached(v) for v in obj)
    return obj


class SegCLIPPreTrainedModel(PreTrainedModel, nn.Module):
    def __init__(self, *inputs, **kwargs):
        super().__init__()
        self.clip = None

    @classmethod
    def from_pretrained(cls, state_dict=None, cache_dir=None, type_vocab_size=2, *inputs, **kwargs):
        """modules/modeling.py:26-75: CLIP state-dict -> renamed keys (visual resblocks i<first_stage ->
        layers0.i, else layers2.(i-first_stage)) -> construct -> name-based load."""
        task_config = kwargs.get("task_config", None)
        if task_config is not None:
            if not hasattr(task_config, "local_rank"):
                task_config.__dict__["local_rank"] = 0
            elif task_config.local_rank == -1:
                task_config.local_rank = 0
        if state_dict is None:
            state_dict = {}
        pretrained_clip_name = get_attr(task_config, "pretrained_clip_name", default_value="ViT-B/16", donot_log=True)
        if pretrained_clip_name in available_models():
            clip_state_dict = CLIP.get_config(pretrained_clip_name=pretrained_clip_name)
        else:
            clip_state_dict = CLIP.get_config(pretrained_clip_name="ViT-B/32")
        for key in ["input_resolution", "context_length", "vocab_size"]:
            if key in clip_state_dict:
                del clip_state_dict[key]
        first_stage = getattr(task_config, "first_stage_layer", 10) if task_config is not None else 10
        for key, val in clip_state_dict.items():
            new_key = "clip." + key
            if "visual.transformer." in key:
                parts = new_key.split(".")
                n_ = int(parts[4])
                if n_ >= first_stage:
                    parts[3] = "layers2"
                    parts[4] = str(n_ - first_stage)
                else:
                    parts[3] = "layers0"
                new_key = ".".join(parts)
            if new_key not in state_dict:
                state_dict[new_key] = val.clone()
        model = cls(clip_state_dict, *inputs, **kwargs)
        if state_dict is not None:
            model = cls.init_preweight(model, state_dict, task_config=task_config, print_logger=get_logger())
        return model


class SegCLIP(SegCLIPPreTrainedModel):
    def __init__(self, clip_state_dict, task_config):
        super().__init__()
        self.task_config = task_config
        self.ignore_image_index = -1
        pretrained_clip_name = get_attr(task_config, "pretrained_clip_name", default_value="ViT-B/16", donot_log=True)
        assert "visual.proj" in clip_state_dict
        vision_width = clip_state_dict["visual.conv1.weight"].shape[0]
        vision_layers = len([k for k in clip_state_dict.keys()
                             if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
        vision_patch_size = clip_state_dict["visual.conv1.weight"].shape[-1]
        grid_size = round((clip_state_dict["visual.positional_embedding"].shape[0] - 1) ** 0.5)
        image_resolution = vision_patch_size * grid_size
        if pretrained_clip_name not in available_models():
            assert pretrained_clip_name[:5] == "ViT-B"
            vision_patch_size = int(pretrained_clip_name.split("/")[-1])
            assert image_resolution % vision_patch_size == 0
        embed_dim = clip_state_dict["text_projection"].shape[1]
        context_length = clip_state_dict["positional_embedding"].shape[0]
        vocab_size = clip_state_dict["token_embedding.weight"].shape[0]
        transformer_width = clip_state_dict["ln_final.weight"].shape[0]
        transformer_heads = transformer_width // 64
        transformer_layers = len(set(k.split(".")[2] for k in clip_state_dict if k.startswith("transformer.resblocks")))
        for name, val in (("embed_dim", embed_dim), ("image_resolution", image_resolution),
                          ("vision_layers", vision_layers), ("vision_width", vision_width),
                          ("vision_patch_size", vision_patch_size), ("context_length", context_length),
                          ("vocab_size", vocab_size), ("transformer_width", transformer_width),
                          ("transformer_heads", transformer_heads), ("transformer_layers", transformer_layers)):
            show_log(task_config, "\t {}: {}".format(name, val))
        self.first_stage_layer = get_attr(task_config, "first_stage_layer", default_value=10)
        self.clip = CLIP(embed_dim, image_resolution, vision_layers, vision_width, vision_patch_size, context_length,
                         vocab_size, transformer_width, transformer_heads, transformer_layers,
                         first_stage_layer=self.first_stage_layer).float()
        self.loss_fct = CrossEn()
        self.loss_fct_stdce = nn.CrossEntropyLoss()  # attribute kept for parity; the HIP CE kernel is used
        mae_vis_mask_ratio = get_attr(task_config, "mae_vis_mask_ratio", default_value=0.75)
        self.use_vision_mae_recon = get_attr(task_config, "use_vision_mae_recon", default_value=False)
        if self.use_vision_mae_recon:
            self.vis_mask_ratio = mae_vis_mask_ratio
            self.vis_mae_decoder = MAEDecoder(vision_width, vision_width // 2, image_resolution, vision_patch_size,
                                              decoder_depth=3, decoder_num_heads=8, mlp_ratio=4.,
                                              norm_layer=partial(nn.LayerNorm, eps=1e-6))
        mae_seq_mask_ratio = get_attr(task_config, "mae_seq_mask_ratio", default_value=0.15)
        self.use_text_mae_recon = get_attr(task_config, "use_text_mae_recon", default_value=False)
        if self.use_text_mae_recon:   # modules/modeling.py:156-166
            self.seq_mask_ratio = mae_seq_mask_ratio
            self.seq_mae_decoder = MAEDecoder(embed_dim, embed_dim // 2, image_resolution, vision_patch_size,
                                              decoder_depth=3, decoder_num_heads=8, mlp_ratio=4., choice_seq=True,
                                              pred_len=vocab_size, seq_len=self.task_config.max_words)
        self.use_seglabel = get_attr(task_config, "use_seglabel", default_value=False)
        self.apply(self.init_weights)
        # per-model overrides of the run-time switches (segclip_amd/config.py), e.g. {"compute_dtype": torch.bfloat16}
        self.segclip_config = {}

    def scope(self):
        """Context manager applying this model's switch overrides (forward() does this itself)."""
        return config.scope(**self.segclip_config)

    def _text_stream(self):
        from .. import streams
        return streams.side_stream("text")

    def _visual_feature(self, image, image_frame):
        """(pooled visual feature (B, 1, embed) or (B, embed), mid_states) of the training forward"""
        if config.fused_head:
            return self.clip.encode_image_pooled(image)
        visual_output, _hidden, mid_states = self.get_visual_output(image, shaped=True, image_frame=image_frame, return_hidden=True)
        return visual_output, mid_states

    # ------------------------------------------------------------------ forward
    def forward(self, input_ids, token_type_ids, attention_mask, image, image_seg=None):
        """modules/modeling.py:174-256.  token_type_ids / attention_mask are accepted and ignored on this
        path exactly like the reference."""
        if self.segclip_config:
            with config.scope(**self.segclip_config):
                return self._forward(input_ids, token_type_ids, attention_mask, image, image_seg)
        return self._forward(input_ids, token_type_ids, attention_mask, image, image_seg)

    def _forward(self, input_ids, token_type_ids, attention_mask, image, image_seg=None):
        input_ids = input_ids.view(-1, input_ids.shape[-1])
        L.require_cuda(input_ids, torch.as_tensor(image))  # fail loudly: there is no CPU / eager fallback
        image_input = torch.as_tensor(image).float()
        b, pair, channel, h, w = image_input.shape
        image = image_input[:, 0].reshape(b, channel, h, w)
        image_frame = 1
        if not self.training:
            return None
        ops._GradFold.advance()          # config.fold_param_grads: nothing of an earlier (possibly failed) backward pass survives
        if config.compute_dtype == torch.bfloat16:
            ops.refresh_weight_shadows(force=not config.trust_weight_shadows)
        # The two towers are independent until the similarity: the text tower (small GEMMs that leave most CUs
        # idle) is enqueued on a second HIP stream so its kernels fill the gaps of the vision tower; autograd runs
        # each node's backward on the stream of its forward, so the overlap carries over to the backward pass.
        # Host launch order matters as much as the streams: the host needs ~3 ms to enqueue the text tower.  Enqueued before
        # the vision tower, the vision stream (the critical path) idles that long at the start of every step - and again
        # at the END of the backward pass, because autograd replays the recording order backwards (vision backward first,
        # then ~6 ms of text-backward launches during which the vision stream has nothing queued).  So the text tower is
        # enqueued after the first config.text_after_blocks vision blocks: in both passes the vision stream then has
        # several milliseconds of work queued while the host feeds the text stream (tools/stream_gaps.py).
        if config.overlap_towers:
            main = torch.cuda.current_stream()
            side = self._text_stream()
            ready = torch.cuda.Event()
            ready.record(main)           # weight shadows / inputs are ready: the text stream need not wait for vision blocks
            box = {}

            def enqueue_text():
                side.wait_event(ready)
                with torch.cuda.stream(side):
                    box["seq"] = self.clip.encode_text_eot(input_ids).unsqueeze(1)
            k = int(config.text_after_blocks)
            if k > 0:
                config.set_stack_hook(k, enqueue_text)
            else:
                enqueue_text()
            pending = None
            try:
                visual_output, mid_states = self._visual_feature(image, image_frame)
                pending = config.take_stack_hook()      # a vision tower without a fused stack never ran the hook
            finally:
                config.take_stack_hook()                # never leave a stale hook behind an exception
            if pending is not None:
                pending[1]()
            sequence_output = box["seq"]
            main.wait_stream(side)
            sequence_output.record_stream(main)
        else:
            sequence_output = self.clip.encode_text_eot(input_ids).unsqueeze(1)
            visual_output, mid_states = self._visual_feature(image, image_frame)
        self.last_mid_states = _detached(mid_states)
        self._last_head = {}
        if config.fused_head:
            # contrastive head (modules/modeling.py:196-210,338-362) as one autograd node: L2-normalise, one stacked
            # all-gather, both logits matrices in exact fp32, both cross entropies, their mean
            loss = ops.ClipLossFn.apply(visual_output.float().view(b, -1), sequence_output.squeeze(1).float(),
                                        self.clip.logit_scale, int(getattr(self.task_config, "rank", 0)), self._last_head)
        else:
            sim_matrix_t2v, sim_matrix_v2t = self._loose_similarity(sequence_output, visual_output)
            offset = sequence_output.size(0) * int(getattr(self.task_config, "rank", 0))
            sim_loss1 = ops.CrossEntropyFn.apply(sim_matrix_t2v, offset)
            sim_loss2 = ops.CrossEntropyFn.apply(sim_matrix_v2t, offset)
            loss = (sim_loss1 + sim_loss2) / 2.
            self._last_head["logits"] = (sim_matrix_t2v.detach(), sim_matrix_v2t.detach())
        self.last_losses = {"contrastive": loss.detach()}
        if self.use_seglabel:
            image_seg_ = torch.as_tensor(image_seg)[:, 0].reshape(b, -1)
            hard = mid_states["attns"][0]["hard_attn"]
            clutering_loss = ops.SuperpixelKLFn.apply(hard, image_seg_)
            loss = loss + clutering_loss
            self.last_losses["kl"] = clutering_loss.detach()
        if self.use_text_mae_recon:   # modules/modeling.py:226-236
            attention_mask = attention_mask.view(-1, attention_mask.shape[-1])
            _, seq_hidden, seq_mae_mask, seq_mae_ids_restore = self.get_sequence_output(
                input_ids, token_type_ids, attention_mask, shaped=True, return_hidden=True, mask_ratio=self.seq_mask_ratio)
            seq_mae_mask = seq_mae_mask.view(-1, seq_mae_mask.size(-1))
            seq_mae_ids_restore = seq_mae_ids_restore.view(-1, seq_mae_ids_restore.size(-1))
            _mae_mask = (seq_mae_mask + attention_mask).gt(1)
            self.last_text_mae = _detached((seq_mae_mask, seq_mae_ids_restore, seq_hidden))
            seq_mae_loss = self.seq_mae_decoder.forward_seq(input_ids, seq_hidden, _mae_mask, seq_mae_ids_restore,
                                                            attention_mask)
            loss = loss + seq_mae_loss
            self.last_losses["text_mae"] = seq_mae_loss.detach()
        if self.use_vision_mae_recon:
            _, vis_hidden, vis_mae_mask, vis_mae_ids_restore, mid_mae_states = self.get_visual_output(
                image, shaped=True, image_frame=image_frame, return_hidden=True, mask_ratio=self.vis_mask_ratio)
            vis_hidden = mid_mae_states["hidden"]
            vis_hidden = ops.mean_cat(vis_hidden)          # cat([mean over the tokens, tokens]) as one kernel
            vis_mae_mask = vis_mae_mask.view(-1, vis_mae_mask.size(-1))
            vis_mae_ids_restore = vis_mae_ids_restore.view(-1, vis_mae_ids_restore.size(-1))
            self.last_mae = _detached((vis_mae_mask, vis_mae_ids_restore, mid_mae_states))
            vis_mae_loss = self.vis_mae_decoder.forward_vis(image, vis_hidden, vis_mae_mask, vis_mae_ids_restore,
                                                            loss_allpatch=False)
            loss = loss + vis_mae_loss
            self.last_losses["mae"] = vis_mae_loss.detach()
        return loss

    @property
    def last_logits(self):
        """(sim_matrix_t2v, sim_matrix_v2t) of the last training forward (inspection only; computed on demand from the raw
        cosines the fused contrastive head keeps)."""
        h = getattr(self, "_last_head", None)
        if not h:
            return None
        if "logits" in h:
            return h["logits"]
        s = torch.clamp(h["logit_scale"].exp(), max=100)
        return s * h["cos"][0], s * h["cos"][1]

    def get_sequence_output(self, input_ids, token_type_ids, attention_mask, shaped=False, return_hidden=False,
                            seq_model=None, mask_ratio=0.):
        """modules/modeling.py:258-279."""
        if shaped is False:
            input_ids = input_ids.view(-1, input_ids.shape[-1])
        seq_model = seq_model or self.clip
        bs_pair = input_ids.size(0)
        sequence_hidden = seq_model.encode_text(input_ids, return_hidden=return_hidden, mask_ratio=mask_ratio)
        if isinstance(sequence_hidden, tuple):
            if mask_ratio > 0:
                return tuple([itm.float().view(bs_pair, -1, itm.size(-1)) for itm in sequence_hidden[:2]]
                             + [itm.view(bs_pair, -1, itm.size(-1)) for itm in sequence_hidden[2:]])
            return tuple(itm.float().view(bs_pair, -1, itm.size(-1)) for itm in sequence_hidden)
        return sequence_hidden.float().view(bs_pair, -1, sequence_hidden.size(-1))

    def get_visual_output(self, image, shaped=False, image_frame=-1, return_hidden=False, vis_model=None, mask_ratio=0.):
        """modules/modeling.py:281-303."""
        if shaped is False:
            image_input = torch.as_tensor(image).float()
            b, pair, channel, h, w = image_input.shape
            image = image_input[:, 0].reshape(b, channel, h, w)
        vis_model = vis_model or self.clip
        bs_pair = image.size(0)
        visual_hidden = vis_model.encode_image(image, video_frame=image_frame, return_hidden=return_hidden,
                                               mask_ratio=mask_ratio)
        if isinstance(visual_hidden, tuple):
            if mask_ratio > 0:
                return tuple([itm.float().view(bs_pair, -1, itm.size(-1)) for itm in visual_hidden[:2]]
                             + [itm.view(bs_pair, -1, itm.size(-1)) for itm in visual_hidden[2:4]] + [visual_hidden[4]])
            return tuple([itm.float().view(bs_pair, -1, itm.size(-1)) for itm in visual_hidden[:2]]
                         + [visual_hidden[2]])
        return visual_hidden.float().view(bs_pair, -1, visual_hidden.size(-1))

    def get_sequence_visual_output(self, input_ids, token_type_ids, attention_mask, image, shaped=False,
                                   image_frame=-1, return_hidden=False, seq_model=None, vis_model=None):
        """modules/modeling.py:305-320."""
        if shaped is False:
            input_ids = input_ids.view(-1, input_ids.shape[-1])
            image_input = torch.as_tensor(image).float()
            b, pair, channel, h, w = image_input.shape
            image = image_input[:, 0].reshape(b, channel, h, w)
        sequence_output = self.get_sequence_output(input_ids, token_type_ids, attention_mask, shaped=True,
                                                   return_hidden=return_hidden, seq_model=seq_model)
        visual_output = self.get_visual_output(image, shaped=True, image_frame=image_frame,
                                               return_hidden=return_hidden, vis_model=vis_model)
        return sequence_output, visual_output

    # modules/modeling.py:322-336: mean-pooling variants of the similarity inputs.  Dead API in the reference's training path
    # (its forward uses _loose_similarity on the pooled CLS / EOT features); carried for callers that import them.  Plain
    # tensor arithmetic on (B, L, D) / (B, G, D) inputs, any device.
    def _mean_pooling_for_similarity_sequence(self, sequence_output, attention_mask):
        keep = attention_mask.to(dtype=torch.float).unsqueeze(-1).clone()
        keep[:, 0, :] = 0.                                       # the start-of-text token does not take part
        return (sequence_output * keep).sum(dim=1) / keep.sum(dim=1, dtype=torch.float)

    def _mean_pooling_for_similarity_visual(self, visual_output):
        return visual_output.mean(dim=1)

    def _mean_pooling_for_similarity(self, sequence_output, visual_output, attention_mask):
        return (self._mean_pooling_for_similarity_sequence(sequence_output, attention_mask),
                self._mean_pooling_for_similarity_visual(visual_output))

    def _loose_similarity(self, sequence_output, visual_output, logit_scale=None):
        """modules/modeling.py:338-362: L2-normalise, clamp(exp(logit_scale), 100), all-gather both
        embedding matrices over RCCL in ONE fused message (training), two logits GEMMs (exact fp32)."""
        visual_output = ops.L2NormFn.apply(visual_output.squeeze(1).float())
        sequence_output = ops.L2NormFn.apply(sequence_output.squeeze(1).float())
        if logit_scale is not None:
            logit_scale = torch.clamp(logit_scale.exp(), max=100)
        else:
            logit_scale = torch.clamp(self.clip.logit_scale.exp(), max=100)
        if self.training:
            both = dist_collect(torch.stack([visual_output, sequence_output], dim=1), self.task_config)
            visual_collect, sequence_collect = both[:, 0], both[:, 1]
            t2v = ops.bmm(sequence_output.unsqueeze(0), visual_collect.unsqueeze(0), transB=True)[0]
            v2t = ops.bmm(visual_output.unsqueeze(0), sequence_collect.unsqueeze(0), transB=True)[0]
            return logit_scale * t2v, logit_scale * v2t
        t2v = logit_scale * ops.bmm(sequence_output.unsqueeze(0), visual_output.unsqueeze(0), transB=True)[0]
        return t2v, t2v.T

    def get_similarity_logits(self, sequence_output, visual_output, attention_mask, shaped=False):
        t2v, v2t = self._loose_similarity(sequence_output, visual_output)
        return t2v, v2t, ()
