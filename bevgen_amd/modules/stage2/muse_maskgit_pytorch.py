"""Drop-in for multi_view_generation/modules/stage2/muse_maskgit_pytorch.py (Route M, MaskGit bidirectional decoder).

Same class names, constructor kwargs, method signatures and ``state_dict`` keys as the reference:
  MaskGitTransformerMultiView   muse_net:204-261, 418-421     (``_target_`` at configs/experiment/muse_stage_two_multi_view.yaml:31)
  SelfCritic                    muse_net:388-396
  MaskGit                       muse_net:467-627              (``_target_`` at muse_stage_two_multi_view.yaml:26)
The modules hold parameters only; ``forward`` / ``generate`` run inside libbevgen_hip (bevgen_muse_forward,
bevgen_maskgit_generate).  Inference only (the surveyed branch has no training, README.md:20).
"""
from __future__ import annotations

import math
from typing import Callable, Iterable, Optional

import torch
import torch.nn as nn

from ... import tables
from ... import weights as W
from ...runtime import Context
from ..options import RuntimeOptionsMixin, pop_runtime_options
from ..params import ParamNode, build_tree, module_device


def cosine_schedule(t):
    return torch.cos(t * math.pi * 0.5)


class TransformerMultiView(nn.Module):
    def __init__(self, *, num_tokens, dim, seq_len, dim_out=None, self_cond=False, add_mask_id=False, cfg=None, depth=None, dim_head=64, heads=8, ff_mult=4, **kwargs):
        super().__init__()
        self.runtime_kwargs = pop_runtime_options(kwargs)   # precision / weights next to this module's `_target_` (muse_net:223 takes **kwargs): read by MaskGit
        if cfg is None:
            raise ValueError("cfg (GPTConfig) is required")
        if self_cond:
            raise NotImplementedError("self conditioning is never enabled by the shipped configs (muse_net:240)")
        if dim_head != 64:
            raise ValueError("libbevgen_hip attention kernels are specialised for dim_head = 64")
        self.cfg = cfg
        self.dim = dim
        self.seq_len = seq_len[0] * seq_len[1] if isinstance(seq_len, Iterable) else seq_len
        self.num_tokens = num_tokens
        self.mask_id = num_tokens if add_mask_id else None
        self.dim_out = dim_out if dim_out is not None else num_tokens
        self.self_cond = False
        self.depth, self.heads, self.dim_head, self.ff_mult = depth, heads, dim_head, ff_mult
        if dim != cfg.num_embed or heads != cfg.num_heads or depth != cfg.num_layers:
            raise ValueError("dim / heads / depth must equal cfg.num_embed / num_heads / num_layers (they are interpolated from cfg in the reference YAML)")
        shapes = W.muse_transformer_shapes(cfg, depth=depth, heads=heads, dim_head=dim_head, ff_mult=ff_mult, num_tokens=num_tokens)
        build_tree(self, shapes)
        if cfg.bev_embed:
            self.bev_grid.copy_(tables.get_bev_grid(cfg))
        self._reset_parameters()

    def _reset_parameters(self):
        # reference defaults: gamma = 1, q/k scale = 1, null_kv ~ randn, Linear/Embedding default inits, zeros for the bias tables
        for name, p in self.named_parameters():
            leaf = name.rsplit(".", 1)[-1]
            if leaf in ("gamma", "q_scale", "k_scale"):
                p.data.fill_(1.0)
            elif leaf == "null_kv":
                p.data.normal_()
            elif leaf in ("bev_cam_pos_emb", "camera_bias_emb") or leaf == "bias":
                p.data.zero_()
            elif "emb.weight" in name:
                p.data.normal_()
            elif p.dim() >= 2:
                fan_in = p[0].numel()
                p.data.uniform_(-1 / math.sqrt(fan_in), 1 / math.sqrt(fan_in))


class MaskGitTransformerMultiView(TransformerMultiView):
    def __init__(self, *args, **kwargs):
        assert "add_mask_id" not in kwargs
        super().__init__(*args, add_mask_id=True, **kwargs)


class SelfCritic(nn.Module):
    def __init__(self, net):
        super().__init__()
        self.net = net
        self.to_pred = ParamNode()
        self.to_pred.register_parameter("weight", nn.Parameter(torch.zeros(1, net.dim), requires_grad=False))
        self.to_pred.register_parameter("bias", nn.Parameter(torch.zeros(1), requires_grad=False))
        nn.init.uniform_(self.to_pred.weight, -1 / math.sqrt(net.dim), 1 / math.sqrt(net.dim))


class MaskGit(RuntimeOptionsMixin, nn.Module):
    def __init__(self, image_size, transformer: MaskGitTransformerMultiView, noise_schedule: Callable = cosine_schedule, token_critic=None,
                 self_token_critic=False, cond_image_size=None, cond_drop_prob=0.5, self_cond_prob=0.9, no_mask_token_prob=0.0, critic_loss_weight=1.0,
                 precision=None, weights=None):
        """muse_net:467-509 + the library's modes (``precision`` / ``weights``; bevgen_amd/modules/options.py): explicit here > given to the transformer >
        $BEVGEN_PRECISION / $BEVGEN_WEIGHTS > f16x3 / f32."""
        super().__init__()
        self._ctx: Optional[Context] = None
        opts = dict(getattr(transformer, "runtime_kwargs", {}))
        opts.update({k: v for k, v in (("precision", precision), ("weights", weights)) if v is not None})
        self._init_runtime_options(opts)
        self.image_size = image_size[0] * image_size[1] if isinstance(image_size, Iterable) else image_size
        self.cond_image_size = cond_image_size
        self.cond_drop_prob = cond_drop_prob
        self.transformer = transformer
        self.self_cond = transformer.self_cond
        self.mask_id = transformer.mask_id
        self.noise_schedule = noise_schedule
        assert not (self_token_critic and token_critic is not None)
        self.token_critic = SelfCritic(transformer) if self_token_critic else token_critic
        if self.token_critic is not None and not isinstance(self.token_critic, SelfCritic):
            raise NotImplementedError("a separate TokenCritic network is not implemented: the shipped configuration uses self_token_critic (or none: confidence scores)")
        self.critic_loss_weight = critic_loss_weight
        self.self_cond_prob = self_cond_prob
        self.no_mask_token_prob = no_mask_token_prob

    # ---------------------------------------------------------------------------------- device context
    def load_state_dict(self, *a, **k):
        out = super().load_state_dict(*a, **k)
        self.invalidate()
        return out

    def invalidate(self):
        """Drop the device copy of the weights (call after modifying parameters in place)."""
        if self._ctx is not None:
            self._ctx.close()
            self._ctx = None

    def context(self) -> Context:
        if self._ctx is None:
            dev = module_device(self)
            if dev.type != "cuda":
                raise RuntimeError("MaskGit must be moved to a ROCm device (model.to('cuda')) before sampling; libbevgen_hip has no CPU path")
            cfg = self.transformer.cfg
            ctx = Context(cfg, route="maskgit", device=dev.index if dev.index is not None else torch.cuda.current_device(), **self.runtime_options("maskgit"))
            sd = self.state_dict()
            ctx.load_state_dict({k: v for k, v in sd.items() if not k.startswith("token_critic.net.")})
            ctx.set_tables(cfg)
            ctx.finalize()
            self._ctx = ctx
        return self._ctx

    # ---------------------------------------------------------------------------------- reference API
    @torch.no_grad()
    def generate(self, init_ids: Optional[torch.Tensor] = None, cond_images: Optional[torch.Tensor] = None, fmap_size=None, temperature=1.0,
                 topk_filter_thres=0.9, can_remask_prev_masked=False, force_not_use_token_critic=False, timesteps=12, cond_scale=3,
                 critic_noise_scale=1, batch=None, noise=None, samples_per_layout=1):
        """muse_net:511-627.  ``noise``: None -> stochastic like the reference (uniforms drawn in the sampler kernels, seeded from torch's generator);
        an int -> that seed; 'greedy' -> gumbel noise 0 / critic uniform 0.5; or {'gumbel_u','critic_u'} explicit uniforms."""
        use_token_critic = self.token_critic is not None and not force_not_use_token_critic   # muse_net:553
        if not use_token_critic and can_remask_prev_masked:
            assert self.no_mask_token_prob > 0., 'without training with some of the non-masked tokens forced to predict, not sure if the logits will be meaningful for these token'   # muse_net:621
        cfg = self.transformer.cfg
        ctx = self.context()
        if fmap_size is not None and tuple(fmap_size) != (cfg.cam_latent_h, cfg.cam_latent_w):
            raise ValueError(f"fmap_size {tuple(fmap_size)} != cam_latent_res {(cfg.cam_latent_h, cfg.cam_latent_w)}")
        B = len(cond_images)
        rows, T, V = B * cfg.num_cams, cfg.num_cam_tokens, cfg.vocab_size
        gu = cu = None
        seed = 0
        if noise is None:
            # stochastic like the reference: the uniforms are drawn inside the sampler kernels (Philox keyed by a seed taken from torch's CPU generator, so
            # torch.manual_seed / pl.seed_everything make a run reproducible); nothing of size [timesteps, rows, T, V] is ever materialised
            seed = int(torch.randint(1, 2 ** 62, (), dtype=torch.int64).item())
        elif isinstance(noise, int):
            seed = noise
        elif isinstance(noise, str) and noise == "greedy":
            pass
        else:
            gu, cu = noise["gumbel_u"], noise["critic_u"]
        return ctx.maskgit_generate(cond_images, batch["intrinsics_inv"], batch["extrinsics_inv"], timesteps=timesteps, temperature=temperature,
                                    topk_filter_thres=topk_filter_thres, critic_noise_scale=critic_noise_scale, gumbel_u=gu, critic_u=cu, init_ids=init_ids,
                                    noise_seed=seed, use_token_critic=use_token_critic, can_remask_prev_masked=can_remask_prev_masked, samples_per_layout=samples_per_layout)

    @torch.no_grad()
    def transformer_forward(self, x, conditioning_token_ids, batch, return_embed=False):
        """TransformerMultiView.forward in eval mode (muse_net:283-371): x [(B*C),T] -> logits [(B*C),T,V] (and embed)."""
        logits, embed = self.context().muse_forward(x, conditioning_token_ids, batch["intrinsics_inv"], batch["extrinsics_inv"])
        return (logits, embed) if return_embed else logits

    def forward(self, *a, **k):
        raise NotImplementedError("training losses are outside the inference path this library accelerates (README.md:20: no training in the released branch)")
