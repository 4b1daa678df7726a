"""Parameter inventories (reference ``state_dict`` key names + shapes) and a deterministic generator.

Key names follow the reference checkpoints so that ``pretrained/argoverse_stage_two.ckpt`` etc. load
unchanged (loader semantics: multi_view_generation/utils/general.py:119-160):

* Route M: ``MaskGit.state_dict()``       - muse_maskgit_pytorch.py:204-261 (TransformerMultiView),
                                            :90-115 (Attention), :78-88 (FeedForward), :388-392 (SelfCritic)
* Route A: ``GPT.state_dict()``           - mingpt_sparse.py:267-308, :157-177, :215-238
* stage 1: ``VQModel.state_dict()``       - stage1/vqgan.py:31-80, stage1/model.py:342-537, stage1/quantize.py:213-245

No pretrained weights, datasets or network exist in the build/bench environment, so benchmarks and
parity fixtures use ``generate_state_dict``: a counter-based (Philox, keyed by seed and parameter name)
generator, so any tensor can be regenerated independently, on any machine, in any order.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Callable, Dict, Iterable, Mapping, Optional, Sequence, Tuple

import numpy as np
import torch

Shape = Tuple[int, ...]


# --------------------------------------------------------------------------------------------
# inventories
# --------------------------------------------------------------------------------------------
def ff_inner_dim(dim: int, mult: float = 4) -> int:
    """muse_net:81 - GEGLU feed-forward inner width, int(dim*mult*2/3) (2730 for dim=1024)."""
    return int(dim * mult * 2 / 3)


def muse_transformer_shapes(cfg, *, depth: int, heads: int, dim_head: int = 64, ff_mult: float = 4, num_tokens: Optional[int] = None) -> "OrderedDict[str, Shape]":
    """Keys of ``MaskGitTransformerMultiView`` (no prefix)."""
    D = cfg.num_embed
    V = num_tokens if num_tokens is not None else cfg.vocab_size
    inner = heads * dim_head
    ffi = ff_inner_dim(D, ff_mult)
    L = cfg.gpt_block_size
    s: "OrderedDict[str, Shape]" = OrderedDict()
    if cfg.bev_embed:
        s["bev_cam_pos_emb"] = (1, cfg.num_cams, cfg.num_cond_tokens, D)
    if cfg.camera_bias:
        s["camera_bias_emb"] = (1, L * (L + 1) // 2)
    if cfg.bev_embed:
        s["bev_grid"] = (3, cfg.bev_latent_res[0], cfg.bev_latent_res[1])
    s["token_emb.weight"] = (V + 1, D)
    s["pos_emb.weight"] = (cfg.num_img_tokens, D)
    s["cond_token_emb.weight"] = (cfg.cond_vocab_size, D)
    s["cond_pos_emb.weight"] = (cfg.num_cond_tokens, D)
    for i in range(depth):
        for j in (0, 1):  # 0 = self attention, 1 = cross attention
            p = f"transformer_blocks.layers.{i}.{j}."
            s[p + "null_kv"] = (2, heads, 1, dim_head)
            s[p + "q_scale"] = (dim_head,)
            s[p + "k_scale"] = (dim_head,)
            s[p + "norm.gamma"] = (D,)
            s[p + "norm.beta"] = (D,)
            s[p + "to_q.weight"] = (inner, D)
            s[p + "to_kv.weight"] = (2 * inner, D)
            s[p + "to_out.weight"] = (D, inner)
        p = f"transformer_blocks.layers.{i}.2."
        s[p + "0.gamma"] = (D,)
        s[p + "0.beta"] = (D,)
        s[p + "1.weight"] = (2 * ffi, D)
        s[p + "3.gamma"] = (ffi,)
        s[p + "3.beta"] = (ffi,)
        s[p + "4.weight"] = (D, ffi)
    s["transformer_blocks.norm.gamma"] = (D,)
    s["transformer_blocks.norm.beta"] = (D,)
    s["norm.gamma"] = (D,)  # unused at inference (muse_net:233), present in checkpoints
    s["norm.beta"] = (D,)
    s["to_logits.weight"] = (V, D)
    s["self_cond_to_init_embed.0.gamma"] = (D,)  # unused (self_cond=False)
    s["self_cond_to_init_embed.0.beta"] = (D,)
    s["self_cond_to_init_embed.1.weight"] = (2 * ffi, D)
    s["self_cond_to_init_embed.3.gamma"] = (ffi,)
    s["self_cond_to_init_embed.3.beta"] = (ffi,)
    s["self_cond_to_init_embed.4.weight"] = (D, ffi)
    if cfg.image_embed:
        s["img_embed.weight"] = (D, 4, 1, 1)
        s["cam_embed.weight"] = (D, 4, 1, 1)
    if cfg.bev_embed:
        s["bev_embed.weight"] = (D, 2, 1, 1)
        s["bev_embed.bias"] = (D,)
    return s


def maskgit_shapes(cfg, **kw) -> "OrderedDict[str, Shape]":
    """Keys of ``MaskGit(self_token_critic=True)``: ``transformer.*``, the aliased ``token_critic.net.*`` and ``token_critic.to_pred.*``."""
    t = muse_transformer_shapes(cfg, **kw)
    s: "OrderedDict[str, Shape]" = OrderedDict()
    for k, v in t.items():
        s["transformer." + k] = v
    for k, v in t.items():
        s["token_critic.net." + k] = v
    s["token_critic.to_pred.weight"] = (1, cfg.num_embed)
    s["token_critic.to_pred.bias"] = (1,)
    return s


def gpt_shapes(cfg) -> "OrderedDict[str, Shape]":
    """Keys of Route-A ``GPT`` (mingpt_sparse.py:267-308)."""
    D, V, L = cfg.num_embed, cfg.vocab_size, cfg.gpt_block_size
    H = cfg.num_heads
    nb = L // cfg.sparse_block_size
    s: "OrderedDict[str, Shape]" = OrderedDict()
    s["x_pos_emb"] = (1, cfg.num_img_tokens, D)
    s["cond_pos_emb"] = (1, cfg.num_cond_tokens, D)
    if cfg.bev_embed:
        s["bev_cam_pos_emb"] = (1, cfg.num_cams, cfg.num_cond_tokens, D)
    if cfg.camera_bias:
        s["camera_bias_emb"] = (1, L * (L + 1) // 2)
    if cfg.bev_embed:
        s["bev_grid"] = (3, cfg.bev_latent_res[0], cfg.bev_latent_res[1])
    s["x_tok_emb.weight"] = (V + 1, D)
    s["cond_tok_emb.weight"] = (cfg.cond_vocab_size, D)
    for i in range(cfg.num_layers):
        p = f"blocks.{i}."
        for ln in ("ln1", "ln2"):
            s[p + ln + ".weight"] = (D,)
            s[p + ln + ".bias"] = (D,)
        for n in ("query", "key", "value"):
            s[p + f"attention.{n}.weight"] = (cfg.hidden_size, cfg.hidden_size)
            s[p + f"attention.{n}.bias"] = (cfg.hidden_size,)
        s[p + "attention.sparse_self_attention.master_layout"] = (H, nb, nb)
        s[p + "mlp.0.weight"] = (4 * D, D)
        s[p + "mlp.0.bias"] = (4 * D,)
        s[p + "mlp.2.weight"] = (D, 4 * D)
        s[p + "mlp.2.bias"] = (D,)
    s["ln_f.weight"] = (D,)
    s["ln_f.bias"] = (D,)
    s["head.weight"] = (V, D)
    if cfg.image_embed:
        s["img_embed.weight"] = (D, 4, 1, 1)
        s["cam_embed.weight"] = (D, 4, 1, 1)
    if cfg.bev_embed:
        s["bev_embed.weight"] = (D, 2, 1, 1)
        s["bev_embed.bias"] = (D,)
    return s


def _resnet_shapes(s, p, cin, cout):
    s[p + "norm1.weight"] = (cin,)
    s[p + "norm1.bias"] = (cin,)
    s[p + "conv1.weight"] = (cout, cin, 3, 3)
    s[p + "conv1.bias"] = (cout,)
    s[p + "norm2.weight"] = (cout,)
    s[p + "norm2.bias"] = (cout,)
    s[p + "conv2.weight"] = (cout, cout, 3, 3)
    s[p + "conv2.bias"] = (cout,)
    if cin != cout:
        s[p + "nin_shortcut.weight"] = (cout, cin, 1, 1)
        s[p + "nin_shortcut.bias"] = (cout,)


def _attn_shapes(s, p, c):
    s[p + "norm.weight"] = (c,)
    s[p + "norm.bias"] = (c,)
    for n in ("q", "k", "v", "proj_out"):
        s[p + n + ".weight"] = (c, c, 1, 1)
        s[p + n + ".bias"] = (c,)


def vq_decoder_shapes(dd: Mapping) -> "OrderedDict[str, Shape]":
    """Keys of ``Decoder`` (stage1/model.py:436-504), prefix ``decoder.``."""
    ch, ch_mult, nrb = dd["ch"], list(dd["ch_mult"]), dd["num_res_blocks"]
    nres = len(ch_mult)
    res = dd["resolution"] // 2 ** (nres - 1)
    attn_res = set(dd["attn_resolutions"])
    s: "OrderedDict[str, Shape]" = OrderedDict()
    block_in = ch * ch_mult[-1]
    s["decoder.conv_in.weight"] = (block_in, dd["z_channels"], 3, 3)
    s["decoder.conv_in.bias"] = (block_in,)
    _resnet_shapes(s, "decoder.mid.block_1.", block_in, block_in)
    _attn_shapes(s, "decoder.mid.attn_1.", block_in)
    _resnet_shapes(s, "decoder.mid.block_2.", block_in, block_in)
    ups = {}
    for lvl in reversed(range(nres)):
        t: "OrderedDict[str, Shape]" = OrderedDict()
        block_out = ch * ch_mult[lvl]
        for b in range(nrb + 1):
            _resnet_shapes(t, f"decoder.up.{lvl}.block.{b}.", block_in, block_out)
            block_in = block_out
            if res in attn_res:
                _attn_shapes(t, f"decoder.up.{lvl}.attn.{b}.", block_in)
        if lvl != 0:
            t[f"decoder.up.{lvl}.upsample.conv.weight"] = (block_in, block_in, 3, 3)
            t[f"decoder.up.{lvl}.upsample.conv.bias"] = (block_in,)
            res *= 2
        ups[lvl] = t
    for lvl in range(nres):  # state_dict order is up.0 .. up.n
        s.update(ups[lvl])
    s["decoder.norm_out.weight"] = (block_in,)
    s["decoder.norm_out.bias"] = (block_in,)
    s["decoder.conv_out.weight"] = (dd["out_ch"], block_in, 3, 3)
    s["decoder.conv_out.bias"] = (dd["out_ch"],)
    return s


def vq_encoder_shapes(dd: Mapping) -> "OrderedDict[str, Shape]":
    """Keys of ``Encoder`` (stage1/model.py:342-403), prefix ``encoder.``."""
    ch, ch_mult, nrb = dd["ch"], list(dd["ch_mult"]), dd["num_res_blocks"]
    nres = len(ch_mult)
    res = dd["resolution"]
    attn_res = set(dd["attn_resolutions"])
    in_mult = [1] + ch_mult
    s: "OrderedDict[str, Shape]" = OrderedDict()
    s["encoder.conv_in.weight"] = (ch, dd["in_channels"], 3, 3)
    s["encoder.conv_in.bias"] = (ch,)
    block_in = ch
    for lvl in range(nres):
        block_in = ch * in_mult[lvl]
        block_out = ch * ch_mult[lvl]
        for b in range(nrb):
            _resnet_shapes(s, f"encoder.down.{lvl}.block.{b}.", block_in, block_out)
            block_in = block_out
            if res in attn_res:
                _attn_shapes(s, f"encoder.down.{lvl}.attn.{b}.", block_in)
        if lvl != nres - 1:
            s[f"encoder.down.{lvl}.downsample.conv.weight"] = (block_in, block_in, 3, 3)
            s[f"encoder.down.{lvl}.downsample.conv.bias"] = (block_in,)
            res //= 2
    _resnet_shapes(s, "encoder.mid.block_1.", block_in, block_in)
    _attn_shapes(s, "encoder.mid.attn_1.", block_in)
    _resnet_shapes(s, "encoder.mid.block_2.", block_in, block_in)
    s["encoder.norm_out.weight"] = (block_in,)
    s["encoder.norm_out.bias"] = (block_in,)
    zc = 2 * dd["z_channels"] if dd.get("double_z", True) else dd["z_channels"]
    s["encoder.conv_out.weight"] = (zc, block_in, 3, 3)
    s["encoder.conv_out.bias"] = (zc,)
    return s


def vqmodel_shapes(dd: Mapping, n_embed: int, embed_dim: int, *, with_encoder: bool = True) -> "OrderedDict[str, Shape]":
    """Keys of ``VQModel`` (stage1/vqgan.py:31-80): encoder.*, decoder.*, quantize.embedding, quant_conv, post_quant_conv."""
    s: "OrderedDict[str, Shape]" = OrderedDict()
    if with_encoder:
        s.update(vq_encoder_shapes(dd))
    s.update(vq_decoder_shapes(dd))
    s["quantize.embedding.weight"] = (n_embed, embed_dim)
    if with_encoder:
        s["quant_conv.weight"] = (embed_dim, dd["z_channels"], 1, 1)
        s["quant_conv.bias"] = (embed_dim,)
    s["post_quant_conv.weight"] = (dd["z_channels"], embed_dim, 1, 1)
    s["post_quant_conv.bias"] = (dd["z_channels"],)
    return s


# --------------------------------------------------------------------------------------------
# deterministic generator
# --------------------------------------------------------------------------------------------
def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.Generator(np.random.Philox(key=[int(seed) & 0xFFFFFFFFFFFFFFFF, zlib.crc32(name.encode())]))


def _fan_in(shape: Shape) -> int:
    return int(np.prod(shape[1:])) if len(shape) > 1 else int(shape[0])


def generate_tensor(name: str, shape: Shape, seed: int, *, logit_gain: float = 8.0) -> torch.Tensor:
    """One synthetic parameter.  Distribution by role (chosen so every term of the path is exercised:
    non-zero biases / positional / camera-bias tables, non-unit norm gains, O(1) activations):

    * norm gains (``gamma``, ``ln*.weight``, ``norm*.weight``)      1 + 0.1 N(0,1)
    * norm / linear / conv biases                                   0.05 N(0,1)      (``beta`` buffers stay 0, muse_net:66)
    * ``q_scale`` / ``k_scale``                                     1 + 0.1 N(0,1)
    * ``null_kv``                                                   N(0,1)           (muse_net:108)
    * ``camera_bias_emb``                                           0.1 N(0,1)
    * positional tables (``*pos_emb*``)                             0.02 N(0,1)
    * token / codebook embeddings                                   N(0,1) (codebook) / 0.5 N(0,1)
    * output heads (``head``, ``to_logits``)                        logit_gain / sqrt(fan_in) N(0,1)  -> well separated logits
    * every other matrix / conv kernel                              N(0,1) / sqrt(fan_in)
    """
    g = _rng(seed, name)
    leaf = name.rsplit(".", 1)[-1]
    n = lambda: g.standard_normal(shape, dtype=np.float32)
    if leaf == "beta":
        a = np.zeros(shape, np.float32)
    elif leaf == "master_layout" or name.endswith("bev_grid"):
        raise KeyError(name)  # tables, not parameters: filled from the config
    elif leaf == "gamma" or (leaf == "weight" and len(shape) == 1):
        a = 1.0 + 0.1 * n()
    elif leaf == "bias":
        a = 0.05 * n()
    elif leaf in ("q_scale", "k_scale"):
        a = 1.0 + 0.1 * n()
    elif leaf == "null_kv":
        a = n()
    elif leaf == "camera_bias_emb":
        a = 0.1 * n()
    elif "pos_emb" in name:
        a = 0.02 * n()
    elif name.endswith("quantize.embedding.weight"):
        a = n()
    elif "tok_emb" in name or "token_emb" in name:
        a = 0.5 * n()
    elif name.endswith("head.weight") or name.endswith("to_logits.weight"):
        a = (logit_gain / np.sqrt(_fan_in(shape))) * n()
    else:
        a = n() / np.sqrt(_fan_in(shape))
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


def generate_state_dict(shapes: Mapping[str, Shape], seed: int, *, tables: Optional[Mapping[str, torch.Tensor]] = None,
                        alias: Optional[Callable[[str], str]] = None, **kw) -> "OrderedDict[str, torch.Tensor]":
    """Materialise ``shapes``.  ``tables`` supplies non-parameter buffers (``bev_grid``, ``master_layout``);
    ``alias`` maps a key to the key whose values it shares (``token_critic.net.X`` -> ``transformer.X``)."""
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape in shapes.items():
        src = alias(name) if alias else name
        if src != name and src in out:
            out[name] = out[src]
            continue
        leaf = src.rsplit(".", 1)[-1]
        if leaf in ("master_layout", "bev_grid"):
            if tables is None or leaf not in tables:
                raise KeyError(f"{name}: table '{leaf}' must be supplied")
            out[name] = torch.as_tensor(tables[leaf]).clone()
            continue
        out[name] = generate_tensor(src, tuple(shape), seed, **kw)
    return out


def maskgit_alias(name: str) -> str:
    return "transformer." + name[len("token_critic.net.") :] if name.startswith("token_critic.net.") else name


def strip_prefixes(sd: Mapping[str, torch.Tensor], ignore_keys: Sequence[str] = ()) -> "OrderedDict[str, torch.Tensor]":
    """Checkpoint key normalisation of ``init_from_ckpt`` (utils/general.py:129-140): unwrap ``state_dict``,
    drop the DeepSpeed ``_forward_module.`` prefix, delete keys containing any of ``ignore_keys``."""
    if "state_dict" in sd:
        sd = sd["state_dict"]
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for k, v in sd.items():
        if k.startswith("_forward_module"):
            k = k.replace("_forward_module.", "")
        if any(ik in k for ik in ignore_keys):
            continue
        out[k] = v
    return out


# --------------------------------------------------------------------------------------------
# whole-model synthetic state dicts (benchmarks, smoke, parity cases)
# --------------------------------------------------------------------------------------------
def maskgit_state_dict(cfg, seed: int, dim_head: int = 64, ff_mult: float = 4) -> "OrderedDict[str, torch.Tensor]":
    from . import tables

    shapes = maskgit_shapes(cfg, depth=cfg.num_layers, heads=cfg.num_heads, dim_head=dim_head, ff_mult=ff_mult, num_tokens=cfg.vocab_size)
    return generate_state_dict(shapes, seed, tables={"bev_grid": tables.get_bev_grid(cfg)}, alias=maskgit_alias)


def gpt_state_dict(cfg, seed: int) -> "OrderedDict[str, torch.Tensor]":
    from . import tables

    return generate_state_dict(gpt_shapes(cfg), seed, tables={"bev_grid": tables.get_bev_grid(cfg), "master_layout": cfg.layout})


def vq_state_dict(dd: Mapping, n_embed: int, embed_dim: int, seed: int, with_encoder: bool = False) -> "OrderedDict[str, torch.Tensor]":
    return generate_state_dict(vqmodel_shapes(dd, n_embed, embed_dim, with_encoder=with_encoder), seed)
