"""CPU oracle: a plain PyTorch-CPU fp32 restatement of the reference's stage-2 sampling path.

TEST INFRASTRUCTURE.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; the product package ``bevgen_amd`` never does (its ops fail loudly when
the HIP library is missing).  Parity status: PINNED - every function below is checked in the build
container against the *imported* reference while oracle/ref_import/make_golden.py generates the golden vectors (it asserts reference ==
restatement on every case before writing tests/golden/*.npz), and tests/test_oracle_golden.py re-checks it against those vectors on every run.
Route A's attention is pinned against the dense restatement of DeepSpeed's block-sparse kernels, not against
DeepSpeed 0.7.4 itself (its Triton kernels are not in the tree and not installable here: "parity unpinned"
at that one boundary).

Everything is functional: ``sd`` is a mapping name -> fp32 tensor with the reference's ``state_dict`` names,
``cfg`` is any object with the ``GPTConfig`` attributes (sizes + tables).  File:line citations are into
/root/reference/multi_view_generation/.
"""
from __future__ import annotations

import math
from typing import Dict, List, Mapping, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ================================================================================================
# shared embedding pieces (modules/transformer/mingpt_sparse.py:331-358 == stage2/muse_maskgit_pytorch.py:309-340)
# ================================================================================================
def image_plane(cfg) -> Tensor:
    """[3, h*w] pixel plane (x*cam_res[0], y*cam_res[1], 1) - gpt:256-264, 288-292."""
    h, w = cfg.cam_latent_h, cfg.cam_latent_w
    xs = torch.linspace(0, 1, w)
    ys = torch.linspace(0, 1, h)
    gx = xs[None, :].expand(h, w) * cfg.cam_res[0]
    gy = ys[:, None].expand(h, w) * cfg.cam_res[1]
    return torch.stack([gx, gy, torch.ones(h, w)], 0).reshape(3, h * w)


def camera_embeddings(sd: Mapping[str, Tensor], prefix: str, cfg, I_inv: Tensor, E_inv: Tensor) -> Tuple[Tensor, Tensor]:
    """(img_embed [B,C,T,D], c_embed [B,C,D]) - gpt:336-349 / muse_net:314-327.

    c_embed = cam_embed(E_inv[..., 3]);  d = E_inv @ [I_inv @ pix; 1];  img = normalize(img_embed(d) - c_embed) with +1e-7.
    """
    Wimg = sd[prefix + "img_embed.weight"].reshape(-1, 4)
    Wcam = sd[prefix + "cam_embed.weight"].reshape(-1, 4)
    c_embed = E_inv[..., 3] @ Wcam.t()  # [B,C,D]
    pix = image_plane(cfg)  # [3,T]
    cam = I_inv @ pix  # [B,C,3,T]
    cam = torch.cat([cam, torch.ones_like(cam[..., :1, :])], dim=-2)  # [B,C,4,T]
    d = E_inv @ cam  # [B,C,4,T]
    d_embed = torch.einsum("od,bcdt->bcto", Wimg, d)  # [B,C,T,D]
    img = d_embed - c_embed[:, :, None, :]
    img = img / (img.norm(dim=-1, keepdim=True) + 1e-7)
    return img, c_embed


def bev_embedding(sd: Mapping[str, Tensor], prefix: str, cfg, c_embed: Tensor) -> Tensor:
    """[B,K,D]: bev_embed(grid) - sum_cams(bev_cam_pos_emb + c_embed) - gpt:353-357 / muse_net:334-338."""
    grid = sd[prefix + "bev_grid"][:2].reshape(2, -1).t()  # [K,2]
    W = sd[prefix + "bev_embed.weight"].reshape(-1, 2)
    grid_embed = grid @ W.t() + sd[prefix + "bev_embed.bias"]  # [K,D]
    cam_part = (sd[prefix + "bev_cam_pos_emb"] + c_embed[:, :, None, :]).sum(dim=1)  # [B,K,D]
    return grid_embed[None] - cam_part


def attention_bias(sd: Mapping[str, Tensor], prefix: str, cfg) -> Tensor:
    """[L,L] fp32: tril-scatter(camera_bias_emb) + prob_matrix - gpt:375-380 / muse_net:343-348."""
    L = cfg.gpt_block_size
    idx = torch.tril_indices(L, L)
    m = torch.zeros((L, L), dtype=torch.float32)
    m[idx[0], idx[1]] = sd[prefix + "camera_bias_emb"].reshape(-1)
    return m + cfg.prob_matrix.to(torch.float32)


# ================================================================================================
# Route M (MaskGit) - modules/stage2/muse_maskgit_pytorch.py
# ================================================================================================
def _ln_gamma(x: Tensor, gamma: Tensor) -> Tensor:
    """muse_net:62-69: LayerNorm with learnable gamma, beta fixed at 0, eps 1e-5."""
    return F.layer_norm(x, x.shape[-1:], gamma, None, 1e-5)


def muse_attention(sd, p: str, x: Tensor, context: Optional[Tensor], bias: Optional[Tensor], heads: int, K: int, scale: float = 8.0) -> Tensor:
    """muse_net:117-169.  ``bias`` is the full [L,L] matrix; self uses [K:,K:], cross uses [K:,:K], both left-padded with a 0 column
    for the null key.  q*scale before l2norm is cancelled by the normalisation (kept for fidelity)."""
    B, N, _ = x.shape
    xn = _ln_gamma(x, sd[p + "norm.gamma"])
    kv_in = context if context is not None else xn
    q = F.linear(xn, sd[p + "to_q.weight"]) * scale
    kv = F.linear(kv_in, sd[p + "to_kv.weight"])
    k, v = kv.chunk(2, dim=-1)
    split = lambda t: t.reshape(t.shape[0], t.shape[1], heads, -1).permute(0, 2, 1, 3)
    q, k, v = split(q), split(k), split(v)
    nk, nv = sd[p + "null_kv"]  # [H,1,dh] each
    k = torch.cat([nk[None].expand(B, -1, -1, -1), k], dim=2)
    v = torch.cat([nv[None].expand(B, -1, -1, -1), v], dim=2)
    q = F.normalize(q, dim=-1) * sd[p + "q_scale"]
    k = F.normalize(k, dim=-1) * sd[p + "k_scale"]
    sim = torch.einsum("bhid,bhjd->bhij", q, k) * scale
    if bias is not None:
        b = bias[K:, :K] if context is not None else bias[K:, K:]
        sim = sim + F.pad(b, (1, 0), value=0.0)
    attn = sim.softmax(dim=-1)
    out = torch.einsum("bhij,bhjd->bhid", attn, v)
    out = out.permute(0, 2, 1, 3).reshape(B, N, -1)
    return F.linear(out, sd[p + "to_out.weight"])


def muse_feedforward(sd, p: str, x: Tensor) -> Tensor:
    """muse_net:71-88: LN -> Linear(D, 2*inner) -> gate*gelu(x) (x = first half) -> LN(inner) -> Linear(inner, D)."""
    h = _ln_gamma(x, sd[p + "0.gamma"])
    h = F.linear(h, sd[p + "1.weight"])
    a, gate = h.chunk(2, dim=-1)
    h = gate * F.gelu(a)
    h = _ln_gamma(h, sd[p + "3.gamma"])
    return F.linear(h, sd[p + "4.weight"])


def muse_context(sd, cfg, cond_ids: Tensor, I_inv: Tensor, E_inv: Tensor, prefix: str = "transformer.") -> Tuple[Optional[Tensor], Tensor]:
    """Per-batch constants of TransformerMultiView.forward (muse_net:309-340): (img_embed [B,N,D] or None, context [B,K,D])."""
    B = cond_ids.shape[0]
    img, c_embed = None, None
    if cfg.image_embed:
        img, c_embed = camera_embeddings(sd, prefix, cfg, I_inv, E_inv)
        img = img.reshape(B, cfg.num_img_tokens, cfg.num_embed)
    context = sd[prefix + "cond_token_emb.weight"][cond_ids]
    if cfg.bev_embed:
        context = context + bev_embedding(sd, prefix, cfg, c_embed)
    context = context + sd[prefix + "cond_pos_emb.weight"][None]
    return img, context


def muse_forward(sd, cfg, ids: Tensor, cond_ids: Tensor, I_inv: Tensor, E_inv: Tensor, *, depth: int, heads: int,
                 prefix: str = "transformer.", collect: Optional[Dict[str, Tensor]] = None) -> Tuple[Tensor, Tensor]:
    """TransformerMultiView.forward in eval mode (muse_net:283-371). ids [(B*C),T] -> (logits [(B*C),T,V], embed [(B*C),T,D])."""
    C, T = cfg.num_cams, cfg.num_cam_tokens
    B = ids.shape[0] // C
    K = cfg.num_cond_tokens
    img, context = muse_context(sd, cfg, cond_ids, I_inv, E_inv, prefix)
    x = sd[prefix + "token_emb.weight"][ids.reshape(B, C * T)]
    if img is not None:
        x = x + img  # muse_net:328
    x = x + sd[prefix + "pos_emb.weight"][None, : C * T]  # muse_net:331
    bias = attention_bias(sd, prefix, cfg) if cfg.camera_bias else None
    if collect is not None:
        collect["x0"] = x.clone()
        collect["context"] = context.clone()
    for i in range(depth):
        lp = f"{prefix}transformer_blocks.layers.{i}."
        x = muse_attention(sd, lp + "0.", x, None, bias, heads, K) + x
        x = muse_attention(sd, lp + "1.", x, context, bias, heads, K) + x
        x = muse_feedforward(sd, lp + "2.", x) + x
        if collect is not None:
            collect[f"x{i + 1}"] = x.clone()
    embed = _ln_gamma(x, sd[prefix + "transformer_blocks.norm.gamma"])
    logits = F.linear(embed, sd[prefix + "to_logits.weight"])
    return logits.reshape(B * C, T, -1), embed.reshape(B * C, T, -1)


def mask_schedule(timesteps: int, seq_len: int) -> List[int]:
    """muse_net:564-567: n_t = max(int(cos(pi/2 * t) * T), 1) for t in linspace(0,1,timesteps), evaluated in fp32 like the reference."""
    out = []
    for t in torch.linspace(0, 1, timesteps):
        out.append(max(int((torch.cos(t * math.pi * 0.5) * seq_len).item()), 1))
    return out


def topk_filter(logits: Tensor, thres: float) -> Tensor:
    """muse_net:453-458: keep the k = ceil((1-thres)*V) largest logits per row, others -inf."""
    k = math.ceil((1 - thres) * logits.shape[-1])
    val, ind = logits.topk(k, dim=-1)
    out = torch.full_like(logits, float("-inf"))
    out.scatter_(-1, ind, val)
    return out


def gumbel_from_uniform(u: Tensor) -> Tensor:
    """muse_net:443-448: -log(-log(u)) with both logs clamped at 1e-20."""
    lg = lambda t: torch.log(t.clamp(min=1e-20))
    return -lg(-lg(u))


def maskgit_generate(sd, cfg, cond_ids: Tensor, I_inv: Tensor, E_inv: Tensor, *, depth: int, heads: int, timesteps: int = 18,
                     temperature: float = 1.0, topk_filter_thres: float = 0.9, critic_noise_scale: float = 1.0,
                     noise: Optional[Mapping[str, Tensor]] = None, init_ids: Optional[Tensor] = None,
                     redundant_forwards: bool = False, trace: Optional[List[Dict[str, Tensor]]] = None,
                     use_token_critic: bool = True, can_remask_prev_masked: bool = False) -> Tensor:
    """MaskGit.generate (muse_net:511-627) -> ids [(B*C), h, w]: with the self token critic, or (``use_token_critic=False`` = the reference's
    ``force_not_use_token_critic`` / a model without critic, muse_net:611-622) with scores = 1 - softmax(logits)[pred].

    ``noise``: {'gumbel_u': [timesteps,(B*C),T,V], 'critic_u': [timesteps,(B*C),T]} uniforms in [0,1) replacing the
    reference's ``uniform_`` draws (torch RNG streams cannot be matched across devices); ``None`` = the deterministic
    ("greedy") setting used for goldens: gumbel noise 0, critic uniform 0.5.
    ``redundant_forwards=True`` additionally executes the classifier-free-guidance "null" forwards exactly like the
    reference (muse_net:272-276, 394-396); in eval mode they are bit-identical to the conditional forward
    (``cond_drop_prob`` is only honoured when training, muse_net:352), so results do not change - only the CPU baseline timing does.
    """
    C, T = cfg.num_cams, cfg.num_cam_tokens
    B = cond_ids.shape[0]
    mask_id = cfg.vocab_size
    ids = torch.full((B * C, T), mask_id, dtype=torch.long)
    scores = torch.zeros((B * C, T), dtype=torch.float32)
    init_mask = None if init_ids is None else (init_ids != mask_id)
    sched = mask_schedule(timesteps, T)
    fwd = lambda x: muse_forward(sd, cfg, x, cond_ids, I_inv, E_inv, depth=depth, heads=heads)
    for step, n_mask in enumerate(sched):
        steps_until_x0 = timesteps - 1 - step
        masked = scores.topk(n_mask, dim=-1).indices
        ids = ids.scatter(1, masked, mask_id)
        if init_ids is not None:
            ids[init_mask] = init_ids[init_mask]
        logits, _ = fwd(ids)
        if redundant_forwards:
            null_logits, _ = fwd(ids)
            logits = null_logits + (logits - null_logits) * 3.0
        filtered = topk_filter(logits, topk_filter_thres)
        temp = temperature * (steps_until_x0 / timesteps)
        g = torch.zeros_like(filtered) if noise is None else gumbel_from_uniform(noise["gumbel_u"][step])
        pred = (filtered / max(temp, 1e-10) + g).argmax(dim=-1)
        is_mask = ids == mask_id
        ids = torch.where(is_mask, pred, ids)
        if use_token_critic:
            _, embed = fwd(ids)
            if redundant_forwards:
                fwd(ids)
            crit = F.linear(embed, sd["token_critic.to_pred.weight"], sd["token_critic.to_pred.bias"])[..., 0]
            u = torch.full_like(crit, 0.5) if noise is None else noise["critic_u"][step]
            scores = crit + (u - 0.5) * critic_noise_scale * (steps_until_x0 / timesteps)
        else:   # muse_net:611-622
            scores = 1 - logits.softmax(dim=-1).gather(2, pred[..., None])[..., 0]
            if not can_remask_prev_masked:
                scores = scores.masked_fill(~is_mask, -1e5)
        if trace is not None:
            trace.append({"ids": ids.clone(), "scores": scores.clone(), "logits": logits.clone()})
    return ids.reshape(B * C, cfg.cam_latent_h, cfg.cam_latent_w)


# ================================================================================================
# Route A (autoregressive, sparse causal + camera bias) - mingpt_sparse.py, sparse_self_attention.py
# ================================================================================================
def sparse_self_attention_dense(q: Tensor, k: Tensor, v: Tensor, layout: Tensor, block: int, attn_mask: Tensor,
                                add_mask: Optional[Tensor]) -> Tensor:
    """Dense restatement of SparseSelfAttention.forward (ssa:103-177) with DeepSpeed 0.7.4's sdd/softmax/dsd semantics:
    ``softmax_rows(dh^-0.5 * (Q K^T + add_mask) + M) V`` with M = 0 where (layout block present and attn_mask != 0) else -inf.
    q,k,v [B,H,L,dh]; layout [H,L/blk,L/blk]; attn_mask [L,L]; add_mask [1,L,L] or [L,L]."""
    dh = q.shape[-1]
    s = torch.einsum("bhid,bhjd->bhij", q, k)
    if add_mask is not None:
        s = s + add_mask.reshape(1, 1, *add_mask.shape[-2:])
    s = s * (float(dh) ** -0.5)
    present = torch.kron(layout.to(torch.float32), torch.ones(block, block)) > 0  # [H,L,L]
    keep = present[None] & (attn_mask != 0)[None, None]
    s = s.masked_fill(~keep, float("-inf"))
    return torch.einsum("bhij,bhjd->bhid", s.softmax(dim=-1), v)


def gpt_embed(sd, cfg, cam_ids: Tensor, cond_ids: Tensor, I_inv: Tensor, E_inv: Tensor) -> Tensor:
    """GPT.forward up to the block stack (gpt:331-373): [B, L, D] = [cond | image tokens in decode order | pad]."""
    B, C, T = cam_ids.shape
    D = cfg.num_embed
    x = sd["x_tok_emb.weight"][cam_ids]  # [B,C,T,D]
    c_embed = None
    if cfg.image_embed:
        img, c_embed = camera_embeddings(sd, "", cfg, I_inv, E_inv)
        x = x + img
    cond = sd["cond_tok_emb.weight"][cond_ids]
    if cfg.bev_embed:
        cond = cond + bev_embedding(sd, "", cfg, c_embed)
    x = x.reshape(B, C * T, D) + sd["x_pos_emb"][:, : C * T]
    cond = cond + sd["cond_pos_emb"]
    x = x[:, cfg.forward_shuffle_idx]
    seq = torch.cat([cond, x], dim=1)
    if cfg.num_pad_tokens:
        pad = sd["x_tok_emb.weight"][cfg.vocab_size][None, None].expand(B, cfg.num_pad_tokens, D)
        seq = torch.cat([seq, pad], dim=1)
    return seq


def layer_layout(sd, cfg, i: int) -> Tensor:
    """Block layout of layer i: the checkpoint's per-layer buffer (the reference draws one per attention module at construction when density < 1,
    gpt:176, maskgen:217-228, and stores it as ``master_layout``), else the configuration's."""
    return sd.get(f"blocks.{i}.attention.sparse_self_attention.master_layout", cfg.layout)


def _gpt_block(sd, i: int, x: Tensor, cfg, bias: Optional[Tensor], rows: Optional[slice] = None) -> Tensor:
    """Block.forward (gpt:240-253): x = ln1(x); x = x + attn(x); x = x + mlp(ln2(x)).  The residual is taken from ln1(x) (reference quirk)."""
    p = f"blocks.{i}."
    H = cfg.num_heads
    x = F.layer_norm(x, x.shape[-1:], sd[p + "ln1.weight"], sd[p + "ln1.bias"], 1e-5)
    split = lambda t: t.reshape(t.shape[0], t.shape[1], H, -1).permute(0, 2, 1, 3)
    q = split(F.linear(x, sd[p + "attention.query.weight"], sd[p + "attention.query.bias"]))
    k = split(F.linear(x, sd[p + "attention.key.weight"], sd[p + "attention.key.bias"]))
    v = split(F.linear(x, sd[p + "attention.value.weight"], sd[p + "attention.value.bias"]))
    a = sparse_self_attention_dense(q, k, v, layer_layout(sd, cfg, i), cfg.sparse_block_size, cfg.attention_mask, bias)
    a = a.permute(0, 2, 1, 3).reshape(x.shape)
    x = x + a  # no output projection (gpt:203-212)
    h = F.layer_norm(x, x.shape[-1:], sd[p + "ln2.weight"], sd[p + "ln2.bias"], 1e-5)
    h = F.linear(F.gelu(F.linear(h, sd[p + "mlp.0.weight"], sd[p + "mlp.0.bias"])), sd[p + "mlp.2.weight"], sd[p + "mlp.2.bias"])
    return x + h


def gpt_forward(sd, cfg, cam_ids: Tensor, cond_ids: Tensor, I_inv: Tensor, E_inv: Tensor, collect: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """GPT.forward(sampling=True) (gpt:319-391): full L-token forward -> logits [B,N,V] in camera-major order."""
    x = gpt_embed(sd, cfg, cam_ids, cond_ids, I_inv, E_inv)
    bias = attention_bias(sd, "", cfg)[None] if cfg.camera_bias else None
    if collect is not None:
        collect["x0"] = x.clone()
    for i in range(cfg.num_layers):
        x = _gpt_block(sd, i, x, cfg, bias)
        if collect is not None:
            collect[f"x{i + 1}"] = x.clone()
    x = F.layer_norm(x, x.shape[-1:], sd["ln_f.weight"], sd["ln_f.bias"], 1e-5)
    K, N = cfg.num_cond_tokens, cfg.num_img_tokens
    logits = F.linear(x[:, K - 1 : K - 1 + N], sd["head.weight"])
    return logits[:, cfg.backward_shuffle_idx]


def top_k_logits(logits: Tensor, k: int) -> Tensor:
    """ar_lm:138-142: values below the k-th largest become -inf (ties with the k-th value are kept)."""
    v, _ = torch.topk(logits, k)
    out = logits.clone()
    out[out < v[..., [-1]]] = float("-inf")
    return out


def pick_token(logits: Tensor, temperature: float, top_k: Optional[int], u: Optional[Tensor]) -> Tensor:
    """ar_lm:204-217.  Greedy (u None) = topk(softmax, 1); stochastic = inverse-CDF draw with the explicit uniform ``u`` [B]
    (the reference calls torch.multinomial, whose RNG stream cannot be reproduced across devices - the draw is defined here as
    the first index whose cumulative probability exceeds u)."""
    logits = logits / temperature
    if top_k is not None:
        logits = top_k_logits(logits, top_k)
    probs = F.softmax(logits, dim=-1)
    if u is None:
        return probs.argmax(dim=-1)
    cdf = probs.cumsum(dim=-1)
    return (cdf <= u[:, None] * cdf[:, -1:]).sum(dim=-1).clamp(max=probs.shape[-1] - 1)


def ar_sample_full_recompute(sd, cfg, cond_ids: Tensor, I_inv: Tensor, E_inv: Tensor, *, temperature: float = 1.0, top_k: Optional[int] = None,
                             noise_u: Optional[Tensor] = None, steps: Optional[int] = None, logits_out: Optional[List[Tensor]] = None,
                             partial_decoding_idx=None, z_indices: Optional[Tensor] = None) -> Tensor:
    """Net2NetTransformer.sample exactly as the reference runs it (ar_lm:154-227): one full L-token forward per generated token.
    Partial decoding (ar_lm:161-165, 181-182): the cameras in ``partial_decoding_idx`` start from the ground-truth ids ``z_indices`` [B,C,T]
    and their positions are skipped by the loop."""
    B = cond_ids.shape[0]
    C, T = cfg.num_cams, cfg.num_cam_tokens
    x = torch.full((B, C, T), cfg.vocab_size, dtype=torch.long)
    fixed = set()
    if partial_decoding_idx is not None:
        fixed = {int(i) for i in partial_decoding_idx}
        for i in fixed:
            x[:, i, :] = z_indices[:, i]
    n_steps = cfg.num_img_tokens if steps is None else steps
    for s in range(n_steps):
        j = int(cfg.forward_shuffle_idx[s])
        if j // T in fixed:
            continue
        logits = gpt_forward(sd, cfg, x, cond_ids, I_inv, E_inv)[:, j]
        if logits_out is not None:
            logits_out.append(logits.clone())
        ix = pick_token(logits, temperature, top_k, None if noise_u is None else noise_u[s])
        x[:, j // T, j % T] = ix
    return x


class ARCache:
    """Prefill + KV-cache decode: the same arithmetic as ``gpt_forward`` restricted to the one row each step needs.
    Valid because image rows are causal in decode order and cond rows only see cond columns (maskgen:148, 202-206);
    checked bit-for-bit-in-argmax against ``ar_sample_full_recompute`` and the reference's own sampling loop in oracle/ref_import/make_golden.py."""

    def __init__(self, sd, cfg, cond_ids: Tensor, I_inv: Tensor, E_inv: Tensor):
        self.sd, self.cfg = sd, cfg
        self.B = cond_ids.shape[0]
        K, D, H = cfg.num_cond_tokens, cfg.num_embed, cfg.num_heads
        self.bias = attention_bias(sd, "", cfg) if cfg.camera_bias else None
        self.present = [torch.kron(layer_layout(sd, cfg, i).to(torch.float32), torch.ones(cfg.sparse_block_size, cfg.sparse_block_size)) > 0 for i in range(cfg.num_layers)]
        self.img_embed = None
        c_embed = None
        if cfg.image_embed:
            self.img_embed, c_embed = camera_embeddings(sd, "", cfg, I_inv, E_inv)  # [B,C,T,D]
        cond = sd["cond_tok_emb.weight"][cond_ids]
        if cfg.bev_embed:
            cond = cond + bev_embedding(sd, "", cfg, c_embed)
        x = cond + sd["cond_pos_emb"]
        self.k: List[Tensor] = []
        self.v: List[Tensor] = []
        rows = torch.arange(K)
        for i in range(cfg.num_layers):
            x = self._block(i, x, rows, prefill=True)
        self.n = K
        self.last_hidden = x[:, -1:]

    def _block(self, i: int, x: Tensor, rows: Tensor, prefill: bool) -> Tensor:
        sd, cfg = self.sd, self.cfg
        p = f"blocks.{i}."
        H = cfg.num_heads
        dh = cfg.hidden_size // H
        x = F.layer_norm(x, x.shape[-1:], sd[p + "ln1.weight"], sd[p + "ln1.bias"], 1e-5)
        split = lambda t: t.reshape(t.shape[0], t.shape[1], H, -1).permute(0, 2, 1, 3)
        q = split(F.linear(x, sd[p + "attention.query.weight"], sd[p + "attention.query.bias"]))
        k = split(F.linear(x, sd[p + "attention.key.weight"], sd[p + "attention.key.bias"]))
        v = split(F.linear(x, sd[p + "attention.value.weight"], sd[p + "attention.value.bias"]))
        if prefill:
            self.k.append(k)
            self.v.append(v)
        else:
            self.k[i] = torch.cat([self.k[i], k], dim=2)
            self.v[i] = torch.cat([self.v[i], v], dim=2)
        kk, vv = self.k[i], self.v[i]
        n = kk.shape[2]
        s = torch.einsum("bhid,bhjd->bhij", q, kk)
        if self.bias is not None:
            s = s + self.bias[rows][:, :n][None, None]
        s = s * (float(dh) ** -0.5)
        keep = self.present[i][:, rows][:, :, :n] & (cfg.attention_mask[rows][:, :n] != 0)[None]
        s = s.masked_fill(~keep[None], float("-inf"))
        a = torch.einsum("bhij,bhjd->bhid", s.softmax(dim=-1), vv).permute(0, 2, 1, 3).reshape(x.shape)
        x = x + a
        h = F.layer_norm(x, x.shape[-1:], sd[p + "ln2.weight"], sd[p + "ln2.bias"], 1e-5)
        h = F.linear(F.gelu(F.linear(h, sd[p + "mlp.0.weight"], sd[p + "mlp.0.bias"])), sd[p + "mlp.2.weight"], sd[p + "mlp.2.bias"])
        return x + h

    def logits(self) -> Tensor:
        x = F.layer_norm(self.last_hidden, (self.cfg.num_embed,), self.sd["ln_f.weight"], self.sd["ln_f.bias"], 1e-5)
        return F.linear(x, self.sd["head.weight"])[:, 0]

    def append(self, step: int, token: Tensor) -> None:
        """Feed the token decoded at ``step`` (sequence row K+step)."""
        cfg, sd = self.cfg, self.sd
        T = cfg.num_cam_tokens
        j = int(cfg.forward_shuffle_idx[step])
        x = sd["x_tok_emb.weight"][token]  # [B,D]
        if self.img_embed is not None:
            x = x + self.img_embed[:, j // T, j % T]
        x = (x + sd["x_pos_emb"][0, j])[:, None]
        rows = torch.tensor([cfg.num_cond_tokens + step])
        for i in range(cfg.num_layers):
            x = self._block(i, x, rows, prefill=False)
        self.last_hidden = x
        self.n += 1


def ar_sample_cached(sd, cfg, cond_ids: Tensor, I_inv: Tensor, E_inv: Tensor, *, temperature: float = 1.0, top_k: Optional[int] = None,
                     noise_u: Optional[Tensor] = None, steps: Optional[int] = None, logits_out: Optional[List[Tensor]] = None,
                     teacher: Optional[Tensor] = None, forced_ids: Optional[Tensor] = None) -> Tensor:
    """Same result as ``ar_sample_full_recompute`` via prefill + per-token decode.  ``teacher`` [B,C,T] forces the fed-back tokens
    (teacher forcing for per-step logits comparisons); the returned ids are still the model's own picks.
    ``forced_ids`` [steps, B] (decode order, >= 0 = emit that token, < 0 = draw): partial decoding with the KV cache - a fixed position still
    appends its K/V row, it just does not draw (``partial_forced_ids`` builds the array from ``partial_decoding_idx`` + ground-truth ids)."""
    B = cond_ids.shape[0]
    C, T = cfg.num_cams, cfg.num_cam_tokens
    x = torch.full((B, C, T), cfg.vocab_size, dtype=torch.long)
    cache = ARCache(sd, cfg, cond_ids, I_inv, E_inv)
    n_steps = cfg.num_img_tokens if steps is None else steps
    for s in range(n_steps):
        j = int(cfg.forward_shuffle_idx[s])
        logits = cache.logits()
        if logits_out is not None:
            logits_out.append(logits.clone())
        ix = pick_token(logits, temperature, top_k, None if noise_u is None else noise_u[s])
        if forced_ids is not None:
            ix = torch.where(forced_ids[s] >= 0, forced_ids[s], ix)
        x[:, j // T, j % T] = ix
        if s + 1 < n_steps:
            cache.append(s, ix if teacher is None else teacher[:, j // T, j % T])
    return x


def partial_forced_ids(cfg, partial_decoding_idx, z_indices: Tensor, steps: Optional[int] = None) -> Tensor:
    """[steps, B] forced-token array of a partial decode: step s of the decode order visits token j = forward_shuffle_idx[s] = (camera j // T,
    position j % T); cameras in ``partial_decoding_idx`` emit their ground-truth id, the others -1."""
    T = cfg.num_cam_tokens
    n_steps = cfg.num_img_tokens if steps is None else steps
    fixed = {int(i) for i in partial_decoding_idx}
    out = torch.full((n_steps, z_indices.shape[0]), -1, dtype=torch.long)
    for s in range(n_steps):
        j = int(cfg.forward_shuffle_idx[s])
        if j // T in fixed:
            out[s] = z_indices[:, j // T, j % T]
    return out


# ================================================================================================
# stage-1 VQGAN decode - modules/stage1/{quantize,vqgan,model}.py, bev_utils/util.py
# ================================================================================================
def _gn_swish(x: Tensor, w: Tensor, b: Tensor, swish: bool = True) -> Tensor:
    """Normalize (s1model:34-35: GroupNorm(32, eps=1e-6)) followed by nonlinearity (s1model:29-31: x*sigmoid(x))."""
    x = F.group_norm(x, 32, w, b, 1e-6)
    return x * torch.sigmoid(x) if swish else x


def _resnet(sd, p: str, x: Tensor) -> Tensor:
    """ResnetBlock.forward with temb=None, dropout 0 (s1model:117-137)."""
    h = _gn_swish(x, sd[p + "norm1.weight"], sd[p + "norm1.bias"])
    h = F.conv2d(h, sd[p + "conv1.weight"], sd[p + "conv1.bias"], padding=1)
    h = _gn_swish(h, sd[p + "norm2.weight"], sd[p + "norm2.bias"])
    h = F.conv2d(h, sd[p + "conv2.weight"], sd[p + "conv2.bias"], padding=1)
    if (p + "nin_shortcut.weight") in sd:
        x = F.conv2d(x, sd[p + "nin_shortcut.weight"], sd[p + "nin_shortcut.bias"])
    return x + h


def _attn_block(sd, p: str, x: Tensor) -> Tensor:
    """AttnBlock.forward (s1model:168-192): single-head spatial attention, scale c^-0.5."""
    b, c, h, w = x.shape
    hn = _gn_swish(x, sd[p + "norm.weight"], sd[p + "norm.bias"], swish=False)
    q = F.conv2d(hn, sd[p + "q.weight"], sd[p + "q.bias"]).reshape(b, c, h * w).permute(0, 2, 1)
    k = F.conv2d(hn, sd[p + "k.weight"], sd[p + "k.bias"]).reshape(b, c, h * w)
    v = F.conv2d(hn, sd[p + "v.weight"], sd[p + "v.bias"]).reshape(b, c, h * w)
    wgt = torch.bmm(q, k) * (int(c) ** (-0.5))
    wgt = F.softmax(wgt, dim=2).permute(0, 2, 1)
    o = torch.bmm(v, wgt).reshape(b, c, h, w)
    return x + F.conv2d(o, sd[p + "proj_out.weight"], sd[p + "proj_out.bias"])


def vq_decoder(sd, dd: Mapping, z: Tensor, prefix: str = "decoder.", collect: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """Decoder.forward (s1model:506-537)."""
    nres = len(dd["ch_mult"])
    nrb = dd["num_res_blocks"]
    h = F.conv2d(z, sd[prefix + "conv_in.weight"], sd[prefix + "conv_in.bias"], padding=1)
    h = _resnet(sd, prefix + "mid.block_1.", h)
    h = _attn_block(sd, prefix + "mid.attn_1.", h)
    h = _resnet(sd, prefix + "mid.block_2.", h)
    if collect is not None:
        collect["mid"] = h.clone()
    for lvl in reversed(range(nres)):
        for b in range(nrb + 1):
            h = _resnet(sd, f"{prefix}up.{lvl}.block.{b}.", h)
            if f"{prefix}up.{lvl}.attn.{b}.norm.weight" in sd:
                h = _attn_block(sd, f"{prefix}up.{lvl}.attn.{b}.", h)
        if lvl != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = F.conv2d(h, sd[f"{prefix}up.{lvl}.upsample.conv.weight"], sd[f"{prefix}up.{lvl}.upsample.conv.bias"], padding=1)
        if collect is not None:
            collect[f"up{lvl}"] = h.clone()
    h = _gn_swish(h, sd[prefix + "norm_out.weight"], sd[prefix + "norm_out.bias"])
    return F.conv2d(h, sd[prefix + "conv_out.weight"], sd[prefix + "conv_out.bias"], padding=1)


DENORM_MEAN = (0.4265, 0.4489, 0.4769)
DENORM_STD = (0.2053, 0.2206, 0.2578)


def denormalize(x: Tensor) -> Tensor:
    """util.denormalize_tensor (bev_utils/util.py:97-118): x*std + mean per channel, clamp [0,1].  [B,3,H,W]."""
    mean = torch.tensor(DENORM_MEAN).reshape(1, 3, 1, 1)
    std = torch.tensor(DENORM_STD).reshape(1, 3, 1, 1)
    return torch.clamp(x * std + mean, 0, 1)


def vq_decode_ids(sd, dd: Mapping, ids: Tensor, latent_hw: Tuple[int, int], prefix: str = "", denorm: bool = True) -> Tensor:
    """decode_to_img (muse_lm:157-164) + denormalize: ids [(B*C),T] -> [(B*C),out_ch,H,W].
    get_codebook_entry (quant:314-329) -> post_quant_conv (vqgan:118-119) -> Decoder."""
    n = ids.shape[0]
    h, w = latent_hw
    zq = sd[prefix + "quantize.embedding.weight"][ids.reshape(-1)].reshape(n, h, w, -1).permute(0, 3, 1, 2).contiguous()
    q = F.conv2d(zq, sd[prefix + "post_quant_conv.weight"], sd[prefix + "post_quant_conv.bias"])
    x = vq_decoder(sd, dd, q, prefix + "decoder.")
    return denormalize(x) if denorm else x


# ================================================================================================
# stage-1 VQGAN encode (the step before the path) - modules/stage1/{model,vqgan,quantize}.py
# ================================================================================================
def vq_encoder(sd, dd: Mapping, x: Tensor, prefix: str = "encoder.") -> Tensor:
    """Encoder.forward (s1model:405-433); Downsample = F.pad(x,(0,1,0,1)) + conv3x3 stride 2 (s1model:68-72)."""
    nres = len(dd["ch_mult"])
    h = F.conv2d(x, sd[prefix + "conv_in.weight"], sd[prefix + "conv_in.bias"], padding=1)
    for lvl in range(nres):
        for b in range(dd["num_res_blocks"]):
            h = _resnet(sd, f"{prefix}down.{lvl}.block.{b}.", h)
            if f"{prefix}down.{lvl}.attn.{b}.norm.weight" in sd:
                h = _attn_block(sd, f"{prefix}down.{lvl}.attn.{b}.", h)
        if lvl != nres - 1:
            h = F.pad(h, (0, 1, 0, 1), mode="constant", value=0)
            h = F.conv2d(h, sd[f"{prefix}down.{lvl}.downsample.conv.weight"], sd[f"{prefix}down.{lvl}.downsample.conv.bias"], stride=2)
    h = _resnet(sd, prefix + "mid.block_1.", h)
    h = _attn_block(sd, prefix + "mid.attn_1.", h)
    h = _resnet(sd, prefix + "mid.block_2.", h)
    h = _gn_swish(h, sd[prefix + "norm_out.weight"], sd[prefix + "norm_out.bias"])
    return F.conv2d(h, sd[prefix + "conv_out.weight"], sd[prefix + "conv_out.bias"], padding=1)


def vq_encode_ids(sd, dd: Mapping, x: Tensor, prefix: str = "", return_distances: bool = False):
    """VQModel.encode (vqgan:84-116, geometric_embedding=False) -> VectorQuantizer2.forward arg-min (quant:271-285): ids [n, h*w]."""
    h = vq_encoder(sd, dd, x, prefix + "encoder.")
    h = F.conv2d(h, sd[prefix + "quant_conv.weight"], sd[prefix + "quant_conv.bias"])
    z = h.permute(0, 2, 3, 1).contiguous()
    zf = z.view(-1, z.shape[-1])
    E = sd[prefix + "quantize.embedding.weight"]
    d = torch.sum(zf ** 2, dim=1, keepdim=True) + torch.sum(E ** 2, dim=1) - 2 * torch.einsum("bd,dn->bn", zf, E.t())
    ids = torch.argmin(d, dim=1).view(x.shape[0], -1)
    return (ids, d) if return_distances else ids
