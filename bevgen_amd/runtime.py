"""Thin Python host over the C ABI: one ``Context`` per (device, model).

PyTorch is used for device memory, streams and (later) ``torch.distributed`` only: every tensor handed to the library is a
raw ``data_ptr()``; all arithmetic of the hot path happens inside libbevgen_hip's HIP kernels.
"""
from __future__ import annotations

import ctypes as C
import os
import math
from typing import Dict, Mapping, Optional, Sequence

import numpy as np
import torch

from . import _lib, tables
from ._lib import bevgen_cfg
from .weights import ff_inner_dim


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream(device=None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _req(t: torch.Tensor, dtype, device, name: str) -> torch.Tensor:
    if t.dtype != dtype:
        t = t.to(dtype)
    if t.device != device:
        t = t.to(device)
    return t.contiguous()


def mask_schedule(timesteps: int, seq_len: int):
    """Host logic of MaskGit.generate (muse_net:564-567), evaluated with torch fp32 exactly like the reference:
    n_t = max(int(cos(t*pi/2) * T), 1) for t in linspace(0,1,timesteps)."""
    out = []
    for t in torch.linspace(0, 1, timesteps):
        out.append(max(int((torch.cos(t * math.pi * 0.5) * seq_len).item()), 1))
    return out


def topk_count(thres: float, vocab: int) -> int:
    """muse_net:454: k = ceil((1 - thres) * V)."""
    return math.ceil((1 - thres) * vocab)


class Context:
    """Owns a ``bevgen_ctx`` (weights, KV cache, workspace live on the device inside it)."""

    def __init__(self, cfg=None, *, route: str = "maskgit", vq_ddconfig: Optional[Mapping] = None, vq_n_embed: int = 0, vq_embed_dim: int = 0,
                 device: Optional[int] = None, max_batch: int = 0, precision: Optional[str] = None, kv_cache: Optional[str] = None, decode_path: Optional[str] = None, decode_weights: Optional[str] = None,
                 weights: Optional[str] = None, decode_chains: Optional[int] = None):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError("bevgen_amd needs a ROCm GPU (MI355X / gfx950); there is no CPU path in the product")
        self.device_index = torch.cuda.current_device() if device is None else int(device)
        self.device = torch.device("cuda", self.device_index)
        self.cfg = cfg
        c = bevgen_cfg()
        c.abi_version = _lib.ABI_VERSION
        c.route = _lib.ROUTE_MASKGIT if route == "maskgit" else _lib.ROUTE_AR
        # every mode: explicit argument, else its environment variable, else this low-level class's default (exact fp32 everywhere; the drop-in modules
        # resolve their own defaults - f16x3 products - in bevgen_amd/modules/options.py and always pass explicit values)
        precision = precision or os.environ.get("BEVGEN_PRECISION") or "fp32"
        kv_cache = kv_cache or os.environ.get("BEVGEN_KV_CACHE") or "f32"
        decode_path = decode_path or os.environ.get("BEVGEN_DECODE_PATH") or "auto"
        decode_weights = decode_weights or os.environ.get("BEVGEN_DECODE_WEIGHTS") or "f32"
        for key, val, ok in (("kv_cache", kv_cache, ("f32", "f16")), ("decode_path", decode_path, ("auto", "fused", "split", "per_op")),
                             ("decode_weights", decode_weights, ("f32", "f16"))):
            if val not in ok:
                raise ValueError(f"{key} must be one of {ok}, got {val!r}")
        if precision not in ("fp32", "f16x3"):
            raise ValueError(f"precision must be 'fp32' or 'f16x3', got {precision!r}")
        self.precision = precision
        c.precision = {"fp32": _lib.PRECISION_FP32, "f16x3": _lib.PRECISION_F16X3}[precision]
        c.max_batch = max_batch
        # Route A KV-cache storage: 'f32' (bit-exact tokens) or 'f16' (fp16 storage / fp32 accumulate, BASELINE config 4: half the decode traffic)
        c.kv_cache_dtype = {"f32": _lib.KV_F32, "f16": _lib.KV_F16}[kv_cache]
        self.kv_cache = kv_cache
        # Route A decode step: 'auto' (default: 'split' for one or two sequences per call, else 'fused'), 'fused' (three launches per layer), 'split' (LayerNorm + QKV projection kernel, then the attention-only kernel: four launches)
        # or 'per_op' (one kernel per operator, the A/B reference)
        c.decode_path = {"fused": _lib.DECODE_FUSED, "per_op": _lib.DECODE_PER_OP, "split": _lib.DECODE_SPLIT, "auto": _lib.DECODE_AUTO}[decode_path]
        self.decode_path = decode_path
        # Route A projection weights: 'f32', or 'f16' = the model with fp16-representable q/k/v, MLP and head weights (rounded at finalize), whose decode
        # step streams 2-byte weights
        c.decode_weight_dtype = {"f32": _lib.W_F32, "f16": _lib.W_F16}[decode_weights]
        self.decode_weights = decode_weights
        # Route A fused decode step: number of independent sequence groups enqueued on separate streams (0 = the library's choice; $BEVGEN_DECODE_CHAINS)
        c.decode_chains = int(os.environ.get("BEVGEN_DECODE_CHAINS", "0")) if decode_chains is None else int(decode_chains)
        # weights='f16' (or $BEVGEN_WEIGHTS): the GEMM / convolution matrices are rounded to f16 at finalize - two MFMAs per product instead of three
        weights = weights or os.environ.get("BEVGEN_WEIGHTS") or "f32"
        if weights not in ("f32", "f16"):
            raise ValueError(f"weights must be 'f32' or 'f16', got {weights!r}")
        c.weight_dtype = {"f32": _lib.W_F32, "f16": _lib.W_F16}[weights]
        self.weights = weights
        if cfg is not None:
            c.num_layers, c.num_heads, c.dim = cfg.num_layers, cfg.num_heads, cfg.num_embed
            c.vocab_size, c.cond_vocab_size = cfg.vocab_size, cfg.cond_vocab_size
            c.num_cams, c.cam_latent_h, c.cam_latent_w = cfg.num_cams, cfg.cam_latent_h, cfg.cam_latent_w
            c.num_cond_tokens, c.seq_len, c.sparse_block_size = cfg.num_cond_tokens, cfg.gpt_block_size, cfg.sparse_block_size
            c.image_embed, c.bev_embed, c.camera_bias = int(cfg.image_embed), int(cfg.bev_embed), int(cfg.camera_bias)
            c.ff_inner = ff_inner_dim(cfg.num_embed) if route == "maskgit" else 0
        self.vq_ddconfig = dict(vq_ddconfig) if vq_ddconfig is not None else None
        if vq_ddconfig is not None:
            dd = vq_ddconfig
            mult = list(dd["ch_mult"])
            c.vq_ch, c.vq_num_res_blocks, c.vq_z_channels = dd["ch"], dd["num_res_blocks"], dd["z_channels"]
            c.vq_embed_dim, c.vq_n_embed, c.vq_resolution, c.vq_out_ch = vq_embed_dim, vq_n_embed, dd["resolution"], dd["out_ch"]
            c.vq_num_levels = len(mult)
            for i, m in enumerate(mult):
                c.vq_ch_mult[i] = m
            attn = list(dd["attn_resolutions"])
            c.vq_attn_resolution = attn[0] if attn else 0
            c.vq_in_channels = dd.get("in_channels", 0)
        self._c = c
        self.route = route
        h = C.c_void_p()
        code = self.lib.bevgen_create(C.byref(c), self.device_index, C.byref(h))
        if code != 0:
            raise _lib.BevgenError(code, (self.lib.bevgen_last_error(None) or b"").decode())
        self._h = h
        self._finalized = False

    # ------------------------------------------------------------------------------------------ lifecycle
    def close(self):
        """Destroys the context.  A status word nobody has read yet (asynchronous calls, no synchronize()) is reported here: an explicit close() raises after the
        context is gone; the garbage-collector path (__del__) leaves it to the library's message on stderr."""
        if getattr(self, "_h", None):
            pending = 0
            try:
                if torch.cuda.is_available():
                    torch.cuda.synchronize(self.device)
                    pending = self.status()
            except Exception:
                pending = 0
            self.lib.bevgen_destroy(self._h)
            self._h = None
            if pending:
                raise _lib.BevgenError(_lib.ERR_NUMERIC if pending & 28 else -4, f"context closed with device status word {pending} pending: results of its last calls were invalid "
                                       "(include/bevgen_hip.h BEVGEN_STATUS_*)")

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self.lib.bevgen_destroy(self._h)   # (prints a pending status word to stderr)
                self._h = None
        except Exception:
            pass

    def _check(self, code):
        _lib.check(self._h, code)

    # ------------------------------------------------------------------------------------------ device status word
    def synchronize(self) -> None:
        """Wait for this context's stream and raise ``BevgenError`` (code ``ERR_NUMERIC`` / internal) if a kernel of the calls enqueued so far flagged a non-finite
        logit / pixel, an operand outside the f16 range of the chosen precision, or a failed fused-MLP launch (include/bevgen_hip.h "Device status word").  This is where
        the reference's per-step ``assert (~logits.isfinite()).sum() == 0`` (gpt:383,388, ar_lm:202) lands: once per call instead of once per token."""
        self._check(self.lib.bevgen_synchronize(self._h, self._s()))

    def status(self) -> int:
        """The raw status word as the host sees it now (no synchronisation, nothing cleared): ``_lib.STATUS_*`` bits."""
        w = C.c_uint(0)
        self._check(self.lib.bevgen_status(self._h, C.byref(w)))
        return int(w.value)

    def _done(self, check: bool) -> None:
        """End of a whole-call wrapper.  check=True (the default of every wrapper that hands results back): synchronise and surface the status word NOW, so a caller can
        never read tokens / pixels of a call that flagged itself.  check=False keeps the call asynchronous (pipelined callers, bench.py); the word is then reported by the next
        call on this context at the latest, or by an explicit ``synchronize()``."""
        if check:
            self.synchronize()

    def _s(self):
        """The current torch stream of THIS context's device (not of whatever device is current)."""
        return _stream(self.device)

    def load_tensor(self, name: str, t) -> None:
        a = t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
        if a.dtype == np.float32:
            dt = _lib.DTYPE_F32
        elif a.dtype == np.float64:
            dt = _lib.DTYPE_F64
        elif a.dtype == np.int64:
            dt = _lib.DTYPE_I64
        elif a.dtype == np.uint8:
            dt = _lib.DTYPE_U8
        else:
            a = a.astype(np.float32)
            dt = _lib.DTYPE_F32
        a = np.ascontiguousarray(a)
        shape = (C.c_int64 * max(a.ndim, 1))(*a.shape)
        self._check(self.lib.bevgen_load_tensor(self._h, name.encode(), a.ctypes.data_as(C.c_void_p), dt, a.ndim, shape))
        self._finalized = False

    def load_state_dict(self, sd: Mapping[str, torch.Tensor], prefix: str = "") -> None:
        """Upload every tensor of a reference-named state_dict under ``prefix`` (e.g. 'first_stage_model.')."""
        for k, v in sd.items():
            self.load_tensor(prefix + k, v)

    def set_tables(self, cfg=None) -> None:
        cfg = cfg or self.cfg
        self.load_tensor("table.forward_shuffle_idx", cfg.forward_shuffle_idx.to(torch.int64))
        self.load_tensor("table.attention_mask", cfg.attention_mask.to(torch.float32))
        self.load_tensor("table.layout", cfg.layout.to(torch.int64))
        if cfg.prob_matrix is not None:
            self.load_tensor("table.prob_matrix", cfg.prob_matrix.to(torch.float32))  # the reference casts the (float64, legacy) prior to the activation dtype
        if cfg.image_embed:
            self.load_tensor("table.image_plane", tables.image_plane(cfg).reshape(3, -1))

    def finalize(self) -> None:
        self._check(self.lib.bevgen_finalize(self._h))
        self._finalized = True

    # ------------------------------------------------------------------------------------------ Route M
    def muse_forward(self, ids, cond_ids, I_inv, E_inv, want_logits=True, want_embed=True, check=True):
        cfg = self.cfg
        d = self.device
        ids = _req(ids, torch.int64, d, "ids")
        cond_ids = _req(cond_ids, torch.int64, d, "cond_ids")
        I_inv = _req(I_inv, torch.float32, d, "I_inv")
        E_inv = _req(E_inv, torch.float32, d, "E_inv")
        B = cond_ids.shape[0]
        rows = B * cfg.num_cams
        logits = torch.empty((rows, cfg.num_cam_tokens, cfg.vocab_size), dtype=torch.float32, device=d) if want_logits else None
        embed = torch.empty((rows, cfg.num_cam_tokens, cfg.num_embed), dtype=torch.float32, device=d) if want_embed else None
        self._check(self.lib.bevgen_muse_forward(self._h, _ptr(ids), _ptr(cond_ids), _ptr(I_inv), _ptr(E_inv), B, _ptr(logits), _ptr(embed), self._s()))
        self._done(check)
        return logits, embed

    def maskgit_generate(self, cond_ids, I_inv, E_inv, *, timesteps=18, temperature=1.0, topk_filter_thres=0.9, critic_noise_scale=1.0,
                         gumbel_u=None, critic_u=None, init_ids=None, noise_seed=0, use_token_critic=True, can_remask_prev_masked=False, samples_per_layout=1, check=True):
        """gumbel_u / critic_u: explicit uniforms (parity tests); else noise_seed != 0: the samplers draw them in registers (Philox); else deterministic.
        use_token_critic=False: scores = 1 - softmax(logits)[pred] (muse_net:611-622; no critic forwards), can_remask_prev_masked as the reference's flag.
        samples_per_layout S: consecutive groups of S scenes share their condition (cross-attention K / V built once per layout)."""
        cfg = self.cfg
        d = self.device
        cond_ids = _req(cond_ids, torch.int64, d, "cond_ids")
        I_inv = _req(I_inv, torch.float32, d, "I_inv")
        E_inv = _req(E_inv, torch.float32, d, "E_inv")
        B = cond_ids.shape[0]
        rows = B * cfg.num_cams
        T = cfg.num_cam_tokens
        sched = mask_schedule(timesteps, T)
        sched_c = (C.c_int32 * timesteps)(*sched)
        if gumbel_u is not None:
            gumbel_u = _req(gumbel_u, torch.float32, d, "gumbel_u")
            assert tuple(gumbel_u.shape) == (timesteps, rows, T, cfg.vocab_size), gumbel_u.shape
        if critic_u is not None:
            critic_u = _req(critic_u, torch.float32, d, "critic_u")
            assert tuple(critic_u.shape) == (timesteps, rows, T), critic_u.shape
        if init_ids is not None:
            init_ids = _req(init_ids, torch.int64, d, "init_ids").reshape(rows, T)
        out = torch.empty((rows, T), dtype=torch.int64, device=d)
        score_mode = 0 if use_token_critic else (2 if can_remask_prev_masked else 1)
        self._check(self.lib.bevgen_maskgit_generate_ex(self._h, _ptr(cond_ids), _ptr(I_inv), _ptr(E_inv), B, timesteps, sched_c, float(temperature),
                                                         topk_count(topk_filter_thres, cfg.vocab_size), float(critic_noise_scale), _ptr(gumbel_u), _ptr(critic_u),
                                                         _ptr(init_ids), _ptr(out), C.c_uint64(int(noise_seed) & 0xFFFFFFFFFFFFFFFF), score_mode, int(samples_per_layout), self._s()))
        self._done(check)
        return out.reshape(rows, cfg.cam_latent_h, cfg.cam_latent_w)

    def philox_uniform(self, seed, it, stream_id, n, V=0):
        """The uniforms maskgit_generate(noise_seed=seed) draws at iteration `it` (stream 0: gumbel [rows*V], 1: critic [rows])."""
        out = torch.empty((n,), dtype=torch.float32, device=self.device)
        self._check(self.lib.bevgen_op_philox_uniform(self._h, C.c_uint64(int(seed)), int(it), int(stream_id), int(V), int(n), _ptr(out), self._s()))
        return out

    # ------------------------------------------------------------------------------------------ Route A
    def sparse_self_attention(self, q, k, v, layout, attn_mask, add_mask, block):
        d = self.device
        q, k, v = (_req(t, torch.float32, d, "qkv") for t in (q, k, v))
        B, H, L, dh = q.shape
        assert dh == 64, "head dimension must be 64"
        layout = _req(layout, torch.int64, d, "layout")
        attn_mask = _req(attn_mask.reshape(L, L), torch.float32, d, "attn_mask")
        add = None if add_mask is None else _req(add_mask.reshape(L, L), torch.float32, d, "add_mask")
        out = torch.empty_like(q)
        self._check(self.lib.bevgen_sparse_self_attention(self._h, _ptr(q), _ptr(k), _ptr(v), _ptr(layout), _ptr(attn_mask), _ptr(add), B, H, L, int(block), _ptr(out), self._s()))
        return out

    def ar_prefill(self, cond_ids, I_inv, E_inv):
        d = self.device
        cond_ids = _req(cond_ids, torch.int64, d, "cond_ids")
        I_inv = _req(I_inv, torch.float32, d, "I_inv")
        E_inv = _req(E_inv, torch.float32, d, "E_inv")
        self._ar_B = cond_ids.shape[0]
        self._check(self.lib.bevgen_ar_prefill(self._h, _ptr(cond_ids), _ptr(I_inv), _ptr(E_inv), self._ar_B, self._s()))

    def ar_logits(self, check=False):
        """(step-level calls stay asynchronous by default: the status word surfaces at the next call or at synchronize())"""
        out = torch.empty((self._ar_B, self.cfg.vocab_size), dtype=torch.float32, device=self.device)
        self._check(self.lib.bevgen_ar_logits(self._h, _ptr(out), self._s()))
        self._done(check)
        return out

    def ar_decode_step(self, token):
        token = _req(token, torch.int64, self.device, "token")
        self._check(self.lib.bevgen_ar_decode_step(self._h, _ptr(token), self._s()))

    def ar_sample(self, cond_ids, I_inv, E_inv, *, steps=None, top_k=None, temperature=1.0, greedy=True, noise_u=None, samples_per_layout=1, return_logits=False,
                  forced_ids=None, check=True):
        """Route A sampling with the KV cache.  forced_ids [steps, B] int64 in decode order (>= 0: emit this token, < 0: draw) = partial decoding."""
        cfg = self.cfg
        d = self.device
        cond_ids = _req(cond_ids, torch.int64, d, "cond_ids")
        I_inv = _req(I_inv, torch.float32, d, "I_inv")
        E_inv = _req(E_inv, torch.float32, d, "E_inv")
        B = cond_ids.shape[0]
        steps = cfg.num_img_tokens if steps is None else int(steps)
        if noise_u is not None:
            noise_u = _req(noise_u, torch.float32, d, "noise_u")
            assert tuple(noise_u.shape) == (steps, B)
        out = torch.empty((B, cfg.num_cams, cfg.num_cam_tokens), dtype=torch.int64, device=d)
        logits = torch.empty((steps, B, cfg.vocab_size), dtype=torch.float32, device=d) if return_logits else None
        if forced_ids is not None:
            forced_ids = _req(forced_ids, torch.int64, d, "forced_ids")
            assert tuple(forced_ids.shape) == (steps, B)
        self._check(self.lib.bevgen_ar_sample_forced(self._h, _ptr(cond_ids), _ptr(I_inv), _ptr(E_inv), B, steps, int(top_k or 0), float(temperature), int(bool(greedy)),
                                                      _ptr(noise_u), int(samples_per_layout), _ptr(forced_ids), _ptr(out), _ptr(logits), self._s()))
        self._done(check)
        return (out, logits) if return_logits else out

    # ------------------------------------------------------------------------------------------ stage 1
    def _vq_levels(self):
        return len(self.vq_ddconfig["ch_mult"]) - 1

    def _latent_grid(self, latent_hw, tokens=None):
        """(lat_h, lat_w): explicit, else the square grid of ddconfig.resolution.  The decoder is fully convolutional: nuScenes uses 14 x 25."""
        if latent_hw is None:
            lat = self.vq_ddconfig["resolution"] >> self._vq_levels()
            latent_hw = (lat, lat)
        lh, lw = int(latent_hw[0]), int(latent_hw[1])
        if tokens is not None and tokens != lh * lw:
            raise ValueError(f"{tokens} token ids per image do not fill the {lh} x {lw} latent grid: pass latent_hw=cam_latent_res")
        return lh, lw

    def vq_decode(self, ids, denormalize=True, latent_hw=None, uint8=False, check=True):
        """ids [n, lat_h*lat_w] -> [n, out_ch, H, W] fp32 (raw or denormalised to [0,1]) or, with uint8=True, the round(255 x) storage format."""
        dd = self.vq_ddconfig
        ids = _req(ids, torch.int64, self.device, "ids")
        n = ids.shape[0]
        ids = ids.reshape(n, -1)
        lh, lw = self._latent_grid(latent_hw, ids.shape[1])
        f = 1 << self._vq_levels()
        mode = _lib.VQ_OUT_U8 if uint8 else (_lib.VQ_OUT_DENORM if denormalize else _lib.VQ_OUT_RAW)
        out = torch.empty((n, dd["out_ch"], lh * f, lw * f), dtype=torch.uint8 if uint8 else torch.float32, device=self.device)
        self._check(self.lib.bevgen_vq_decode(self._h, _ptr(ids), n, lh, lw, mode, _ptr(out), self._s()))
        self._done(check)
        return out

    def vq_encode(self, x, check=True):
        """VQModel.encode -> token ids: x [n, in_channels, H, W] fp32 (NCHW; H, W multiples of 2^(levels-1)) -> ids [n, h*w] int64."""
        dd = self.vq_ddconfig
        x = _req(x, torch.float32, self.device, "x")
        n, ch, H, W = x.shape
        f = 1 << self._vq_levels()
        if ch != dd["in_channels"] or H % f or W % f:
            raise ValueError(f"vq_encode: input {tuple(x.shape)} needs {dd['in_channels']} channels and sides divisible by {f}")
        ids = torch.empty((n, (H // f) * (W // f)), dtype=torch.int64, device=self.device)
        self._check(self.lib.bevgen_vq_encode(self._h, _ptr(x), n, H, W, _ptr(ids), self._s()))
        self._done(check)
        return ids

    def vq_decode_latents(self, zq, denormalize=False, check=True):
        """VQModel.decode(quant): zq [n, embed_dim, h, w] fp32."""
        dd = self.vq_ddconfig
        zq = _req(zq, torch.float32, self.device, "zq")
        n, _, lh, lw = zq.shape
        f = 1 << self._vq_levels()
        out = torch.empty((n, dd["out_ch"], lh * f, lw * f), dtype=torch.float32, device=self.device)
        self._check(self.lib.bevgen_vq_decode_latents(self._h, _ptr(zq), n, lh, lw, _lib.VQ_OUT_DENORM if denormalize else _lib.VQ_OUT_RAW, _ptr(out), self._s()))
        self._done(check)
        return out

    # ------------------------------------------------------------------------------------------ per-kernel HIP-event timing
    PROFILE_KINDS = ("gemm", "conv3x3", "attention", "decode_attention", "gemm_skinny", "gemm_small")

    def profile_begin(self):
        self._check(self.lib.bevgen_profile_begin(self._h))

    def profile_end(self) -> Dict[str, Dict[str, float]]:
        """{kind: {'launches', 'ms', 'work'}}; work = FLOP (gemm/conv/attention) or bytes (decode_attention/gemm_skinny)."""
        buf = (C.c_double * (3 * len(self.PROFILE_KINDS)))()
        self._check(self.lib.bevgen_profile_end(self._h, buf))
        return {k: {"launches": buf[3 * i], "ms": buf[3 * i + 1], "work": buf[3 * i + 2]} for i, k in enumerate(self.PROFILE_KINDS)}

    def ar_step_timing(self, enable: bool = True):
        self._check(self.lib.bevgen_ar_step_timing(self._h, int(enable)))

    def ar_step_times(self, cap: int = 4096):
        """Durations (ms) of the decode steps of the most recent ar_sample (graph replays), as a float32 numpy array."""
        buf = (C.c_float * cap)()
        n = C.c_int(0)
        self._check(self.lib.bevgen_ar_step_times(self._h, buf, cap, C.byref(n)))
        return np.frombuffer(buf, dtype=np.float32, count=n.value).copy()

    def trace_begin(self):
        """Diagnostics: phase timestamps of the fused decode kernels (last launch of each kind)."""
        self._trace = torch.zeros((3, 4096, 8), dtype=torch.int64, device=self.device)
        self._check(self.lib.bevgen_set_trace_buffer(self._h, _ptr(self._trace)))

    def trace_end(self):
        torch.cuda.synchronize()
        self._check(self.lib.bevgen_set_trace_buffer(self._h, None))
        return self._trace.cpu()

    # ------------------------------------------------------------------------------------------ operator level (tests / roofline)
    def op_gemm(self, a, w, bias=None, residual=None, gelu=False, skinny=False):
        M, K = a.shape
        N = w.shape[0]
        c = torch.empty((M, N), dtype=torch.float32, device=self.device)
        self._check(self.lib.bevgen_op_gemm(self._h, _ptr(a), _ptr(w), _ptr(bias), _ptr(residual), _ptr(c), M, N, K, int(gelu), int(skinny), self._s()))
        return c

    def op_ln_gemm(self, a, w, ln_w=None, ln_b=None, bias=None, gelu=False, ksplit=0, eps=1e-5):
        """Decode-step projection kernel: act(LN?(a) w^T + bias); with K split over workgroups the partial sums are added here (the model path folds
        that sum into the consumer's row fetch)."""
        M, K = a.shape
        N = w.shape[0]
        ks = C.c_int(0)
        out = torch.empty((4, M, N), dtype=torch.float32, device=self.device)
        self._check(self.lib.bevgen_op_ln_gemm(self._h, _ptr(a), _ptr(ln_w), _ptr(ln_b), float(eps), _ptr(w), _ptr(bias), _ptr(out), M, N, K, int(gelu), int(ksplit),
                                               C.byref(ks), self._s()))
        return out[0] if ks.value == 1 else out[:ks.value].sum(0)

    def op_mlp_fused(self, x, ln_w, ln_b, w1, b1, w2, b2, w_f16=False, eps=1e-5):
        """Both MLP projections of a Route A decode layer in one launch: Linear2(GELU(Linear1(LayerNorm(x)))) for M <= 64 rows, without the residual."""
        M, D = x.shape
        out = torch.empty((M, D), dtype=torch.float32, device=self.device)
        self._check(self.lib.bevgen_op_mlp_fused(self._h, _ptr(x), _ptr(ln_w), _ptr(ln_b), float(eps), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), int(w_f16), _ptr(out), M, D, self._s()))
        return out

    def op_layernorm(self, x, gamma, beta=None, eps=1e-5):
        y = torch.empty_like(x)
        self._check(self.lib.bevgen_op_layernorm(self._h, _ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), x.shape[0], x.shape[1], float(eps), self._s()))
        return y

    def op_geglu_layernorm(self, h, gamma, ldy=None):
        rows, two_f = h.shape
        F = two_f // 2
        ldy = ldy or F
        y = torch.empty((rows, ldy), dtype=torch.float32, device=self.device)
        self._check(self.lib.bevgen_op_geglu_layernorm(self._h, _ptr(h), _ptr(gamma), _ptr(y), rows, F, ldy, self._s()))
        return y

    def op_attention(self, q, k, v, bias, scale, key_splits=1):
        B, H, Nq, _ = q.shape
        Nk_pad = k.shape[2]
        out = torch.empty((B, Nq, H * 64), dtype=torch.float32, device=self.device)
        ld = 0 if bias is None else bias.shape[-1]
        self._check(self.lib.bevgen_op_attention_ex(self._h, _ptr(q), _ptr(k), _ptr(v), _ptr(bias), ld, B, H, Nq, Nk_pad, float(scale), int(key_splits), _ptr(out),
                                                    self._s()))
        return out

    def op_decode_attention(self, q, kcache, vcache, n, bias=None, keep=None, scale=0.125, kv_dtype=0):
        B, HD = q.shape
        H = HD // 64
        Lmax = kcache.shape[2]
        out = torch.empty((B, HD), dtype=torch.float32, device=self.device)
        ldb = 0 if bias is None else bias.shape[-1]
        ldk = 0 if keep is None else keep.shape[-1]
        khs = 0 if keep is None or keep.dim() < 3 or keep.shape[0] == 1 else keep.shape[1] * keep.shape[2]
        self._check(self.lib.bevgen_op_decode_attention(self._h, _ptr(q), _ptr(kcache), _ptr(vcache), int(kv_dtype), _ptr(bias), ldb, _ptr(keep), ldk, khs,
                                                         B, H, int(n), Lmax, float(scale), _ptr(out), self._s()))
        return out

    def op_ar_attn_fused(self, x, ln_w, ln_b, wqkv, bqkv, kcache, vcache, n, *, partial=None, rbias=None, bias=None, attn_mask=None, layout=None, block=1, G=1, prefix=0,
                         kv_dtype=0, w_f16=False, split=False):
        """Attention half of a fused Route A decode layer (appends k/v to cache row n-1 in place) -> x2 [B, D]."""
        B, D = x.shape
        H = D // 64
        Lmax = kcache.shape[2]
        out = torch.empty((B, D), dtype=torch.float32, device=self.device)
        ns = 0 if partial is None else partial.shape[0]
        self._check(self.lib.bevgen_op_ar_attn_fused(self._h, _ptr(x), _ptr(partial), ns, _ptr(rbias), _ptr(ln_w), _ptr(ln_b), _ptr(wqkv), _ptr(bqkv), int(w_f16),
                                                     _ptr(kcache), _ptr(vcache), int(kv_dtype), _ptr(bias), 0 if bias is None else bias.shape[-1], _ptr(attn_mask),
                                                     _ptr(layout), int(block), B, int(G), H, int(n), Lmax, int(prefix), int(split), _ptr(out), self._s()))
        return out

    def op_conv3x3(self, x_nhwc, w_ohwi, bias, residual=None, upsample=False, kernel="auto"):
        """kernel: 'auto' (fp32 input: the register-staged kernel), 'dma' (the LDS-DMA kernel on planes split inside the call), 'dma_general' (its general variant
        also where the stride-1 variant applies)."""
        n, H, W, Cin = x_nhwc.shape
        Cout = w_ohwi.shape[0]
        oh, ow = (2 * H, 2 * W) if upsample else (H, W)
        y = torch.empty((n, oh, ow, Cout), dtype=torch.float32, device=self.device)
        flags = int(upsample) | {"auto": 0, "dma": 2, "dma_general": 6}[kernel]
        self._check(self.lib.bevgen_op_conv3x3(self._h, _ptr(x_nhwc), _ptr(w_ohwi), _ptr(bias), _ptr(residual), _ptr(y), n, H, W, Cin, Cout, flags, self._s()))
        return y

    def op_groupnorm(self, x_nhwc, gamma, beta, swish=True):
        n, H, W, Cc = x_nhwc.shape
        y = torch.empty_like(x_nhwc)
        self._check(self.lib.bevgen_op_groupnorm(self._h, _ptr(x_nhwc), _ptr(gamma), _ptr(beta), _ptr(y), n, H * W, Cc, int(swish), self._s()))
        return y
