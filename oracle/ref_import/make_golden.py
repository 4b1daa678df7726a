"""Generate tests/golden/*.npz by RUNNING THE IMPORTED REFERENCE (build container only).

TEST INFRASTRUCTURE.  Usage:  python -m oracle.ref_import.make_golden [--only NAME ...] [--skip-full]

Every fixture is data: seeded inputs + the outputs the reference modules (imported from /root/reference through
oracle/ref_import/stubs.py) produce for them, with this repo's deterministic weights loaded into the reference modules.
While generating, the CPU restatement (oracle/restate.py) is run on the same inputs and must agree (tolerances below), which
is what pins the oracle.  Weights are not stored (regenerated from seed + parameter name).
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from bevgen_amd import presets, synthetic  # noqa: E402
from oracle import cases, restate as R  # noqa: E402
from oracle.ref_import import refmodels as RM, stubs  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def sha(t: torch.Tensor) -> str:
    return hashlib.sha256(t.contiguous().numpy().tobytes()).hexdigest()


def rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()


def top2_margin(logits: torch.Tensor) -> float:
    v = logits.topk(2, dim=-1).values
    return float((v[..., 0] - v[..., 1]).min())


def save(name, **arrays):
    """Fixture = data only: wall-clock fields (`*_seconds`: how long the imported reference took here) go to tests/golden/ref_timings.json, so that
    re-running this script reproduces every .npz byte for byte (numpy stamps the zip members with a fixed date)."""
    os.makedirs(GOLDEN, exist_ok=True)
    path = os.path.join(GOLDEN, name + ".npz")
    timings = {k: round(float(np.asarray(v)), 3) for k, v in arrays.items() if k.endswith("_seconds")}
    np.savez_compressed(path, **{k: (v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrays.items() if k not in timings})
    if timings:
        tpath = os.path.join(GOLDEN, "ref_timings.json")
        allt = json.load(open(tpath)) if os.path.exists(tpath) else {}
        allt[name] = timings
        with open(tpath, "w") as f:
            json.dump(allt, f, indent=1, sort_keys=True)
    print(f"  wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


# ------------------------------------------------------------------------------------------------ tables
TABLE_CASES = {
    "cfg1": presets.config1,
    "nusc6_224x400": presets.config4,
    "argo3_rays": lambda: presets.config2(3),
    "nusc6_rays": lambda: presets.config2(6),
    "nusc3_ablation": lambda: presets.route_a(3, num_layers=2),
    "tiny_a_blk4": lambda: presets.tiny_route_a(3, block=4),
}


def golden_tables():
    for name, mk in TABLE_CASES.items():
        cfg = mk()
        ref = RM.ref_gpt_config(cfg)
        layouts, allowed = ref.get_mask()
        L = ref.gpt_block_size
        K = ref.num_cond_tokens
        rows = sorted({0, K - 1, K, K + 1, min(L - 1, K + 37), L - 13 if L > 13 else 0, L - 1})
        prob32 = ref.prob_matrix.to(torch.float32)
        save("tables_" + name,
             sizes=np.array([ref.num_cond_tokens, ref.num_cam_tokens, ref.num_img_tokens, ref.num_pad_tokens, ref.gpt_block_size]),
             forward_shuffle_idx=ref.forward_shuffle_idx.to(torch.int32), layout_bits=np.packbits(layouts.numpy().astype(np.uint8)),
             layout_shape=np.array(layouts.shape), mask_bits=np.packbits(ref.attention_mask.numpy().astype(np.uint8)),
             prob_rows_idx=np.array(rows), prob_rows=prob32[rows], prob_sha256=np.array(sha(prob32)), prob_is_f64=np.array(ref.prob_matrix.dtype == torch.float64),
             image_plane=stubs.import_reference().gpt.generate_grid(ref.cam_latent_h, ref.cam_latent_w).reshape(3, -1) * torch.tensor([ref.cam_res[0], ref.cam_res[1], 1.0])[:, None],
             bev_grid=stubs.import_reference().gpt.get_bev_grid(ref))


# ------------------------------------------------------------------------------------------------ Route M
def golden_route_m(case: cases.Case, full: bool):
    cfg = case.make_cfg()
    sd = cases.case_state_dict(case, cfg)
    mg, _ = RM.build_ref_maskgit(cfg, sd)
    bt = cases.inputs(case, cfg)
    batch = {"intrinsics_inv": bt["intrinsics_inv"], "extrinsics_inv": bt["extrinsics_inv"]}
    rows, T = case.batch * cfg.num_cams, cfg.num_cam_tokens
    g = torch.Generator().manual_seed(case.input_seed)
    ids = torch.randint(0, cfg.vocab_size + 1, (rows, T), generator=g)
    t0 = time.time()
    with torch.no_grad():
        lr, er = mg.transformer(ids, return_embed=True, conditioning_token_ids=bt["cond_ids"], batch=batch)
    t_fwd = time.time() - t0
    lo, eo = R.muse_forward(sd, cfg, ids, bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"], depth=cfg.num_layers, heads=cfg.num_heads)
    assert rel(lo, lr) < 2e-5 and rel(eo, er) < 2e-5, (rel(lo, lr), rel(eo, er))
    # deterministic generate
    with RM.deterministic_maskgit_noise(None), torch.no_grad():
        gen_ref = mg.generate(cond_images=bt["cond_ids"], fmap_size=cfg.cam_latent_res, batch=batch, timesteps=case.timesteps)
    trace = []
    gen_or = R.maskgit_generate(sd, cfg, bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"], depth=cfg.num_layers, heads=cfg.num_heads,
                                timesteps=case.timesteps, trace=trace)
    assert torch.equal(gen_ref, gen_or), "oracle generate != reference generate (greedy)"
    out = dict(ids_in=ids.to(torch.int16), cond_ids=bt["cond_ids"].to(torch.int16), I_inv=bt["intrinsics_inv"], E_inv=bt["extrinsics_inv"],
               gen_greedy=gen_ref.to(torch.int16), trace_ids=torch.stack([t["ids"] for t in trace]).to(torch.int16),
               min_margin_greedy=np.array(min(top2_margin(t["logits"]) for t in trace)), ref_forward_seconds=np.array(t_fwd))
    if case.heavy:   # how far the activations of this fixture leave the O(1-10) range of the N(0, 0.02) fixtures (recorded, and asserted to be far)
        out.update(logits_absmax=np.array(float(lr.abs().max())), embed_absmax=np.array(float(er.abs().max())))
        assert float(lr.abs().max()) > 50.0, "the heavy-tailed case was expected to produce large logits"
    if not full:
        noise = cases.maskgit_noise(case, cfg)
        with RM.deterministic_maskgit_noise(noise), torch.no_grad():
            gen_ref_n = mg.generate(cond_images=bt["cond_ids"], fmap_size=cfg.cam_latent_res, batch=batch, timesteps=case.timesteps)
        gen_or_n = R.maskgit_generate(sd, cfg, bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"], depth=cfg.num_layers, heads=cfg.num_heads,
                                      timesteps=case.timesteps, noise=noise)
        assert torch.equal(gen_ref_n, gen_or_n), "oracle generate != reference generate (explicit noise)"
        out.update(logits=lr, embed=er, gen_noisy=gen_ref_n.to(torch.int16), trace_scores=torch.stack([t["scores"] for t in trace]))
    else:
        out.update(logits_rows=lr[:, :4].clone(), logits_sha256=np.array(sha(lr)), embed_rows=er[:, :2].clone())
    save("route_m_" + case.name, **out)


def golden_route_m_branches(case: cases.Case):
    """The remaining branches of the reference's MaskGit.generate on a tiny case, each run by the imported reference and asserted against the restatement:
    force_not_use_token_critic (muse_net:611-619), + can_remask_prev_masked (:620-622; the reference asserts no_mask_token_prob > 0, a training-time knob the
    generate path reads nowhere else), and BASELINE config 5's top-k threshold 0.96875 with the token critic - greedy and with explicit noise."""
    cfg = case.make_cfg()
    sd = cases.maskgit_state_dict(cfg, case.weight_seed)
    mg, _ = RM.build_ref_maskgit(cfg, sd)
    mg.no_mask_token_prob = 0.1
    bt = cases.inputs(case, cfg)
    batch = {"intrinsics_inv": bt["intrinsics_inv"], "extrinsics_inv": bt["extrinsics_inv"]}
    noise = cases.maskgit_noise(case, cfg, seed=13)
    out = dict(cond_ids=bt["cond_ids"].to(torch.int16), I_inv=bt["intrinsics_inv"], E_inv=bt["extrinsics_inv"])
    variants = {"nocritic": dict(force_not_use_token_critic=True), "nocritic_remask": dict(force_not_use_token_critic=True, can_remask_prev_masked=True),
                "topk096875": dict(topk_filter_thres=0.96875)}
    for tag, kw in variants.items():
        okw = dict(use_token_critic=not kw.get("force_not_use_token_critic", False), can_remask_prev_masked=kw.get("can_remask_prev_masked", False),
                   topk_filter_thres=kw.get("topk_filter_thres", 0.9))
        for ntag, nz in (("greedy", None), ("noisy", noise)):
            with RM.deterministic_maskgit_noise(nz), torch.no_grad():
                ref = mg.generate(cond_images=bt["cond_ids"], fmap_size=cfg.cam_latent_res, batch=batch, timesteps=case.timesteps, **kw)
            mine = R.maskgit_generate(sd, cfg, bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"], depth=cfg.num_layers, heads=cfg.num_heads,
                                      timesteps=case.timesteps, noise=nz, **okw)
            assert torch.equal(ref, mine), f"oracle generate != reference generate ({tag}, {ntag})"
            out[f"gen_{tag}_{ntag}"] = ref.to(torch.int16)
    assert not torch.equal(out["gen_nocritic_noisy"], out["gen_nocritic_remask_noisy"]) or not torch.equal(out["gen_nocritic_greedy"], out["gen_nocritic_remask_greedy"]), \
        "the re-masking variant must differ somewhere"
    save("route_m_branches_" + case.name, **out)


# ------------------------------------------------------------------------------------------------ Route A
class _RefSampler:
    """Drives the reference's OWN sampling loop, `Net2NetTransformer.sample` (ar_lm:154-227: mask-id initialisation, decode order from a fresh
    CustomPermuter, temperature, `top_k_logits` ar_lm:138-142, softmax, topk(1) / multinomial), as an unbound method on a minimal stand-in object:
    the LightningModule constructor would instantiate both VQGANs through Hydra, which the loop does not need.  The stand-in supplies exactly the
    attributes the loop reads (cfg, transformer, skip_sampling, debug_viz, top_k_logits, and for partial decoding get_input / encode_to_z /
    expand_all_images returning the given ground-truth ids).  `steps`: the loop has no step limit, so for the full-size head cases the module's
    tqdm is replaced by a truncating iterator; the final `x.max() < vocab_size` assert then fires and the partially filled x is read back from the
    tensor the loop handed to the transformer.  Stochastic sampling: torch.multinomial is replaced by the inverse-CDF draw on explicit uniforms
    (the definition this repo uses: first index whose cumulative probability exceeds u * total)."""

    def __init__(self, gpt, rcfg):
        import types

        self.ns = stubs.import_reference()
        self.N2N = self.ns.ar_lm.Net2NetTransformer
        outer = self

        class Transformer:
            training = False

            def __call__(self, x, cond, batch, sampling=True):
                outer.x_seen = x
                out = gpt(x, cond, batch, sampling=sampling)
                if outer.step_logits is not None:
                    outer.step_logits.append(out)
                return out

        class Shim:
            pass

        o = Shim()
        o.cfg, o.transformer, o.skip_sampling, o.debug_viz, o.first_stage_key = rcfg, Transformer(), False, False, "image"
        o.top_k_logits = types.MethodType(self.N2N.top_k_logits, o)
        o.get_input = lambda key, batch: None
        o.expand_all_images = lambda z: z
        self.o = o
        self.x_seen = None
        self.step_logits = None

    def run(self, B, cond, batch, *, steps=0, temperature=1.0, top_k=None, noise_u=None, partial_idx=None, z=None, record=None):
        import itertools

        mod = self.ns.ar_lm
        self.step_logits = record
        self.o.encode_to_z = lambda x, batch: (None, z)
        old_tqdm, old_mn = mod.tqdm, torch.multinomial
        state = {"s": 0}

        def multinomial(probs, num_samples=1, **kw):
            u = noise_u[state["s"]]
            state["s"] += 1
            cdf = probs.cumsum(dim=-1)
            return (cdf <= u[:, None] * cdf[:, -1:]).sum(dim=-1).clamp(max=probs.shape[-1] - 1)[:, None]

        mod.tqdm = (lambda it: itertools.islice(it, steps)) if steps else (lambda it: it)
        if noise_u is not None:
            torch.multinomial = multinomial
        try:
            x0 = torch.zeros((B, 1), dtype=torch.long)
            try:
                return self.N2N.sample(self.o, x0, cond, batch, temperature=temperature, sample=noise_u is not None, top_k=top_k, partial_decoding_idx=partial_idx)
            except AssertionError:
                assert steps, "the reference loop failed its own assertions"
                return self.x_seen.clone()   # truncated run: unfilled positions still hold the mask id
        finally:
            mod.tqdm, torch.multinomial = old_tqdm, old_mn


def golden_route_a(case: cases.Case, full: bool):
    cfg = case.make_cfg()
    sd = cases.case_state_dict(case, cfg)
    bt = cases.inputs(case, cfg)
    batch = {"intrinsics_inv": bt["intrinsics_inv"], "extrinsics_inv": bt["extrinsics_inv"]}
    B, C, T, N = case.batch, cfg.num_cams, cfg.num_cam_tokens, cfg.num_img_tokens
    g = torch.Generator().manual_seed(case.input_seed)
    ids = torch.randint(0, cfg.vocab_size, (B, C, T), generator=g)
    layout_seed = case.layout_seed
    while True:
        # density < 1: sd receives the layouts the reference draws, layer by layer.  A draw can leave a condition row without any visible block in some
        # head (nothing forces the cond x cond blocks in); the reference then fails its own finite-logits assert (gpt:388): take the next seed.
        gpt, rcfg = RM.build_ref_gpt(cfg, sd, layout_seed=layout_seed)
        try:
            with torch.no_grad():
                lr = gpt(ids.clone(), bt["cond_ids"], batch, sampling=True)
            break
        except AssertionError:
            assert layout_seed and layout_seed < case.layout_seed + 64, "reference forward is not finite"
            print(f"  layout seed {layout_seed}: a row without visible keys, trying the next seed")
            layout_seed += 1
    lo = R.gpt_forward(sd, cfg, ids, bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"])
    assert rel(lo, lr) < 2e-5, rel(lo, lr)
    # greedy sampling by the reference's own loop (one full forward per token)
    sampler = _RefSampler(gpt, rcfg)
    n_steps = case.steps or N
    rec = []
    t0 = time.time()
    x = sampler.run(B, bt["cond_ids"], batch, steps=case.steps, record=rec)
    t_ref = time.time() - t0
    order = [int(cfg.forward_shuffle_idx[s]) for s in range(n_steps)]
    step_logits = [rec[s][:, order[s]].clone() for s in range(n_steps)]
    del rec
    lc = []
    xc = R.ar_sample_cached(sd, cfg, bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"], logits_out=lc, steps=n_steps)
    assert torch.equal(x, xc), "KV-cache oracle != reference Net2NetTransformer.sample (greedy)"
    err = max(rel(a, b) for a, b in zip(lc, step_logits))
    assert err < 5e-5, err
    sl = torch.stack(step_logits)  # [N,B,V]
    out = dict(ids_in=ids.to(torch.int16), cond_ids=bt["cond_ids"].to(torch.int16), I_inv=bt["intrinsics_inv"], E_inv=bt["extrinsics_inv"],
               sample_greedy=x.to(torch.int16), min_margin_greedy=np.array(top2_margin(sl)), ref_sample_seconds=np.array(t_ref))
    if case.layout_seed:
        lay = torch.stack([sd[f"blocks.{i}.attention.sparse_self_attention.master_layout"] for i in range(cfg.num_layers)])
        assert any(not torch.equal(lay[0], lay[i]) for i in range(1, cfg.num_layers)), "the layers were expected to draw different layouts"
        out.update(layer_layout_bits=np.packbits(lay.numpy().astype(np.uint8)), layer_layout_shape=np.array(lay.shape), layout_seed=np.array(layout_seed),
                   layout_fill=np.array(float(lay.float().mean())))
    if not full:
        noise_u = synthetic.uniform_noise((N, B), 11, 2)
        xs = sampler.run(B, bt["cond_ids"], batch, temperature=0.9, top_k=8, noise_u=noise_u)   # reference loop incl. its top_k_logits
        xs1 = R.ar_sample_full_recompute(sd, cfg, bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"], temperature=0.9, top_k=8, noise_u=noise_u)
        xs2 = R.ar_sample_cached(sd, cfg, bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"], temperature=0.9, top_k=8, noise_u=noise_u)
        assert torch.equal(xs, xs1) and torch.equal(xs, xs2), "oracle top-k sampling != reference loop"
        # top_k_logits tie behaviour (ar_lm:138-142: values equal to the k-th largest are kept), straight from the reference method
        tl = torch.tensor([[1.0, 3.0, 3.0, 2.0, 3.0, 0.5], [0.0, -1.0, 0.0, 0.0, -2.0, 5.0]])
        tk = sampler.o.top_k_logits(tl, 2)
        assert torch.equal(tk, R.top_k_logits(tl, 2)), "oracle top_k_logits != reference"
        out.update(logits_full=lr, step_logits=sl, sample_topk8=xs.to(torch.int16), topk_tie_in=tl, topk_tie_out=tk)
        # partial decoding (ar_lm:161-165, 181-182) by the reference loop: the fixed cameras start from "ground-truth" ids (encode_to_z output in
        # the reference; seeded random ids here) and their positions are skipped
        partial_idx = [1] if C < 6 else [0, 2]
        z = torch.randint(0, cfg.vocab_size, (B, C, T), generator=g)
        xp = sampler.run(B, bt["cond_ids"], batch, partial_idx=partial_idx, z=z)
        xp_or = R.ar_sample_full_recompute(sd, cfg, bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"], partial_decoding_idx=partial_idx, z_indices=z)
        xp_kv = R.ar_sample_cached(sd, cfg, bt["cond_ids"], bt["intrinsics_inv"], bt["extrinsics_inv"], forced_ids=R.partial_forced_ids(cfg, partial_idx, z))
        assert torch.equal(xp, xp_or), "oracle partial decoding != reference loop"
        assert torch.equal(xp, xp_kv), "KV-cache partial decoding (forced tokens) != reference loop"
        out.update(partial_idx=np.array(partial_idx), partial_z=z.to(torch.int16), sample_partial=xp.to(torch.int16))
    else:
        keep = sorted({0, 1, 17, N // 2, N - 1}) if n_steps == N else list(range(n_steps))
        out.update(step_logits_idx=np.array(keep), step_logits=sl[keep])
    save("route_a_" + case.name, **out)


def golden_tables_density():
    """S4 with density < 1 (maskgen:217-251, perm:125-143): the reference's multi_outward_pattern under a fixed torch seed, and this repo's
    tables.head_layouts under the same seed (same primitive, same call order) - asserted equal here, re-checked by tests/test_tables.py."""
    from bevgen_amd import tables

    for name, mk in (("nusc6_224x400_d035", lambda: presets.config4(density=0.35)), ("tiny_a_blk4_d035", lambda: presets.tiny_route_a(3, block=4, density=0.35))):
        cfg = mk()
        ref = RM.ref_gpt_config(cfg)
        seed = 4242
        torch.manual_seed(seed)
        lay_ref, _ = ref.get_mask()
        torch.manual_seed(seed)
        lay = tables.head_layouts(cfg, cfg._patterns)
        assert torch.equal(lay_ref.to(torch.int64), lay), "tables.head_layouts != reference multi_outward_pattern under the same seed"
        assert not torch.equal(lay[0], lay[1]), "heads were expected to differ"
        save("tables_" + name, seed=np.array(seed), layout_bits=np.packbits(lay_ref.numpy().astype(np.uint8)), layout_shape=np.array(lay_ref.shape),
             layout_fill=np.array(float(lay_ref.float().mean())), density=np.array(cfg.density))


# ------------------------------------------------------------------------------------------------ VQGAN decode
def golden_vq_full():
    """Full-size f16 decoder (42 M parameters, 252 GFLOP per image): one image, pixels stored as float16 of the denormalised [0,1] output
    (quantisation 2.4e-4, inside the 1e-3 parity tolerance) plus exact per-channel statistics of the fp32 reference output."""
    v = cases.VQ_FULL
    dd = v["dd"]
    sd = cases.vq_state_dict(dd, v["n_embed"], v["embed_dim"], v["seed"], with_encoder=True)   # strict load in the reference; decoder tensors do not depend on it
    lat = dd["resolution"] // 2 ** (len(dd["ch_mult"]) - 1)
    vq = RM.build_ref_vqmodel(dd, v["n_embed"], v["embed_dim"], sd, (dd["resolution"],) * 2, (lat, lat))
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, v["n_embed"], (v["n_images"], lat * lat), generator=g)
    t0 = time.time()
    with torch.no_grad():
        zq = vq.quantize.get_codebook_entry(ids.reshape(-1), shape=(v["n_images"], lat, lat, v["embed_dim"]))
        xr = vq.decode(zq)
        xd = stubs.import_reference().util.denormalize_tensor(xr, keep_tensor=True)
    t_ref = time.time() - t0
    xo = R.vq_decode_ids(sd, dd, ids, (lat, lat), denorm=False)
    assert rel(xo, xr) < 2e-5, rel(xo, xr)
    save("vq_full", ids=ids.to(torch.int16), pixels_denorm_f16=xd.to(torch.float16), raw_mean=xr.mean(dim=(0, 2, 3)), raw_absmax=np.array(float(xr.abs().max())),
         raw_rows=xr[:, :, ::64, :].clone(), ref_decode_seconds=np.array(t_ref))


def golden_vq():
    v = cases.VQ_TINY
    dd = v["dd"]
    sd = cases.vq_state_dict(dd, v["n_embed"], v["embed_dim"], v["seed"], with_encoder=True)
    lat = dd["resolution"] // 2 ** (len(dd["ch_mult"]) - 1)
    vq = RM.build_ref_vqmodel(dd, v["n_embed"], v["embed_dim"], sd, (dd["resolution"],) * 2, (lat, lat))
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(0, v["n_embed"], (v["n_images"], lat * lat), generator=g)
    with torch.no_grad():
        zq = vq.quantize.get_codebook_entry(ids.reshape(-1), shape=(v["n_images"], lat, lat, v["embed_dim"]))
        xr = vq.decode(zq)
        xd = stubs.import_reference().util.denormalize_tensor(xr, keep_tensor=True)
    xo = R.vq_decode_ids(sd, dd, ids, (lat, lat), denorm=False)
    assert rel(xo, xr) < 1e-5, rel(xo, xr)
    assert (R.denormalize(xo) - xd).abs().max() < 1e-5
    # encode side (VQModel.encode -> VectorQuantizer2.forward arg-min), image model and the 7-channel BEV segmentation model
    enc = {}
    for tag, vv, model in (("img", v, vq), ("seg", cases.VQ_TINY_SEG, None)):
        ddv = vv["dd"]
        sdv = cases.vq_state_dict(ddv, vv["n_embed"], vv["embed_dim"], vv["seed"], with_encoder=True)
        if model is None:
            ns = stubs.import_reference()
            from multi_view_generation.modules.losses.vqperceptual import DummyLoss
            model = ns.vqgan.VQSegmentationModel(n_labels=ddv["in_channels"], ddconfig=dict(ddv), lossconfig=DummyLoss(), n_embed=vv["n_embed"], embed_dim=vv["embed_dim"],
                                                 cam_res=(ddv["resolution"],) * 2, cam_latent_res=(lat, lat), cam_emd_dim=vv["embed_dim"])
            sdl = dict(sdv); sdl["colorize"] = model.state_dict()["colorize"]
            model.load_state_dict(sdl, strict=True)
            model.eval()
        gx = torch.Generator().manual_seed(17)
        xin = torch.randn(vv["n_images"], ddv["in_channels"], ddv["resolution"], ddv["resolution"], generator=gx)
        with torch.no_grad():
            _, _, info = model.encode(xin, None)
        ids_ref = info[2].view(vv["n_images"], -1)
        ids_or, dist = R.vq_encode_ids(sdv, ddv, xin, return_distances=True)
        assert torch.equal(ids_ref, ids_or), "oracle encode ids != reference"
        top2 = dist.topk(2, dim=1, largest=False).values
        enc.update({f"enc_{tag}_x": xin, f"enc_{tag}_ids": ids_ref.to(torch.int16), f"enc_{tag}_min_margin": np.array(float((top2[:, 1] - top2[:, 0]).min())),
                    f"enc_{tag}_dist_scale": np.array(float(dist.abs().mean()))})
    save("vq_tiny", ids=ids.to(torch.int16), pixels_raw=xr, pixels_denorm=xd, **enc)


def golden_vq_heavy(v=None, name="vq_tiny_heavy"):
    """vq_tiny with trained-like heavy-tailed weights (cases.heavy_tail: outlier channels x 30-100 in the GroupNorm gains, the nin_shortcut 1x1 convolutions - which
    run on UN-normalised tensors - and conv_in): decode by the imported reference, asserted against the restatement."""
    v = v or cases.VQ_TINY_HEAVY
    dd = v["dd"]
    sd = cases.heavy_tail(cases.vq_state_dict(dd, v["n_embed"], v["embed_dim"], v["seed"], with_encoder=True), v["heavy"], lo=v.get("lo", 30.0), hi=v.get("hi", 100.0))
    lat = dd["resolution"] // 2 ** (len(dd["ch_mult"]) - 1)
    vq = RM.build_ref_vqmodel(dd, v["n_embed"], v["embed_dim"], sd, (dd["resolution"],) * 2, (lat, lat))
    g = torch.Generator().manual_seed(43)
    ids = torch.randint(0, v["n_embed"], (v["n_images"], lat * lat), generator=g)
    with torch.no_grad():
        zq = vq.quantize.get_codebook_entry(ids.reshape(-1), shape=(v["n_images"], lat, lat, v["embed_dim"]))
        xr = vq.decode(zq)
        xd = stubs.import_reference().util.denormalize_tensor(xr, keep_tensor=True)
    xo = R.vq_decode_ids(sd, dd, ids, (lat, lat), denorm=False)
    assert rel(xo, xr) < 2e-5, rel(xo, xr)
    save(name, ids=ids.to(torch.int16), pixels_raw=xr, pixels_denorm=xd, raw_absmax=np.array(float(xr.abs().max())))


def golden_vq_rect():
    """Non-square latents (the reference's nuScenes experiment: cam_res [224, 400], cam_latent_res [14, 25], configs/experiment/muse_stage_two_multi_view.yaml)
    through the fully convolutional decoder / encoder: tiny model with full tensors, and the released f16 architecture at 14 x 25 -> 224 x 400
    (pixels as float16 of the denormalised output) plus the full-size ENCODER (s1model:342-433, quant:271-312) at 256 x 256 and 224 x 400."""
    ns = stubs.import_reference()
    out = {}
    # ---- tiny: latent 3 x 5 -> 24 x 40 pixels
    v = cases.VQ_TINY
    dd = v["dd"]
    f = 2 ** (len(dd["ch_mult"]) - 1)
    sd = cases.vq_state_dict(dd, v["n_embed"], v["embed_dim"], v["seed"], with_encoder=True)
    lh, lw = 3, 5
    vq = RM.build_ref_vqmodel(dd, v["n_embed"], v["embed_dim"], sd, (lh * f, lw * f), (lh, lw))
    g = torch.Generator().manual_seed(23)
    ids = torch.randint(0, v["n_embed"], (2, lh * lw), generator=g)
    with torch.no_grad():
        zq = vq.quantize.get_codebook_entry(ids.reshape(-1), shape=(2, lh, lw, v["embed_dim"]))
        xr = vq.decode(zq)
        xd = ns.util.denormalize_tensor(xr, keep_tensor=True)
        xin = torch.randn(2, dd["in_channels"], lh * f, lw * f, generator=g)
        _, _, info = vq.encode(xin, None)
    xo = R.vq_decode_ids(sd, dd, ids, (lh, lw), denorm=False)
    assert rel(xo, xr) < 1e-5, rel(xo, xr)
    ids_enc = info[2].view(2, -1)
    ids_or, dist = R.vq_encode_ids(sd, dd, xin, return_distances=True)
    assert torch.equal(ids_enc, ids_or)
    top2 = dist.topk(2, dim=1, largest=False).values
    out.update(tiny_latent=np.array([lh, lw]), tiny_ids=ids.to(torch.int16), tiny_pixels_raw=xr, tiny_pixels_denorm=xd, tiny_enc_x=xin, tiny_enc_ids=ids_enc.to(torch.int16),
               tiny_enc_min_margin=np.array(float((top2[:, 1] - top2[:, 0]).min())))
    # ---- full size f16: decode 14 x 25 -> 224 x 400, encode 224 x 400 and 256 x 256 (inputs regenerated from their seeds by the tests)
    v = cases.VQ_FULL
    dd = v["dd"]
    sd = cases.vq_state_dict(dd, v["n_embed"], v["embed_dim"], v["seed"], with_encoder=True)
    lh, lw = 14, 25
    vq = RM.build_ref_vqmodel(dd, v["n_embed"], v["embed_dim"], sd, (224, 400), (lh, lw))
    g = torch.Generator().manual_seed(29)
    ids = torch.randint(0, v["n_embed"], (1, lh * lw), generator=g)
    t0 = time.time()
    with torch.no_grad():
        zq = vq.quantize.get_codebook_entry(ids.reshape(-1), shape=(1, lh, lw, v["embed_dim"]))
        xr = vq.decode(zq)
        xd = ns.util.denormalize_tensor(xr, keep_tensor=True)
    t_dec = time.time() - t0
    xo = R.vq_decode_ids(sd, dd, ids, (lh, lw), denorm=False)
    assert rel(xo, xr) < 2e-5, rel(xo, xr)
    out.update(full_latent=np.array([lh, lw]), full_ids=ids.to(torch.int16), full_pixels_denorm_f16=xd.to(torch.float16), full_raw_rows=xr[:, :, ::56, :].clone(),
               full_raw_absmax=np.array(float(xr.abs().max())), full_decode_seconds=np.array(t_dec))
    for tag, (H, W), seed in (("enc256", (256, 256), 31), ("enc224x400", (224, 400), 37)):
        gx = torch.Generator().manual_seed(seed)
        xin = torch.randn(1, 3, H, W, generator=gx)
        t0 = time.time()
        with torch.no_grad():
            _, _, info = vq.encode(xin, None)
        t_enc = time.time() - t0
        ids_ref = info[2].view(1, -1)
        ids_or, dist = R.vq_encode_ids(sd, dd, xin, return_distances=True)
        assert torch.equal(ids_ref, ids_or), "oracle encode ids != reference (full size)"
        top2 = dist.topk(2, dim=1, largest=False).values
        out.update({f"{tag}_seed": np.array(seed), f"{tag}_hw": np.array([H, W]), f"{tag}_ids": ids_ref.to(torch.int16), f"{tag}_x_sha256": np.array(sha(xin)),
                    f"{tag}_min_margin": np.array(float((top2[:, 1] - top2[:, 0]).min())), f"{tag}_dist_scale": np.array(float(dist.abs().mean())),
                    f"{tag}_seconds": np.array(t_enc)})
    save("vq_rect", **out)


def golden_keys():
    """state_dict key -> shape of the reference modules (tiny sizes), as instantiated by the reference's own constructors."""
    import json

    out = {}
    cfg = presets.tiny_route_m(3, legacy=False)
    mg, _ = RM.build_ref_maskgit(cfg, cases.maskgit_state_dict(cfg, 1))
    out["maskgit_tiny_route_m_3cam"] = {k: list(v.shape) for k, v in mg.state_dict().items()}
    ca = presets.tiny_route_a(3)
    gpt, _ = RM.build_ref_gpt(ca, cases.gpt_state_dict(ca, 1))
    out["gpt_tiny_route_a_3cam"] = {k: list(v.shape) for k, v in gpt.state_dict().items()}
    dd = presets.VQ_DDCONFIG_TINY
    vq = RM.build_ref_vqmodel(dd, 64, 64, cases.vq_state_dict(dd, 64, 64, 1, with_encoder=True), (64, 64), (8, 8))
    out["vqmodel_tiny"] = {k: list(v.shape) for k, v in vq.state_dict().items()}
    path = os.path.join(GOLDEN, "state_dict_keys.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    print(f"  wrote {path}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", nargs="*", default=None)
    ap.add_argument("--skip-full", action="store_true")
    ap.add_argument("--vq-full", action="store_true", help="also (re)generate the full-size VQGAN decode golden")
    args = ap.parse_args()
    torch.manual_seed(0)
    want = lambda n: args.only is None or n in args.only
    if want("tables"):
        print("tables")
        golden_tables()
    if want("tables_density"):
        print("tables_density")
        golden_tables_density()
    if want("vq"):
        print("vq")
        golden_vq()
    if want("vq_heavy"):
        print("vq_heavy")
        golden_vq_heavy()
        golden_vq_heavy(cases.VQ_TINY_HEAVY_MILD, "vq_tiny_heavy_mild")
    if want("vq_rect"):
        print("vq_rect")
        golden_vq_rect()
    if want("keys"):
        print("keys")
        golden_keys()
    if (args.only is not None and "vq_full" in args.only) or (args.only is None and not args.skip_full) or args.vq_full:
        print("vq_full")
        golden_vq_full()
    if want("m_branches"):
        print("m_branches")
        golden_route_m_branches(cases.CASES["m_tiny_rays"])
    for name, case in cases.CASES.items():
        full = name in ("a_config1", "m_full_3cam", "m_full_6cam", "a_config4_head", "a_config4_d035_head")
        if not want(name) or (full and args.skip_full):
            continue
        print(name)
        t0 = time.time()
        (golden_route_m if case.route == "m" else golden_route_a)(case, full)
        print(f"  {time.time() - t0:.1f}s")


if __name__ == "__main__":
    main()
