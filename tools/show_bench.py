import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
def show(k,v):
    r=v.get("roofline_decode_attention",{}); s=v.get("decode_step_roofline",{})
    ph=r.get("attention_phase",{})
    print(f"{k}: ms/step {v.get('ms_per_decode_step',0):.3f} attn-kernel frac {r.get('frac',0):.3f} by8d {r.get('frac_by_survey_8d_fp16_bytes',0):.3f} phase {ph.get('frac') and round(ph.get('frac'),3)} avg_us {r.get('avg_us',0):.1f} step frac {s.get('frac',0):.3f} visible {v.get('visible_fraction_of_causal_keys')}")
print("value", round(d["value"],3), "ms/step", round(d["ms_per_step"],1), "roofline", round(d["roofline"]["frac"],3), {k: round(v,1) for k,v in d["kernel_tflops"].items()}, {k: round(v,3) for k,v in d["kernel_time_share"].items()})
show("decode_f32", d)
for k in ("decode_f16_kv_cache","decode_f16_kv_cache_f16_weights","decode_density_035_f16_kv_cache","decode_split_path_f16_kv_cache_f16_weights"):
    if k in d: show(k,d[k])
print({k: round(d[k]["value"],2) for k in ("exact_fp32_mode","f16_weights_mode","released_3_camera_shape") if k in d})
if "config5_topk32_4_samples_per_layout" in d: print("config5 ms/step", d["config5_topk32_4_samples_per_layout"]["ms_per_decode_step"])
if "cpu_baseline" in d: print("cpu", d["cpu_baseline"]["value"])

if "single_scene_latency" in d:
    print("single scene", round(d["single_scene_latency"]["value"], 1), "ms")
