#!/usr/bin/env python
"""stdin: one bench.py JSON line -> the decode-leg numbers."""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print("scenes/s", round(d["value"], 3), "| ms/decode-step", round(d.get("ms_per_decode_step", 0), 4), "| decode scenes/s", round(d.get("decode_scenes_per_s", 0), 3),
      "| decode-attn frac", round(d.get("roofline_decode_attention", {}).get("frac", 0), 4), "| weight stream", d.get("decode_weight_stream"))
