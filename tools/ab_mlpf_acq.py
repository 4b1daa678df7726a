# decode step A/B of the fused MLP launch's acquire form (BEVGEN_MLPF_ACQ is read once per process)
import sys, json, time, torch
sys.path.insert(0, '.')
import bench
r = bench.decode_leg(0, 16, 600, kv_cache="f16", weights="f16", path="fused")
print(json.dumps({k: r[k] for k in r if k in ("ms_per_decode_step", "ms_step", "median_ms", "step_frac")} or {k: (v if not isinstance(v, dict) else '...') for k, v in r.items()}, default=str)[:600])
