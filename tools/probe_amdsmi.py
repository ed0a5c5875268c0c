"""What amdsmi exposes on the GPU box (one-off probe for bench.py's gpu_state block)."""
import json, time
import amdsmi
amdsmi.amdsmi_init()
hs = amdsmi.amdsmi_get_processor_handles()
print("handles", len(hs))
h = hs[0]
print("bdf", amdsmi.amdsmi_get_gpu_device_bdf(h))
t0 = time.perf_counter()
m = amdsmi.amdsmi_get_gpu_metrics_info(h)
print("metrics call ms", (time.perf_counter() - t0) * 1e3)
print(json.dumps({k: (v if not isinstance(v, (bytes,)) else str(v)) for k, v in m.items()}, default=str, indent=0)[:6000])
for name in ("amdsmi_get_power_info", "amdsmi_get_gpu_activity", "amdsmi_get_violation_status", "amdsmi_get_gpu_compute_partition", "amdsmi_get_gpu_memory_partition"):
    try:
        print(name, getattr(amdsmi, name)(h))
    except Exception as e:
        print(name, "ERR", e)
for ct in ("GFX", "MEM", "SOC", "DF"):
    try:
        print("clock", ct, amdsmi.amdsmi_get_clock_info(h, getattr(amdsmi.AmdSmiClkType, ct)))
    except Exception as e:
        print("clock", ct, "ERR", e)
import torch
p = torch.cuda.get_device_properties(0)
print({k: getattr(p, k) for k in dir(p) if k.startswith("pci") or k in ("name", "multi_processor_count", "clock_rate", "memory_clock_rate")})
