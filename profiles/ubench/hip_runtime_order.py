"""Load order of the two HIP runtimes a Python process can end up with (torch's bundled copy, /opt/rocm's behind libdbgk.so).
    python profiles/ubench/hip_runtime_order.py capi_first | torch_first
Round 3, MI355X box (before dbg_assembly_amd/capi.py imported torch itself):
  capi_first : both copies mapped, torch.cuda.is_available() False, pin_memory(): "No HIP GPUs are available"
  torch_first: only torch's copy mapped, both work
"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
order = sys.argv[1]
if order == "capi_first":
    from dbg_assembly_amd import capi
    print("dbgk devices", capi.lib().dbgk_device_count())
    import torch
    print("torch avail", torch.cuda.is_available(), torch.cuda.device_count())
else:
    import torch
    print("torch avail", torch.cuda.is_available(), torch.cuda.device_count())
    from dbg_assembly_amd import capi
    print("dbgk devices", capi.lib().dbgk_device_count())
try:
    t = torch.empty(1 << 20, dtype=torch.uint8).pin_memory()
    print("pinned ok")
except Exception as e:
    print("pin failed", e)
with open("/proc/self/maps") as f:
    libs = sorted({ln.split()[-1] for ln in f if "libamdhip64" in ln or "libhsa-runtime" in ln})
print(libs)
