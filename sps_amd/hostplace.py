"""Host placement of a GPU process: run (and first-touch its pinned buffers) on the CPUs of the NUMA node the GPU hangs off.

One process drives one GPU (bench.py, scripts/predict.py, scripts/train.py under torchrun).  On a two-socket host the launcher
may start it on the far socket: every doorbell, every SDMA copy submission and the pinned staging buffers then cross the
socket interconnect.  ``bind_to_gpu_numa`` reads the device's ``local_cpulist`` from sysfs and narrows the process's CPU
affinity to it (never widens it; no-op when sysfs does not tell).  Call it before the first pinned allocation."""
from __future__ import annotations

import os


def _parse_cpulist(text: str) -> set[int]:
    cpus: set[int] = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def bind_to_gpu_numa(device_index: int = 0) -> dict:
    """Returns {"pci": ..., "numa_node": ..., "cpus_before": n, "cpus_after": n, "bound": bool}; never raises."""
    info = {"pci": None, "numa_node": None, "cpus_before": None, "cpus_after": None, "bound": False}
    try:
        import torch
        p = torch.cuda.get_device_properties(device_index)
        pci = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        info["pci"] = pci
        base = os.path.join("/sys/bus/pci/devices", pci)
        with open(os.path.join(base, "numa_node")) as f:
            info["numa_node"] = int(f.read().strip())
        with open(os.path.join(base, "local_cpulist")) as f:
            local = _parse_cpulist(f.read())
        cur = os.sched_getaffinity(0)
        info["cpus_before"] = len(cur)
        want = cur & local
        if want and want != cur:
            os.sched_setaffinity(0, want)
            info["bound"] = True
        info["cpus_after"] = len(os.sched_getaffinity(0))
    except Exception as e:          # sysfs not exposed (containers), old torch: run where the launcher put us
        info["note"] = f"{type(e).__name__}: {e}"
    return info
