"""Keep the process on the CPU cores of the GPU's NUMA node.

A training step of this path issues ~1100 kernel launches, and for a fifth of the step the GPU waits for the host to issue
them.  On a two-socket host a process that the scheduler places (or migrates) on the socket the GPU is not
attached to pays the inter-socket hop on every doorbell write and every host read.  ``pin_to_gpu_node`` restricts the
calling thread - and the threads it creates afterwards, the library's launch helper among them - to the cores of the
GPU's node.  No effect (and no error) where the topology cannot be read; ``GPN_NO_PIN=1`` disables it."""
import os
from typing import Optional, Set


def _parse_cpulist(text: str) -> Set[int]:
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-")
            cpus.update(range(int(a), int(b) + 1))
        else:
            cpus.add(int(part))
    return cpus


def gpu_numa_node(device_index: int = 0) -> Optional[int]:
    try:
        import torch
        props = torch.cuda.get_device_properties(device_index)
        bdf = f"{getattr(props, 'pci_domain_id', 0):04x}:{props.pci_bus_id:02x}:{props.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as fh:
            node = int(fh.read().strip())
        return node if node >= 0 else None
    except Exception:  # no such attribute / file: leave the affinity alone
        return None


def pin_to_gpu_node(device_index: int = 0) -> Optional[Set[int]]:
    """-> the CPU set now in force, or None if nothing was changed"""
    if os.environ.get("GPN_NO_PIN") == "1" or not hasattr(os, "sched_setaffinity"):
        return None
    node = gpu_numa_node(device_index)
    if node is None:
        return None
    try:
        with open(f"/sys/devices/system/node/node{node}/cpulist") as fh:
            node_cpus = _parse_cpulist(fh.read())
        allowed = os.sched_getaffinity(0) & node_cpus
        if not allowed or allowed == os.sched_getaffinity(0):
            return None
        os.sched_setaffinity(0, allowed)
        return allowed
    except OSError:
        return None
