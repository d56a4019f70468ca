"""Core / NUMA placement of the ranks of one node (patch-DDP: one process per GPU, BASELINE config 4).

Every rank issues ~1300 kernel launches per training step from ONE Python thread (plus the autograd engine's thread and the prefetcher's): at
eight ranks per host the launch threads must not migrate or share cores, and the pinned staging buffers should live on the NUMA node of the
rank's GPU.  `pin_rank(...)` gives each local rank a disjoint slice of the cores of its GPU's NUMA node (from sysfs: the GPU's PCI address ->
/sys/bus/pci/devices/<addr>/numa_node -> /sys/devices/system/node/node<k>/cpulist), or an even split of the allowed cores when the topology
is not exposed (containers often report numa_node = -1).  Host-side only; nothing here touches the GPU work."""
import os


def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    out = []
    for part in text.strip().split(","):
        part = part.strip()
        if not part:
            continue
        if "-" in part:
            a, b = part.split("-")
            out.extend(range(int(a), int(b) + 1))
        else:
            out.append(int(part))
    return out


def plan(local_rank, local_world, allowed, numa_of_rank=None, node_cpus=None):
    """the cores of `local_rank`: ranks sharing a NUMA node split that node's (allowed) cores evenly, in rank order; unknown topology
    (numa_of_rank is None or contains -1 / None): an even split of all allowed cores.  Never empty, never overlapping between ranks."""
    allowed = sorted(set(int(c) for c in allowed))
    if not allowed:
        return []
    known = numa_of_rank is not None and node_cpus is not None and all(n is not None and n >= 0 for n in numa_of_rank)
    if known:
        node = numa_of_rank[local_rank]
        mates = [r for r in range(local_world) if numa_of_rank[r] == node]
        pool = [c for c in node_cpus.get(node, []) if c in set(allowed)]
        if len(pool) >= len(mates):
            i, per = mates.index(local_rank), len(pool) // len(mates)
            return pool[i * per:(i + 1) * per]
    per = max(1, len(allowed) // max(1, local_world))
    if per * local_world > len(allowed):        # more ranks than cores: share round-robin
        return [allowed[local_rank % len(allowed)]]
    return allowed[local_rank * per:(local_rank + 1) * per]


def gpu_numa_node(device_index):
    """NUMA node of a torch CUDA device via its PCI address, or None"""
    try:
        import torch
        p = torch.cuda.get_device_properties(device_index)
        addr = "%04x:%02x:%02x.0" % (int(getattr(p, "pci_domain_id", 0)), int(p.pci_bus_id), int(p.pci_device_id))
        with open("/sys/bus/pci/devices/%s/numa_node" % addr) as f:
            n = int(f.read().strip())
        return n if n >= 0 else None
    except Exception:
        return None


def node_cpulists():
    out = {}
    base = "/sys/devices/system/node"
    try:
        for d in os.listdir(base):
            if d.startswith("node") and d[4:].isdigit():
                with open(os.path.join(base, d, "cpulist")) as f:
                    out[int(d[4:])] = parse_cpulist(f.read())
    except OSError:
        pass
    return out


def pin_rank(local_rank, local_world, device_indices=None, set_torch_threads=True):
    """Pin the calling process (all its threads started later inherit it) to its slice; returns a record for logs / the bench line.
    device_indices: the CUDA device of every local rank (default: rank r -> device r)."""
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except AttributeError:
        return {"pinned": False, "why": "no sched_getaffinity on this platform"}
    if device_indices is None:
        device_indices = list(range(local_world))
    # placement is an optimisation: whatever goes wrong here (sysfs without the expected files, a cgroup that refuses the mask) must leave
    # the rank running unpinned, never take a multi-GPU job down
    try:
        numa = [gpu_numa_node(d) for d in device_indices]
        cpus = plan(local_rank, local_world, allowed, numa, node_cpulists())
        if not cpus:
            return {"pinned": False, "why": "no allowed cores"}
        os.sched_setaffinity(0, cpus)
    except Exception as e:       # noqa: BLE001
        return {"pinned": False, "why": "%s: %s" % (type(e).__name__, e)}
    rec = {"pinned": True, "local_rank": local_rank, "numa_node": numa[local_rank], "cores": len(cpus), "first_core": cpus[0], "last_core": cpus[-1]}
    if set_torch_threads:
        try:
            import torch
            torch.set_num_threads(max(1, min(len(cpus), 16)))      # intra-op pool of the few CPU-side torch ops: never wider than the slice
            rec["torch_threads"] = torch.get_num_threads()
        except Exception:
            pass
    return rec
