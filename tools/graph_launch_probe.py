"""What does a hipGraph replay cost on this stack?  N tiny dependent kernels (in-place adds on a 4 KB tensor) and N larger ones
(64 MB adds: ~20 us each) launched eagerly and replayed from a captured graph: host time to issue and total time, per kernel.
Run under different runtime flags (DEBUG_CLR_GRAPH_PACKET_CAPTURE, HIP_FORCE_DEV_KERNARG ...) from the shell.  One JSON line."""
import json, os, sys, time
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
dev = torch.device("cuda:0")
rec = {"n_kernels": n, "env": {k: os.environ[k] for k in os.environ if k.startswith(("DEBUG_", "HIP_FORCE", "AMD_DIRECT"))}}
for tag, numel in (("tiny_4KB", 1024), ("big_64MB", 16 << 20)):
    x = torch.zeros(numel, device=dev)

    def body():
        for _ in range(n):
            x.add_(1.0)
    body(); torch.cuda.synchronize()
    t0 = time.time(); body(); th = time.time() - t0; torch.cuda.synchronize(); tt = time.time() - t0
    rec[tag + "_eager_host_us_per_kernel"] = round(th / n * 1e6, 2)
    rec[tag + "_eager_total_us_per_kernel"] = round(tt / n * 1e6, 2)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        body()
    g.replay(); torch.cuda.synchronize()
    reps = 5
    t0 = time.time()
    for _ in range(reps):
        g.replay()
    th = time.time() - t0; torch.cuda.synchronize(); tt = time.time() - t0
    rec[tag + "_graph_host_us_per_kernel"] = round(th / n / reps * 1e6, 2)
    rec[tag + "_graph_total_us_per_kernel"] = round(tt / n / reps * 1e6, 2)
print(json.dumps(rec))
