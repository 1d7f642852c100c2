"""How many __amd_rocclr_copyBuffer launches does ONE replay of a captured HIP graph cost on this stack, as a function of the number of
kernel nodes?  Run under rocprofv3 --kernel-trace and count by kernel name (tools/graph_copy_probe.sh).

    python tools/graph_copy_probe.py N_NODES REPLAYS
"""
import sys
import torch

n, replays = int(sys.argv[1]), int(sys.argv[2])
x = torch.zeros(1024, device='cuda')
y = torch.zeros(1024, device='cuda')
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        x.add_(1.0)
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for i in range(n):
        x.add_(1.0)            # one elementwise kernel node each
torch.cuda.synchronize()
for _ in range(replays):
    g.replay()
torch.cuda.synchronize()
print('nodes', n, 'replays', replays, 'x[0]', float(x[0]))
