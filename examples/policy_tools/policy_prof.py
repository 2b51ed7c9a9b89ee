import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dcmrta_amd.policy import AttentionNet
B, A, T = 4096, 20, 50
dev = "cuda:0"
net = AttentionNet().to(dev).eval(); net.assume_no_padding = True
tasks, agents = torch.rand(B, T + 1, 5, device=dev), torch.rand(B, A, 6, device=dev)
mask = torch.rand(B, T + 1, device=dev) < 0.3; mask[:, 0] = False
dt = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": None}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
m = net if dt is None else net.rollout_copy(dt)
for _ in range(5):
    with torch.no_grad():
        m(tasks, agents, mask)
torch.cuda.synchronize()
