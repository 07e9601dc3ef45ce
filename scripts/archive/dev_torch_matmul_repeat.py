import torch, subprocess
out = subprocess.run(["rocm-smi", "--showserial"], capture_output=True, text=True).stdout
print([ln.split(":")[-1].strip() for ln in out.splitlines() if "Serial Number" in ln])
torch.manual_seed(0)
for (M, N, K) in [(2048, 3072, 512), (2048, 3072, 256), (4096, 2304, 768), (8192, 4096, 2048)]:
    x = (torch.randn(M, K, device="cuda") * 0.5).bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
    outs = [(x @ w.t()).clone() for _ in range(40)]
    print(M, N, K, "torch.matmul launches that differ from the first:", sum(not torch.equal(outs[0], o) for o in outs[1:]))
