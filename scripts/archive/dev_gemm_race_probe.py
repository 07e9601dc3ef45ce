"""Dev-only: is a GEMM the same bits every launch?  x [M, K] . w [N, K]^T repeated; reports how many launches differ from the first and the largest deviation from an fp32 product."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ecg_byte_amd import decoder_ops as ops
torch.manual_seed(0)
tiles = [int(t) for t in os.environ.get("TILES", "0").split(",")]
for tile in tiles:
  ops.set_gemm_tile(tile); print("--- ecgb_set_gemm_tile", tile)
  for (M, N, K) in [(2048, 3072, 512), (2048, 768, 512), (2048, 512, 512), (2048, 512, 1536), (2048, 3072, 256), (2048, 3072, 1024), (2048, 3072, 2048), (4096, 2304, 768), (256, 5056, 512), (32768, 3072, 512)]:
      x = (torch.randn(M, K, device="cuda") * 0.5).bfloat16(); w = (torch.randn(N, K, device="cuda") * 0.05).bfloat16()
      ref = (x.float() @ w.float().t())
      outs = [ops.gemm_nt(x, w).clone() for _ in range(40)]
      torch.cuda.synchronize()
      nd = sum(not torch.equal(outs[0], o) for o in outs[1:])
      err = max(float((o.float() - ref).abs().max()) for o in outs)
      bad_el = max(int((o != outs[0]).sum()) for o in outs)
      print(f"gemm_nt M {M} N {N} K {K}: launches that differ from the first {nd}/39, most elements differing {bad_el}, max |err| {err:.4f} (scale {float(ref.abs().max()):.2f})")

ops.set_gemm_tile(0)
