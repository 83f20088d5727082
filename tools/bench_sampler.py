"""Dev tool (GPU box): throughput of the GPU-resident episode sampler at the bench shapes, table sized like iNat-Anim."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from fumi_amd.dataset.gpu_sampler import GpuEpisodeSampler
dev = torch.device("cuda:0")
n_img, C, D, L = 195000, 675, 2048, 128
g = torch.Generator(device=dev).manual_seed(0)
images = torch.randn(n_img, D, device=dev, generator=g)
coi = np.random.RandomState(0).randint(0, C, n_img)
tokens = torch.randint(1, 20000, (C, L), device=dev, generator=g)
smp = GpuEpisodeSampler(images, coi, tokens, num_ways=5, num_shots=5, num_shots_test=32, batch_size=32, seed=1)
for i in range(10): smp.batch(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 200
for i in range(n): b = smp.batch(100 + i)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
moved = 32 * 185 * D * 4 * 2
print(f"table {images.numel() * 4 / 1e9:.2f} GB; {dt * 1e6:.1f} us per meta-batch of 32 episodes ({32 / dt:.0f} episodes/s); "
      f"{moved / dt / 1e12:.2f} TB/s of row traffic (read + write)")
