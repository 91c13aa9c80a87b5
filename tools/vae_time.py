"""VAE decode time (latent -> uint8 image on the device): `python tools/vae_time.py [size] [batch]`."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    from minsdtf_amd.stable_diffusion import StableDiffusion

    sd = StableDiffusion(size, size, jit_compile=True, device=torch.device("cuda:0"))
    sd.image_decoder.load_synthetic(seed=0)
    lat = torch.randn(B, size // 8, size // 8, 4, device="cuda:0")
    for _ in range(3):
        sd.image_decoder.decode_to_uint8(lat)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            sd.image_decoder.decode_to_uint8(lat)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 4)
    print(f"VAE decode {size}x{size} batch {B}: {best:.3f} ms")


if __name__ == "__main__":
    main()
