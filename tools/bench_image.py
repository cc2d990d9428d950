"""Timing of the device image transform (diagnostic): B decoded images of h x w -> [B, 3, S, S]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mvuld_amd.data.image_ingest import DeviceImageTransform

dev = torch.device("cuda:0")
for B, h, w, S in [(32, 600, 800, 448), (32, 1200, 1600, 448), (32, 448, 448, 448), (256, 600, 800, 448)]:
    x = torch.randint(0, 256, (B, h, w, 3), dtype=torch.uint8, device=dev)
    tf = DeviceImageTransform(S)
    for _ in range(3):
        tf(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        tf(x)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    by = B * (h * w * 3 + (2 * h * S * 3 if w != S else 0) + 3 * S * S * 4)
    print(f"B={B} {h}x{w} -> {S}: {us:8.1f} us  {by / us / 1e3:7.1f} GB/s (bytes in + temp + out)  {B / us * 1e6:9.0f} images/s")
