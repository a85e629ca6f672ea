#!/usr/bin/env python3
"""Secondary metric: images/s of the VGG-16 fc7 extractor (001_prepro_img_vgg.lua) on one MI355X.
30.93 GFLOP per image; synthetic He-scaled weights and images.  The timed call is
nvqa_vgg16_fc7 as the reference script would use it: host images in (PCIe), host features out."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    bf16 = len(sys.argv) > 3 and sys.argv[3] == "bf16"
    pkg = ge.load_package()
    v = pkg.binding.Vgg16(0, 1, 224, max_batch=n)
    rng = np.random.default_rng(0)
    chans = [64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512, 512]
    parts, cin = [], 3
    for c in chans:
        parts += [rng.standard_normal(c * cin * 9).astype(np.float32) * np.sqrt(2.0 / (cin * 9)), np.zeros(c, np.float32)]
        cin = c
    for k in (25088, 4096):
        parts += [rng.standard_normal(4096 * k).astype(np.float32) * np.sqrt(2.0 / k), np.zeros(4096, np.float32)]
    v.set_weights(np.concatenate(parts))
    if bf16:
        v.set_precision(1)
    x = rng.uniform(-120, 130, (n, 3, 224, 224)).astype(np.float32)
    v.fc7(x)
    t0 = time.perf_counter()
    for _ in range(iters):
        f = v.fc7(x)
    dt = (time.perf_counter() - t0) / iters
    gflop = 30.93 * n
    print(json.dumps({"metric": "VGG-16 fc7 images/s (%s, batch %d, host in / host out)" % ("bf16 operands" if bf16 else "fp32", n),
                      "value": round(n / dt, 1), "unit": "images/s", "ms_per_batch": round(dt * 1e3, 2),
                      "tflops": round(gflop / dt / 1e3, 1), "frac_of_mfma_peak": round(gflop / dt / 1e3 / (2500.0 if bf16 else 157.3), 3),
                      "nonzero_features": float((f > 0).mean())}))
    v.close()


if __name__ == "__main__":
    main()
