#!/usr/bin/env python
"""EXPERIMENT (needs `make -C sipmask_amd/csrc EXPERIMENTS=1` for the ablation rows): shader clock and socket power while the
patch-conv tile runs back to back -- rocm-smi sampled from a background thread during ~2 s loops of (a) 30 tiles, (b) 255
tiles, (c) 255 tiles without LDS-DMA, (d) 255 tiles without MFMA / fragment reads."""
import os, re, subprocess, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sipmask_amd import hip_ops as H
UNIFORM, NO_DMA, NO_MFMA = 0x4000, 0x200, 0x100
dev = torch.device("cuda")
w = torch.randn(256, 256, 3, 3, device=dev) / 48
wp, cpp = H.prep_conv_weight_patch(w)


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=10).stdout
    except Exception as e:                      # noqa: BLE001
        return None, None, str(e)
    sclk = re.findall(r"sclk clock level:?\s*\S*\s*\(?(\d+)Mhz", out)
    pw = re.findall(r"(?:Average|Current Socket) Graphics Package Power \(W\):\s*([\d.]+)", out)
    return (int(sclk[0]) if sclk else None), (float(pw[0]) if pw else None), out


s0 = smi()
print("idle: sclk %s MHz, power %s W" % (s0[0], s0[1]))
if s0[0] is None:
    print(s0[2][:1500])
for name, B, fl in (("30 tiles, full kernel", 2, UNIFORM), ("255 tiles, full kernel", 17, UNIFORM),
                    ("255 tiles, no LDS-DMA", 17, UNIFORM | NO_DMA), ("255 tiles, no MFMA", 17, UNIFORM | NO_MFMA),
                    ("255 tiles, zero operands", 17, UNIFORM)):
    sizes = [(30, 126)]
    lv = H.Levels(B, sizes)
    x = (torch.randn(lv.rows, 256, device=dev) * 0.5).to(torch.bfloat16)
    if "zero" in name:
        x.zero_()
    y = torch.zeros(lv.rows, 256, dtype=torch.bfloat16, device=dev)
    d = H.make_conv_desc(B, sizes, sizes, lv.row0, lv.row0, 256, 256, cpp, 3, 1, 1, 256, 256, flags=fl)
    samples, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            c, p, _ = smi()
            samples.append((c, p))
    th = threading.Thread(target=poll)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 30000
    th.start()
    e0.record()
    for _ in range(n):
        H.conv3x3_patch(d, x, wp, None, y)
    e1.record()
    torch.cuda.synchronize()
    stop.set()
    th.join()
    us = e0.elapsed_time(e1) / n * 1e3
    cl = [c for c, _ in samples if c]
    pws = [p for _, p in samples if p]
    print("%-28s %.1f us per launch over %.1f s; sclk samples (MHz) min %s median %s max %s; power (W) median %s  [n=%d]" % (
        name, us, us * n / 1e6, min(cl) if cl else None, sorted(cl)[len(cl) // 2] if cl else None, max(cl) if cl else None,
        sorted(pws)[len(pws) // 2] if pws else None, len(samples)), flush=True)
