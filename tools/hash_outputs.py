#!/usr/bin/env python
"""Prints SHA-256 digests of everything a seeded run of the benchmarked plans produces (detections, labels, masks,
GroupNorm statistics, head tensors).  Used to A/B two builds of libsipmask_hip.so for BIT equality:

    python tools/hash_outputs.py > new.txt;  (swap the .so);  python tools/hash_outputs.py > old.txt;  diff new.txt old.txt
"""
import hashlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def digest(t):
    t = t.detach().contiguous().cpu()
    return hashlib.sha256(t.view(torch.uint8).numpy().tobytes()).hexdigest()[:16]


def main():
    from oracle import model as OM
    from sipmask_amd.engine import SipMaskEngine, SubBatchPlan
    dev = torch.device("cuda:0")
    sd = OM.init_state_dict(50, 0)
    g = torch.Generator().manual_seed(7)
    img = torch.randn(4, 3, 800, 1344, generator=g).to(dev)
    for precision in ("bf16", "head_x3"):
        for name, mk in (("subbatch", lambda: SubBatchPlan([SipMaskEngine(sd, 2, (800, 1344), 50, sub_plan=True, precision=precision)
                                                            for _ in range(2)])),
                         ("lanes1", lambda: SipMaskEngine(sd, 4, (800, 1344), 50, precision=precision))):
            eng = mk()
            out = eng.run(img)
            out = eng.run(img)
            torch.cuda.synchronize()
            for k in ("det_bboxes", "det_labels", "ndet", "masks"):
                print(precision, name, k, digest(out[k]))
            for i, e in enumerate(getattr(eng, "engines", [eng])):
                for attr in ("gn_stats", "offsets", "aligned", "reg_out", "basis"):
                    v = getattr(e, attr, None)
                    if torch.is_tensor(v):
                        print(precision, name, "plan%d" % i, attr, digest(v))
                print(precision, name, "plan%d" % i, "deform", getattr(e, "deform_choice", None) and e.deform_choice["kernel"])
            del eng, out
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
