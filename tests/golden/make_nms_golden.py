"""Generate tests/golden/nms_kat.json from the reference's own NMS tests.

Run in the build container (needs /root/reference, which does NOT exist on the
GPU box):  python tests/golden/make_nms_golden.py
Sources (numbers only, parsed with ast -- no reference code is copied):
  B/tests/test_nms.py:16-58   5 boxes x 5 thresholds -> exact keep sets
  B/tests/test_nms.py:60-221  53 boxes, thr 0.5 -> 26 exact keep indices
  M/mmdet/ops/nms/nms_wrapper.py:25-34  7 dets, thr 0.7 -> 3 kept (doctest)
  M/tests/test_nms.py:17-26             4 dets, thr 0.7 -> 3 kept
  M/mmdet/core/bbox/geometry.py:22-44   3x3 IoU(+1) matrix to 4 dp (doctest)
"""
import ast
import json
import os
import re

REF = "/root/reference"


def _np_array_literals(fn_node):
    """All list literals passed to np.array(...) inside a function, in order."""
    out = []
    for node in ast.walk(fn_node):
        if isinstance(node, ast.Call) and getattr(node.func, "attr", "") == "array" and node.args:
            try:
                out.append(ast.literal_eval(node.args[0]))
            except ValueError:
                pass
    return out


def main():
    src = open(os.path.join(REF, "SipMask-benchmark/tests/test_nms.py")).read()
    tree = ast.parse(src)
    fns = {n.name: n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)}
    kat = {}
    # --- case 1
    f = fns["test_nms_cpu"]
    arrs = _np_array_literals(f)
    flat = arrs[0]
    assigns = {t.id: ast.literal_eval(n.value) for n in ast.walk(f) if isinstance(n, ast.Assign)
               for t in n.targets if isinstance(t, ast.Name) and t.id in ("test_thresh", "gt_indices")}
    kat["caffe2_5box"] = dict(dets=[flat[i:i + 5] for i in range(0, len(flat), 5)],
                              thresh=assigns["test_thresh"], keep=assigns["gt_indices"],
                              source="SipMask-benchmark/tests/test_nms.py:16-58")
    # --- case 2
    f = fns["test_nms1_cpu"]
    arrs = _np_array_literals(f)
    named = {}
    for n in ast.walk(f):
        if isinstance(n, ast.Assign) and isinstance(n.targets[0], ast.Name):
            named[n.targets[0].id] = n.value
    def lit(name):
        node = named[name]
        for sub in ast.walk(node):
            if isinstance(sub, ast.Call) and getattr(sub.func, "attr", "") in ("tensor", "array"):
                return ast.literal_eval(sub.args[0])
        return ast.literal_eval(node)
    kat["boxes53"] = dict(boxes=lit("boxes"), scores=lit("scores"), thresh=0.5, keep=lit("gt_indices"),
                          source="SipMask-benchmark/tests/test_nms.py:60-221")
    # --- doctest vectors (numbers transcribed by regex from the docstrings)
    w = open(os.path.join(REF, "SipMask-mmdetection/mmdet/ops/nms/nms_wrapper.py")).read()
    doc = w[w.index("Example:"):w.index('"""', w.index("Example:"))]
    rows = re.findall(r"\[([0-9.,\s]+)\]", doc)
    dets = [[float(v) for v in r.split(",")] for r in rows if r.count(",") == 4]
    kat["wrapper_doctest"] = dict(dets=dets[:7], thresh=0.7, n_keep=3,
                                  source="SipMask-mmdetection/mmdet/ops/nms/nms_wrapper.py:25-34")
    t = open(os.path.join(REF, "SipMask-mmdetection/tests/test_nms.py")).read()
    seg = t[t.index("base_dets"):t.index("# CPU can handle")]
    rows = re.findall(r"\[([0-9.,\s]+)\]", seg)
    kat["mmdet_test4"] = dict(dets=[[float(v) for v in r.split(",")] for r in rows][:4], thresh=0.7, n_keep=3,
                              source="SipMask-mmdetection/tests/test_nms.py:17-26")
    kat["iou_doctest"] = dict(
        bboxes1=[[0, 0, 10, 10], [10, 10, 20, 20], [32, 32, 38, 42]],
        bboxes2=[[0, 0, 10, 20], [0, 10, 10, 19], [10, 10, 20, 20]],
        iou=[[0.5238, 0.0500, 0.0041], [0.0323, 0.0452, 1.0000], [0.0, 0.0, 0.0]],
        source="SipMask-mmdetection/mmdet/core/bbox/geometry.py:22-44")
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nms_kat.json")
    json.dump(kat, open(out, "w"), indent=0)
    print("wrote", out, {k: (len(v.get("dets", v.get("boxes", []))),) for k, v in kat.items()})


if __name__ == "__main__":
    main()
