"""Build-time lint of the generated ISA (no GPU): kernels that reach LDS through GENERIC pointers (flat_load / flat_store) must not
run into an s_barrier with such stores possibly in flight.  Round 6: hipcc 7.2 left the barrier behind the compare-exchange loop of
nms_class_kernel's bitonic sort without any s_waitcnt, and beside another kernel's LDS traffic the sort lost keys -- the cause of
the pipelined plan's rare "Memory access fault by GPU" (csrc/common.h: sm_syncthreads_flat).  The lint
compiles the two sources whose kernels mix flat stores and barriers and runs a small forward dataflow over each kernel's basic blocks
(loops included): no s_barrier may be reachable with a flat store issued since the last s_waitcnt that covers lgkmcnt(0)."""
import os
import re
import shutil
import subprocess

import pytest

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "sipmask_amd", "csrc")


def _kernels(asm):
    """-> (kernel name, [instruction or 'label:' strings]) per kernel"""
    name, body = None, []
    for line in asm.split("\n"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            if name:
                yield name, body
            name, body = m.group(1), []
            continue
        t = line.strip()
        if not name or not t or t.startswith(";"):
            continue
        if re.match(r"^\.LBB\w+:", t):
            body.append(t.split(":")[0] + ":")
        elif not t.startswith("."):
            body.append(t.split(";")[0].strip())
    if name:
        yield name, body


def barriers_with_flat_stores_in_flight(body):
    """forward dataflow over the kernel's basic blocks: `pending` = a flat store / atomic may have been issued since the last
    s_waitcnt that includes lgkmcnt(0) on SOME path to this point (loops included).  Returns the offending s_barrier positions."""
    blocks, cur = [], dict(label=None, ins=[], start=0)
    for k, ins in enumerate(body):
        if ins.endswith(":"):
            if cur["ins"] or cur["label"] is not None:
                blocks.append(cur)
            cur = dict(label=ins[:-1], ins=[], start=k + 1)
            continue
        cur["ins"].append((k, ins))
        if ins.startswith(("s_branch", "s_cbranch", "s_endpgm")):
            blocks.append(cur)
            cur = dict(label=None, ins=[], start=k + 1)
    if cur["ins"] or cur["label"] is not None:
        blocks.append(cur)
    index = {b["label"]: i for i, b in enumerate(blocks) if b["label"] is not None}
    succ = []
    for i, b in enumerate(blocks):
        last = b["ins"][-1][1] if b["ins"] else ""
        out = []
        if last.startswith(("s_branch", "s_cbranch")):
            tgt = last.split()[-1]
            if tgt in index:
                out.append(index[tgt])
        if not last.startswith(("s_branch", "s_endpgm")) and i + 1 < len(blocks):
            out.append(i + 1)
        succ.append(out)
    state_in = [False] * len(blocks)
    bad, changed = set(), True
    while changed:
        changed = False
        for i, b in enumerate(blocks):
            pending = state_in[i]
            for k, ins in b["ins"]:
                if ins.startswith(("flat_store", "flat_atomic")):
                    pending = True
                elif ins.startswith("s_waitcnt") and "lgkmcnt(0)" in ins:
                    pending = False
                elif ins == "s_barrier" and pending:
                    bad.add(k)
            for j in succ[i]:
                if pending and not state_in[j]:
                    state_in[j] = changed = True
    return sorted(bad)


_DEFAULT = ["detect.hip", "rle.hip"]
# SIPMASK_LINT_ALL=1: every source of the library (the scan DESIGN.md section 6 reports; several minutes of hipcc)
_SOURCES = (sorted(f for f in os.listdir(CSRC) if f.endswith(".hip")) if os.environ.get("SIPMASK_LINT_ALL") == "1" else _DEFAULT)


@pytest.mark.parametrize("src", _SOURCES)
def test_no_barrier_with_flat_lds_stores_in_flight(src, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    out = tmp_path / (src + ".s")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", "-I", CSRC,
                    os.path.join(CSRC, src), "-o", str(out)], check=True, capture_output=True, timeout=600)
    bad, seen = [], 0
    for name, body in _kernels(out.read_text()):
        if not any(i.startswith("flat_store") for i in body) or "s_barrier" not in body:
            continue
        seen += 1
        for k in barriers_with_flat_stores_in_flight(body):
            bad.append((name, k, body[max(0, k - 6):k + 1]))
    if src in _DEFAULT:
        assert seen > 0, "the lint no longer sees a kernel with flat stores and barriers in %s: retire or retarget it" % src
    assert not bad, "s_barrier with flat stores possibly in flight:\n" + "\n".join("%s @%d: %s" % b for b in bad)


def test_lint_flags_the_loop_that_lost_keys():
    """the shape of the round-6 miscompile: stores at the end of a loop body, the next step's barrier at the loop head"""
    body = ["s_mov_b32 s0, 0", ".LBB0_1:", "s_barrier", ".LBB0_2:", "flat_load_dwordx2 v[0:1], v[2:3]",
            "s_waitcnt vmcnt(0) lgkmcnt(0)", "flat_store_dwordx2 v[2:3], v[0:1]", "s_cbranch_scc1 .LBB0_2",
            "s_cbranch_scc0 .LBB0_1", "s_endpgm"]
    assert barriers_with_flat_stores_in_flight(body) == [2]
    fixed = body[:2] + ["s_waitcnt vmcnt(0) lgkmcnt(0)"] + body[2:]
    assert barriers_with_flat_stores_in_flight(fixed) == []
