#!/usr/bin/env python
"""Static check of the generated gfx950 ISA of the recurrent kernels for the prefetch-ring drain of round 4: a run of
`s_waitcnt vmcnt(N) ; v_mov` pairs in front of a loop's back-edge branch means the compiler copies loop-carried registers that are
the destinations of loads still in flight - every trip of the loop then waits for ALL of them (vmcnt counts down to 0 across the
run), i.e. the software prefetch overlaps nothing.  Cause: a ring slot refilled while its old value is still live (the loads
get registers of their own).  Fix: refill behind the slot's last use, branch-free.  (csrc/lstm.hip, csrc/lstm_cluster.hip)

python tools/vmcnt_drain_check.py [file.hip ...]   -> exit status 1 if a kernel has such a run of >= MIN_PAIRS pairs."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "self-attention-tacotron_amd", "csrc")
FILES = ["lstm.hip", "lstm_cluster.hip", "attn_cluster.hip", "attn_rnn.hip"]
MIN_PAIRS = 3


def back_edge_follows(lines, k, labels):
    """the first branch behind line k (same basic-block run: up to the next label) targets a label DEFINED EARLIER in the listing -
    the run sits in front of a loop's back edge (the pattern of round 4).  The same wait / copy run in front of a FORWARD branch is
    a different thing: e.g. the result registers of an exchange poll zeroed in front of its `workgroup is dead` test (r5) - the
    waits there are met long before (and the first poll is better off behind the wave's publishing stores, measured)"""
    for j in range(k, min(k + 12, len(lines))):
        t = lines[j].strip()
        if re.match(r"^[.\w$]+:", t):
            return False
        m = re.match(r"^s_c?branch\w*\s+(\S+)", t)
        if m:
            return labels.get(m.group(1), 1 << 60) < j
    return False


def runs(lines):
    """[(function, first line, pairs)] of maximal runs of (s_waitcnt vmcnt(N) [v_mov...]) groups whose counts fall to 0"""
    labels = {}
    for j, l in enumerate(lines):
        m = re.match(r"^([.\w$]+):", l.strip())
        if m:
            labels.setdefault(m.group(1), j)
    out, fn, i, n = [], None, 0, len(lines)
    while i < n:
        t = lines[i].strip()
        m = re.match(r"^(_Z\w+):", t)
        if m:
            fn = m.group(1)
        if t.startswith("s_waitcnt vmcnt("):
            k, pairs, last = i, 0, None
            while k < n and lines[k].strip().startswith("s_waitcnt vmcnt("):
                cnt = int(re.match(r"s_waitcnt vmcnt\((\d+)\)", lines[k].strip()).group(1))
                k += 1
                movs = 0
                while k < n and lines[k].strip().startswith(("v_mov_b32", "v_mov_b64", "v_accvgpr_write")):
                    k += 1; movs += 1
                if movs == 0:
                    break
                pairs += 1; last = cnt
            if pairs >= MIN_PAIRS and last == 0 and back_edge_follows(lines, k, labels):
                out.append((fn, i + 1, pairs))
            i = max(k, i + 1)
            continue
        i += 1
    return out


def check_file(path):
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-gpu-rdc", "-S", "--cuda-device-only", "-o", asm, path],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-2000:])
        return runs(open(asm).read().split("\n"))


if __name__ == "__main__":
    files = sys.argv[1:] or [os.path.join(CSRC, f) for f in FILES]
    bad = 0
    for f in files:
        for fn, line, pairs in check_file(f):
            print("%s: %s: line %d: %d wait/copy pairs down to vmcnt(0)" % (os.path.basename(f), fn, line, pairs))
            bad += 1
    print("no prefetch-ring drain found" if bad == 0 else "%d drain(s)" % bad)
    sys.exit(1 if bad else 0)
