#!/usr/bin/env python
"""Static check of the software-managed MFMA result hazard on the generated gfx950 ISA.

The recurrent kernels issue v_mfma_f32_16x16x32_bf16 from inline asm (csrc/mfma_rec.h), which the compiler's hazard
recogniser cannot see.  Every block carries its own cover (s_nop 7 + s_nop 0 = 9 wait states) EXCEPT the "chained" forms
(LAST = false), which rely on this property of the surrounding code: until 9 wait states have passed after an MFMA, the only
instructions that touch its destination registers are MFMAs that use them as vDst / SrcC (interlocked by the hardware).
This script compiles the kernels to assembly and verifies that property for EVERY MFMA on every path (it follows branches):

    python tools/mfma_hazard_check.py            # attn_cluster.hip lstm_cluster.hip lstm.hip
    python tools/mfma_hazard_check.py file.s     # an existing device assembly file

It also verifies the LEADING hazard of blocks without their s_nop 2 (FIRST = false): no VALU instruction writes an MFMA source
operand (SrcA / SrcB / SrcC, VGPR or AGPR) within 2 wait states in front of the MFMA, on any path into it.

Wait states are counted conservatively: one per instruction, N + 1 for s_nop N.  Exit status 1 on a violation."""
import os, re, subprocess, sys, tempfile

NEED = 9
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "self-attention-tacotron_amd", "csrc")
DEFAULT = ["attn_cluster.hip", "lstm_cluster.hip", "lstm.hip"]
NEED_PRE = 2
AREG = re.compile(r"\ba(\d+)\b|\ba\[(\d+):(\d+)\]")


def aregs(text):
    out = set()
    for m in AREG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def states(ins):
    return int(ins[1][0], 0) + 1 if ins[0] == "s_nop" else 1


def check_leading(name, ins, labels, path, i):
    """VALU write of a source operand of the MFMA at index i within NEED_PRE wait states, on any path into it"""
    mn, ops, ln = ins[i]
    src_v, src_a = vregs(" ".join(ops[1:])), aregs(" ".join(ops[1:]))
    at_label = {}
    for lab, idx in labels.items():
        at_label.setdefault(idx, []).append(lab)
    bad = 0
    stack, seen = [(i, 0)], set()          # (index whose predecessors are examined, wait states already between)
    while stack:
        k, ws = stack.pop()
        if ws >= NEED_PRE or (k, ws) in seen:
            continue
        seen.add((k, ws))
        preds = []
        if k > 0 and ins[k - 1][0] not in ("s_branch", "s_endpgm", "s_setpc_b64"):
            preds.append(k - 1)
        for lab in at_label.get(k, []):
            preds += [b for b, (m2, o2, _) in enumerate(ins) if m2.startswith(("s_branch", "s_cbranch")) and o2 and o2[0] == lab]
        for j in preds:
            m2, o2, l2 = ins[j]
            if m2.startswith("v_") and not m2.startswith("v_mfma") and o2:
                if (vregs(o2[0]) & src_v) or (aregs(o2[0]) & src_a):
                    print("%s:%d: %s: `%s %s` writes a source of the MFMA at line %d only %d wait states ahead (need %d)"
                          % (path, l2, name, m2, ", ".join(o2), ln, ws, NEED_PRE))
                    bad += 1
            stack.append((j, ws + states(ins[j])))
    return bad


REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def vregs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def parse(path):
    """-> list of functions: (name, instrs[(mnemonic, operand list, line no)], labels{name: index})"""
    funcs, cur, labels, name = [], None, None, None
    for ln, raw in enumerate(open(path), 1):
        line = raw.split(";")[0].strip()
        if not line:
            continue
        m = re.match(r"^([A-Za-z_.$][\w.$]*):$", line)
        if not m and line.startswith("."):
            continue                       # assembler directive
        if m:
            lab = m.group(1)
            if not lab.startswith(".L") and not lab.startswith("BB"):
                if cur:
                    funcs.append((name, cur, labels))
                name, cur, labels = lab, [], {}
            elif cur is not None:
                labels[lab] = len(cur)
            continue
        if cur is None:
            continue
        parts = line.split(None, 1)
        ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
        cur.append((parts[0], ops, ln))
    if cur:
        funcs.append((name, cur, labels))
    return funcs


def check_function(name, ins, labels, path):
    bad = 0
    n_mfma = 0
    for i, (mn, ops, ln) in enumerate(ins):
        if not mn.startswith("v_mfma"):
            continue
        n_mfma += 1
        bad += check_leading(name, ins, labels, path, i)
        dst = vregs(ops[0])
        # walk every path from i + 1 until NEED wait states have passed
        stack, seen = [(i + 1, 0)], set()
        while stack:
            j, ws = stack.pop()
            while j < len(ins) and ws < NEED:
                if (j, ws) in seen:
                    break
                seen.add((j, ws))
                m2, o2, l2 = ins[j]
                if m2.startswith("v_mfma"):
                    srcab = vregs(" ".join(o2[1:3]))
                    if srcab & dst:
                        print("%s:%d: %s reads the result of the MFMA at line %d as SrcA/B after %d wait states" % (path, l2, name, ln, ws))
                        bad += 1
                    d2 = vregs(o2[0])
                    if d2 & dst and d2 != dst:
                        print("%s:%d: %s partially overlapping MFMA destination (line %d)" % (path, l2, name, ln))
                        bad += 1
                    ws += 1
                elif m2 == "s_nop":
                    ws += int(o2[0], 0) + 1
                elif m2 in ("s_endpgm",):
                    break
                elif m2 == "s_branch":
                    tgt = labels.get(o2[0])
                    if tgt is None:
                        break
                    j = tgt; ws += 1
                    continue
                elif m2.startswith("s_cbranch"):
                    tgt = labels.get(o2[0])
                    if tgt is not None:
                        stack.append((tgt, ws + 1))
                    ws += 1
                else:
                    if vregs(" ".join(o2)) & dst:
                        print("%s:%d: %s: `%s %s` touches the result of the MFMA at line %d after %d wait states (need %d)"
                              % (path, l2, name, m2, ", ".join(o2), ln, ws, NEED))
                        bad += 1
                    ws += 1
                j += 1
    return n_mfma, bad


def compile_to_asm(src, out):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "--cuda-device-only", "-S",
           src, "-o", out, "-I", os.path.join(ROOT, "include"), "-I", CSRC] + os.environ.get("SATT_EXTRA_FLAGS", "").split()
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.exit("hipcc failed: %s\n%s" % (" ".join(cmd), r.stderr))


def main():
    args = sys.argv[1:]
    paths = []
    tmp = None
    if args and all(a.endswith(".s") for a in args):
        paths = args
    else:
        tmp = tempfile.mkdtemp(prefix="mfma_hz_")
        for f in (args or DEFAULT):
            src = f if os.path.exists(f) else os.path.join(CSRC, f)
            out = os.path.join(tmp, os.path.basename(src).replace(".hip", ".s"))
            compile_to_asm(src, out)
            paths.append(out)
    total = bad = 0
    for p in paths:
        for name, ins, labels in parse(p):
            n, b = check_function(name, ins, labels, os.path.basename(p))
            total += n; bad += b
            if n:
                print("%-110s %5d MFMAs  %s" % (name[:110], n, "ok" if b == 0 else "%d VIOLATIONS" % b))
    print("checked %d MFMA instructions: %s" % (total, "no hazard" if bad == 0 else "%d violations" % bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
