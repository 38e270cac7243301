#!/usr/bin/env python
"""Issue-floor table of the recurrent kernels: dynamic instruction counts per wave and decoder step from the
SQ_INSTS_* counters (tools/pmc_insts.sh -> gpurun_out/insts/insts.txt) x the measured issue cost, beside the measured step time.

  python tools/issue_floor.py gpurun_out/insts/insts.txt <fwd us/step> <bwd us/step> [steps_profiled=3] > profiles/r03_attn_issue_floor.txt

Issue model: a CU offers each SIMD one issue turn every 4 clocks and a wave issues at most one instruction per turn, so the
floor of a wave's step is 4 clk x (its instructions) at 2.4 GHz - confirmed by the counters themselves: SQ_ACTIVE_INST_ANY /
instructions = 1.07 quad-cycles.  (A lone wave issues VALU instructions only every other turn, profiles/r02_valu_issue_rates.txt:
the column "floor, VALU x2" prices that.)"""
import re
import sys


def main():
    path = sys.argv[1]
    t_meas = {"attn_cluster_fwd_k": float(sys.argv[2]), "attn_cluster_bwd_k": float(sys.argv[3])}
    nsteps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    Td, waves = 400, 32 * 4 * 8
    cur, data = None, {}
    for ln in open(path):
        m = re.match(r"counter (\S+)", ln)
        if m:
            cur = m.group(1); continue
        m = re.search(r"\d+(\w+_k)\w*\s+n=\s*(\d+)\s+sum=\s*([\d.]+)", ln)
        if m and cur:
            data.setdefault(m.group(1), {})[cur] = float(m.group(3)) / nsteps / waves / Td
    cats = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM",
            "SQ_INSTS_MFMA"]
    print("# dynamic instructions per wave and decoder step (B=32, Ti=160, Td=400, cluster of 4 x 8 waves; averages over the 1024 waves)")
    print("%-22s %s %9s %9s %11s %11s %7s" % ("kernel", " ".join("%9s" % c.replace("SQ_INSTS_", "") for c in cats), "total", "floor us", "VALU x2 us", "measured us", "ratio"))
    for k in ("attn_cluster_fwd_k", "attn_cluster_bwd_k"):
        d = data.get(k, {})
        v = [d.get(c, 0.0) for c in cats]
        valu = v[0] - v[7] if v[0] > v[7] else v[0]      # SQ_INSTS_VALU includes the MFMA instructions
        other = sum(v[1:])
        floor, floor2 = 4.0 * (valu + other) / 2400.0, (8.0 * valu + 4.0 * other) / 2400.0
        print("%-22s %s %9.0f %9.2f %11.2f %11.2f %7.2f" % (k, " ".join("%9.0f" % x for x in v), valu + other, floor, floor2, t_meas[k], t_meas[k] / floor))
    print("# wave-cycle split of the same kernels, all in quad-cycles (SQ_WAVE_CYCLES = resident; SQ_ACTIVE_INST_ANY = spent on an instruction;")
    print("# SQ_WAIT_INST_ANY = ready but waiting for an issue turn; the rest = parked at barriers, exchange polls and s_waitcnt):")
    for k in ("attn_cluster_fwd_k", "attn_cluster_bwd_k"):
        d = data.get(k, {})
        wc, act, wi = d.get("SQ_WAVE_CYCLES", 0), d.get("SQ_ACTIVE_INST_ANY", 0), d.get("SQ_WAIT_INST_ANY", 0)
        if wc:
            print("%-22s quad-cycles/step %8.0f  working on instructions %4.1f %%  waiting for issue %4.1f %%  parked (barriers, polls, waitcnt) %4.1f %%"
                  % (k, wc, 100 * act / wc, 100 * wi / wc, 100 * (1 - (act + wi) / wc)))


if __name__ == "__main__":
    main()
