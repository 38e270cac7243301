#!/usr/bin/env python
"""gpurun_out/final (raw output of tools/final_measure.sh on the GPU box) -> profiles/rNN_* (the tracked summaries).
usage: python tools/collect_profiles.py r02"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "final")
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
P = lambda name: os.path.join(ROOT, "profiles", "%s_%s" % (tag, name))


def newest(pattern):
    fs = glob.glob(os.path.join(SRC, pattern), recursive=True)
    return max(fs, key=os.path.getmtime) if fs else None


def kernel_stats(steps=7):
    f = newest("trace/**/*kernel_stats.csv")
    rows = list(csv.DictReader(open(f)))
    tot = sum(int(r["TotalDurationNs"]) for r in rows)
    out = ["# rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-decode (%d train steps"
           % steps, "# incl. warm-up; one MI355X, B=32, Ti=160, Tm=800, bf16).  Kernel time summed over the 4 streams = %.1f ms per step"
           % (tot / steps / 1e6), "# (the layers overlap; the sum contains in-kernel waits and the 1-thread streamOpsWait kernels).",
           "%-78s %7s %10s %11s %7s" % ("kernel", "calls", "total ms", "avg us", "share")]
    for r in rows[:60]:
        name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        name = name.split("(")[0] if len(name) > 78 else name
        out.append("%-78s %7d %10.3f %11.2f %6.2f%%" % (name[:78], int(r["Calls"]), int(r["TotalDurationNs"]) / 1e6,
                                                        float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
    open(P("rocprofv3_kernel_stats.txt"), "w").write("\n".join(out) + "\n")


def decode_kernel_stats():
    f = newest("megatrace/**/*kernel_stats.csv")
    if not f:
        return
    rows = list(csv.DictReader(open(f)))
    out = ["# rocprofv3 --kernel-trace --stats -- python tools/bench_infer.py --steps 192 (config 5: B=1, Ti=100, bf16; warm-up utterance +",
           "# timed utterance = 2 x 192 decoder steps in launches of <= DecodeSession.MEGA_STEPS steps of the persistent step kernel; encoder and memory kernels below it)",
           "%-78s %7s %10s %11s %7s" % ("kernel", "calls", "total ms", "avg us", "share")]
    for r in rows[:25]:
        name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        name = name.split("(")[0] if len(name) > 78 else name
        out.append("%-78s %7d %10.3f %11.2f %6.2f%%" % (name[:78], int(r["Calls"]), int(r["TotalDurationNs"]) / 1e6,
                                                        float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
    mega = [r for r in rows if "dec_mega2_k" in r["Name"]]
    if mega:
        out.append("# persistent step kernel: %d launches, %.2f us per launch, %.2f us per decoder step (total / 384 steps)"
                   % (int(mega[0]["Calls"]), float(mega[0]["AverageNs"]) / 1e3, int(mega[0]["TotalDurationNs"]) / 384e3))
    open(P("decode_kernel_stats.txt"), "w").write("\n".join(out) + "\n")


def main():
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    kernel_stats()
    decode_kernel_stats()
    fdb, wdb = newest("pmc_FETCH_SIZE/**/*.db"), newest("pmc_WRITE_SIZE/**/*.db")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_pmc_traffic.py"), fdb, wdb, P("pmc_traffic.json")],
                          stdout=subprocess.DEVNULL)
    for src, dst in (("gemm_roofline.txt", "gemm_roofline.txt"), ("gemm_paths.txt", "gemm_paths.txt"),
                     ("phase_marks.txt", "step_phases.txt"), ("bench_tacotron.json", "bench_tacotron.json"),
                     ("infer.json", "infer_config5.json"), ("infer_b8.json", "infer_config5_batch8.json"), ("infer_b2.json", "infer_config5_batch2.json"),
                     ("bench.json", "bench.json"), ("gpu_tests.log", "gpu_tests.log"), ("bench_vctk.json", "bench_vctk.json"),
                     ("phase_marks_rccl.txt", "step_phases_one_rank_rccl.txt"),
                     ("bench_rccl_one_rank.json", "bench_one_rank_rccl.json"), ("decode_timeline.txt", "decode_timeline.txt"),
                     ("attn_loop_phases.txt", "attn_loop_phases.txt"), ("parity_bench_workloads.log", "parity_bench_workloads.log"),
                     ("insts.txt", "pmc_instruction_counts.txt"), ("encoder_lstm.txt", "encoder_lstm.txt"),
                     ("small_attn.txt", "small_attn.txt"), ("flash.txt", "flash.txt"), ("decode_phases.txt", "decode_phases.txt"),
                     ("infer_graph_path.json", "infer_config5_graph_path.json"), ("residency_sweep.txt", "residency_sweep.txt"),
                     ("decode_golden.log", "decode_golden.log"), ("parity_frozen_oracle.log", "parity_frozen_oracle.log"), ("host_enqueue.txt", "host_enqueue.txt"), ("lds_poison_sweep.txt", "lds_poison_sweep_final.txt")):
        if not os.path.exists(os.path.join(SRC, src)):
            continue
        lines = [ln for ln in open(os.path.join(SRC, src)).read().splitlines(True) if "amdgpu.ids" not in ln]
        open(P(dst), "w").write("".join(lines))
    b = json.load(open(P("bench.json")))
    print("%s: %.3f ms/step, %.0f mel-frames/s, decode %.1f us/step" % (tag, b["ms_per_step"], b["value"], 1e3 * b["decode"]["ms_per_step"]))


if __name__ == "__main__":
    main()
