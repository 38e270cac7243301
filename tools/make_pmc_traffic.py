#!/usr/bin/env python
"""profiles/<out>.json from two rocprofv3 --pmc runs (FETCH_SIZE and WRITE_SIZE, separate passes as the MI355X guide
prescribes): per kernel family, HBM bytes per launch and per train step (counter-collecting runs serialise kernels, so the engine
falls back to one attention launch per pipeline chunk there; bench.py divides the per-step bytes by the launches per
step of the timed run).  Correction per /opt/skills/guides/MI355X_MICROARCH.md (HBM
section): on gfx950 FETCH_SIZE (KB) tallies 128-B read requests as 64 B -> doubled; WRITE_SIZE (KB) is taken as is.
usage: python tools/make_pmc_traffic.py <fetch_dir> <write_dir> <out.json>"""
import glob, json, os, sqlite3, sys

FAMILIES = {"attn_rnn_fwd": "attn_cluster_fwd_k", "attn_rnn_bwd": "attn_cluster_bwd_k", "lstm_cluster_fwd": "lstm_cluster_fwd_k",
            "lstm_cluster_bwd": "lstm_cluster_bwd_k", "enc_lstm_fwd": "lstm_fwd_mfma_k", "enc_lstm_bwd": "lstm_bwd_mfma_k",
            "gemm": "gemm_kernel", "gemm_tile_rk": "gemm_rk_k", "gemm_tile_dw": "gemm_dw_k", "flash_fwd": "flash_fwd_k",
            "flash_bwd": "flash_bwd_k", "attn_param_grads": "attn_param_grads_k"}


def per_kernel(src, cname):
    if os.path.isdir(src):
        src = sorted(glob.glob(os.path.join(src, "**", "*.db"), recursive=True))[-1]
    db = sqlite3.connect(src); cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    pe = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
    pi = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    ids = [r[0] for r in cur.execute("select id from %s where name=?" % pi, (cname,))]
    q = ("select s.kernel_name, count(distinct d.id), sum(e.value) from %s e join %s d on e.event_id=d.event_id "
         "join %s s on d.kernel_id=s.id where e.pmc_id in (%s) group by s.kernel_name" % (pe, kd, ks, ",".join(map(str, ids))))
    return cur.execute(q).fetchall()


def main():
    fetch, write, out = sys.argv[1:4]
    res = {"_note": "HBM bytes per LAUNCH from rocprofv3 PMC (FETCH_SIZE, WRITE_SIZE in separate passes); "
                    "hbm_bytes = 2 * FETCH_SIZE_KB * 1024 + WRITE_SIZE_KB * 1024 (gfx950 FETCH_SIZE correction of the MI355X guide)"}
    fr, wr = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    # train steps profiled = dispatches of the once-per-step optimizer prologue
    sf = sum(c for n, c, _ in fr if "adam_prepare_k" in n); sw = sum(c for n, c, _ in wr if "adam_prepare_k" in n)
    res["_steps_profiled"] = {"fetch_pass": sf, "write_pass": sw}
    for fam, pat in FAMILIES.items():
        f = [(n, c, v) for n, c, v in fr if pat in n]; w = [(n, c, v) for n, c, v in wr if pat in n]
        if not f or not w:
            continue
        nf, vf = sum(c for _, c, _ in f), sum(v for _, _, v in f)
        nw, vw = sum(c for _, c, _ in w), sum(v for _, _, v in w)
        res[fam] = {"launches_profiled": nf, "fetch_kb_raw_per_launch": vf / nf, "write_kb_raw_per_launch": vw / nw,
                    "hbm_bytes_per_launch": 2 * 1024 * vf / nf + 1024 * vw / nw}
        if sf and sw:     # per TRAIN STEP: independent of how many launches the step was split into when profiled
            res[fam]["launches_per_step_profiled"] = nf / sf
            res[fam]["hbm_bytes_per_step"] = 2 * 1024 * vf / sf + 1024 * vw / sw
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
