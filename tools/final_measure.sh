#!/bin/bash
# round measurements: run from the repo root on the GPU box (gpurun); raw files under gpurun_out/final, summaries are
# copied into profiles/ by tools/collect_profiles.py
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for cnt in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $cnt --kernel-trace -d $O/pmc_$cnt -- python $R/bench.py --steps 2 --warmup 1 --chunks 1 --no-cpu-baseline --no-decode --time-all-kernels > $O/pmc_$cnt.log 2>&1 < /dev/null
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-decode > $O/trace.log 2>&1 < /dev/null
cd $R
timeout 200 python tools/phase_marks.py 2>&1 < /dev/null | tail -26 > $O/phase_marks.txt
timeout 300 python tools/bench_gemm.py --iters 30 > $O/gemm_roofline.txt 2>&1 < /dev/null
timeout 200 python tools/gemm_paths.py > $O/gemm_paths.txt 2>&1 < /dev/null
timeout 200 python tools/bench_infer.py --steps 200 > $O/infer.json 2> $O/infer.err < /dev/null
timeout 200 python tools/bench_infer.py --steps 200 --batch 8 > $O/infer_b8.json 2>> $O/infer.err < /dev/null
cd /tmp
rm -rf $O/dectrace $O/megatrace
SATT_DECODE_MEGA=0 timeout 200 rocprofv3 --kernel-trace --output-format csv -d $O/dectrace -- python $R/tools/bench_infer.py --steps 64 > $O/dectrace.log 2>&1 < /dev/null
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/megatrace -- python $R/tools/bench_infer.py --steps 192 > $O/megatrace.log 2>&1 < /dev/null
cd $R
(echo "# rocprofv3 --kernel-trace of SATT_DECODE_MEGA=0 tools/bench_infer.py --steps 64 (B=1, Ti=100, bf16: the LAUNCH-PER-LAYER path, hipGraph of 8 steps per replay), tools/decode_timeline.py:"; echo "# two consecutive decoder steps; every launch starts when its predecessor ends (gap 0): the step is a chain of dependent"; echo "# launches, each >= 4.7 us start to start however little it does (round 2 start: 11 launches, 75 us)."; timeout 60 python tools/decode_timeline.py $O/dectrace < /dev/null) > $O/decode_timeline.txt 2>&1
timeout 300 python bench.py --model tacotron --no-decode > $O/bench_tacotron.json 2> $O/bench_tacotron.err < /dev/null
timeout 300 python bench.py --model vctk --no-decode --no-cpu-baseline > $O/bench_vctk.json 2> $O/bench_vctk.err < /dev/null
timeout 200 python tools/phase_marks.py --dist 2>&1 < /dev/null | grep " ms$\|deferred" > $O/phase_marks_rccl.txt
timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-decode --force-dist 2>/dev/null < /dev/null | grep "^{" > $O/bench_rccl_one_rank.json
timeout 100 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 < /dev/null
timeout 900 python -m pytest tests -m gpu -q 2>&1 < /dev/null | tail -4 > $O/gpu_tests.log
# round 3: dynamic instruction counts (issue-floor table), per-phase anatomy of the attention loops (profile / trace builds made
# beforehand with tools/build_variant.sh: prof, tracefwd, tracebwd), the tests at the exact bench workloads with their prints
# r5: the probe variants are built HERE (tools/probes/*.so no longer travels with the snapshot)
mkdir -p $R/tools/probes
if [ -z "$SATT_MEASURE_SKIP_ATTN" ]; then   # (the attention-loop anatomy needs the variant builds; skipped when the kernels did not change)
bash tools/build_variant.sh prof "attn_cluster.hip" -DSATT_PROFILE > /dev/null 2>&1
bash tools/build_variant.sh tracefwd "attn_cluster.hip" -DSATT_PROFILE -DSATT_TRACE_ONLY > /dev/null 2>&1
bash tools/build_variant.sh tracebwd "attn_cluster.hip" -DSATT_PROFILE -DSATT_TRACE_ONLY -DSATT_TRACE_BWD > /dev/null 2>&1
bash tools/pmc_insts.sh > $O/pmc_insts.log 2>&1 < /dev/null
cp $R/gpurun_out/insts/insts.txt $O/insts.txt
(SATT_PROF_LIB=tools/probes/libsatt_prof.so SATT_LIB_PATH=tools/probes/libsatt_prof.so timeout 200 python tools/prof_attn.py 2>&1 | grep -v amdgpu.ids | tail -26;
 SATT_TRACE=1 SATT_PROF_LIB=tools/probes/libsatt_tracefwd.so SATT_LIB_PATH=tools/probes/libsatt_tracefwd.so timeout 200 python tools/prof_attn.py 2>&1 | tail -12;
 SATT_TRACE_BWD=1 SATT_PROF_LIB=tools/probes/libsatt_tracebwd.so SATT_LIB_PATH=tools/probes/libsatt_tracebwd.so timeout 200 python tools/prof_attn.py 2>&1 | tail -17) > $O/attn_loop_phases.txt 2>&1 < /dev/null
fi
timeout 600 python -m pytest tests/test_pinned_gpu.py tests/test_model_gpu.py -m gpu -q -s -k "bench_workload or vctk_workload or unrounded or folded_context or golden" 2>&1 < /dev/null | grep "bf16 vs f32 mode\|full size\|fold vs\|^small\|^medium\|passed\|failed" | cut -c1-330 > $O/parity_bench_workloads.log
# r5: encoder LSTM / small-attention kernels stand-alone, the persistent decode step's phase anatomy, the residency sweep, host enqueue
timeout 200 python tools/lstm_time.py 2>/dev/null < /dev/null > $O/encoder_lstm.txt
timeout 100 python tools/small_attn_time.py 2>/dev/null < /dev/null > $O/small_attn.txt
timeout 100 python tools/flash_time.py 2>/dev/null < /dev/null > $O/flash.txt
bash tools/build_variant.sh megaprof "decode_mega2.hip" -DSATT_MEGA_PROF > /dev/null 2>&1
(SATT_LIB_PATH=tools/probes/libsatt_megaprof.so timeout 100 python tools/decode_mega_prof.py 1 2>&1 | grep -v amdgpu.ids | tail -36) > $O/decode_phases.txt < /dev/null
timeout 200 python tools/bench_infer.py --steps 200 --batch 2 > $O/infer_b2.json 2>> $O/infer.err < /dev/null
SATT_DECODE_MEGA=0 timeout 200 python tools/bench_infer.py --steps 200 > $O/infer_graph_path.json 2>> $O/infer.err < /dev/null
timeout 600 python -m pytest tests/test_pinned_gpu.py -m gpu -q -s -k "sized_from" 2>&1 < /dev/null | grep "B=\|passed\|failed" | cut -c1-200 > $O/residency_sweep.txt
timeout 300 python -m pytest tests/test_pinned_gpu.py -m gpu -q -s -k frozen_float64 2>&1 < /dev/null | grep -v amdgpu.ids | cut -c1-400 > $O/parity_frozen_oracle.log
timeout 300 python -m pytest tests/test_decode_golden_gpu.py -m gpu -q -s 2>&1 < /dev/null | grep "decode b\|stop rule\|mel abs\|passed\|failed" | cut -c1-330 > $O/decode_golden.log
(for i in 1 2 3; do timeout 100 python tools/host_enqueue_time.py 2>/dev/null | tail -1; done; SATT_BTT=32,80,500 timeout 100 python tools/host_enqueue_time.py 2>/dev/null | tail -1 | sed 's/^/VCTK shape (B=32, Ti=80, Tm=500): /') > $O/host_enqueue.txt < /dev/null
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err < /dev/null
ls -la $O
# r6: the LDS-poison sweep (every kernel launch preceded by a launch that leaves the pattern in every LDS word of every CU) over the
# parity tests, NaN against 1.0; the cold-start trials of the persistent decode kernel are part of the GPU suite (tests/test_decode_cold_gpu.py)
(for pat in 7fc00000 3f800000; do echo "SATT_DEBUG_POISON_LDS=$pat:"; SATT_DEBUG_POISON_LDS=$pat timeout 900 python -m pytest tests/test_decode_golden_gpu.py tests/test_inference_gpu.py tests/test_ops_gpu.py tests/test_flash_gpu.py tests/test_gemm_tile_gpu.py tests/test_model_gpu.py tests/test_modules_gpu.py -m gpu -q 2>&1 < /dev/null | tail -1; SATT_DEBUG_POISON_LDS=$pat timeout 600 python -m pytest tests/test_pinned_gpu.py -m gpu -q -k "frozen_float64 or golden_fixtures" 2>&1 < /dev/null | tail -1; done) > $O/lds_poison_sweep.txt
