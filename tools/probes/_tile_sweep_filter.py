"""stdin: the table of tools/bench_gemm.py; stdout: the `tile us` column of its forward / input-gradient rows on one line -
for tile-shape sweeps: `for c in "" SATT_TILE_BM=128 SATT_TILE_BN=64 ...; do env $c python tools/bench_gemm.py --no-dw --no-generic | python tools/probes/_tile_sweep_filter.py; done`"""
import re,sys
out=[]
for ln in sys.stdin:
    m=re.match(r"^(.*?)\s+(\d+\.\d+)\s+(\d+\.\d+)\s+(\d+\.\d+)\s+0\.\d+", ln)
    if m and ("fwd" in ln or "dX" in ln): out.append(m.group(3))
print(" ".join(out))
