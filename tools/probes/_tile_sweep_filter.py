import re,sys
out=[]
for ln in sys.stdin:
    m=re.match(r"^(.*?)\s+(\d+\.\d+)\s+(\d+\.\d+)\s+(\d+\.\d+)\s+0\.\d+", ln)
    if m and ("fwd" in ln or "dX" in ln): out.append(m.group(3))
print(" ".join(out))
