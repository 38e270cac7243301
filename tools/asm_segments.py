"""Static instruction mix between workgroup barriers of one kernel in a hipcc -S listing (quick look at where the
per-step instruction issue goes: VALU / scalar / LDS / memory, and scalar-spill traffic via v_readlane/v_writelane)."""
import sys

def main(path, needle):
    lines = open(path).read().split('\n')
    start = [i for i, l in enumerate(lines) if l.startswith('_Z') and needle in l and ': ' in l][0]
    end = [i for i, l in enumerate(lines[start:]) if 's_endpgm' in l][-1] + start
    end = min(end, [i for i, l in enumerate(lines[start:]) if '.end_amdhsa_kernel' in l][0] + start)
    seg, cur = [], dict(n=0, valu=0, vmem=0, lds=0, salu=0, lane=0, mfma=0)
    for i in range(start, end):
        l = lines[i].strip()
        if not l or l[0] in ';.' or l.endswith(':'):
            continue
        op = l.split()[0]
        cur['n'] += 1
        if op.startswith(('v_readlane', 'v_writelane')): cur['lane'] += 1
        if op.startswith('v_mfma'): cur['mfma'] += 1
        elif op.startswith('v_'): cur['valu'] += 1
        elif op.startswith(('global_', 'buffer_', 'flat_')): cur['vmem'] += 1
        elif op.startswith('ds_'): cur['lds'] += 1
        elif op.startswith('s_'): cur['salu'] += 1
        if op == 's_barrier':
            seg.append(cur); cur = dict(n=0, valu=0, vmem=0, lds=0, salu=0, lane=0, mfma=0)
    seg.append(cur)
    for k, s in enumerate(seg):
        print(k, s)

if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
