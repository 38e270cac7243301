"""Static instruction mix per phase of one kernel in a `hipcc -S -DSATT_ASM_MARKS` listing (quick look at where the
per-step instruction issue goes: VALU / scalar / LDS / memory / MFMA, and scalar-spill traffic via v_readlane /
v_writelane).  Segments are delimited by the `; SATT_MARK n` comments the PROF() macro leaves in that build.
usage: python tools/asm_segments.py listing.s <substring of the mangled kernel name> [--hist MARK]"""
import collections
import sys


def main(path, needle, hist_mark=None):
    lines = open(path).read().split('\n')
    start = [i for i, l in enumerate(lines) if l.startswith('_Z') and needle in l and ': ' in l][0]
    segs, cur, name = [], collections.Counter(), 'prologue'
    hist = collections.Counter()
    for i in range(start, len(lines)):
        l = lines[i].strip()
        if '.end_amdhsa_kernel' in l:
            break
        if 'SATT_MARK' in l:
            segs.append((name, cur)); cur = collections.Counter(); name = 'after mark ' + l.split('SATT_MARK')[1].strip()
            continue
        if not l or l[0] in ';.' or l.endswith(':'):
            continue
        op = l.split()[0]
        cur['n'] += 1
        if hist_mark is not None and name == 'after mark ' + hist_mark:
            hist[op] += 1
        if op.startswith(('v_readlane', 'v_writelane')): cur['lane'] += 1
        if op.startswith('v_mfma'): cur['mfma'] += 1
        elif op.startswith('v_'): cur['valu'] += 1
        elif op.startswith(('global_', 'buffer_', 'flat_')): cur['vmem'] += 1
        elif op.startswith('ds_'): cur['lds'] += 1
        elif op.startswith('s_barrier'): cur['barrier'] += 1
        elif op.startswith('s_'): cur['salu'] += 1
    segs.append((name, cur))
    for nm, s in segs:
        print('%-16s' % nm, ' '.join('%s=%d' % (k, s[k]) for k in ('n', 'valu', 'lane', 'salu', 'lds', 'vmem', 'mfma', 'barrier')))
    for k, v in hist.most_common(40):
        print(v, k)


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], sys.argv[sys.argv.index('--hist') + 1] if '--hist' in sys.argv else None)
