#!/usr/bin/env python3
"""Static instruction counts per marked region of mega_bvh_kernel<768,false>.

    hipcc ... -DMORT_REGION_MARKS --cuda-device-only -S -o k.s mort_amd/csrc/hip/mort_hip.hip
    scripts/isa_regions.py k.s

Regions are the asm comments REGION(...) leaves; block placement is the compiler's, so a region's count is what lies
between its mark and the next one in the listing (approximate when blocks are reordered).
"""
import re, sys
txt = open(sys.argv[1]).read().split('\n')
name = sys.argv[2] if len(sys.argv) > 2 else '_Z15mega_bvh_kernelILi768ELb0EEv8FastArgs'
a = next(i for i, l in enumerate(txt) if l.startswith(name + ':'))
b = next(i for i in range(a, len(txt)) if txt[i].startswith('.Lfunc_end'))
lines = txt[a:b]
marks = [(0, 'start')] + [(i, l.split('REGION')[1].strip()) for i, l in enumerate(lines) if '; REGION' in l] + [(len(lines), 'end')]
tot = 0
for (x, n), (y, _) in zip(marks[:-1], marks[1:]):
    seg = lines[x:y]
    c = lambda pat: sum(1 for l in seg if re.match(pat, l))
    v = c(r'\s+v_'); tot += v
    cols = [('valu', v), ('f64', c(r'\s+v_.*_f64')), ('mov', c(r'\s+v_mov_b')), ('scratch', c(r'\s+scratch_')),
            ('div32', c(r'\s+v_div_fixup_f32')), ('div64', c(r'\s+v_div_fixup_f64')), ('sqrt32', c(r'\s+v_sqrt_f32')),
            ('lds', c(r'\s+ds_')), ('salu', c(r'\s+s_'))]
    print('%5d %-18s ' % (x, n) + ' '.join('%s %4d' % kv for kv in cols))
print('total valu', tot)
