#!/usr/bin/env python3
"""Wait-count census of the hot conv_mfma_kernel instantiations in a device assembly listing (round 6: how the forced
`s_waitcnt vmcnt(0)` drains of the LDS-DMA staging were found, profiles/README.md "Round 6" section 1).

    hipcc -O3 --offload-arch=gfx950 -std=c++17 -S --cuda-device-only [-DCFG_...] -Iinclude \
          glomeruli_segmentation_amd/csrc/espnet.hip -o /tmp/espnet.s
    python tools/asm_waits.py /tmp/espnet.s [name="<template-argument tail to match>" ...]

Prints, per kernel: lines, matrix instructions, `vmcnt(0)` waits, all waits, v_mov copies, VGPRs, accumulator offset, scratch; writes
each kernel's listing to <file>_<name>.s for reading."""
import re, subprocess, sys
f = sys.argv[1]
lines = open(f).read().split('\n')
pats = {"l2esp_fused": "4, 9, 287747>", "l2esp_last": "4, 9, 41999>", "l2down": "4, 9, 279553>", "l3esp_fused": "2, 13, 4989955>",
        "l3esp_last": "2, 13, 4727811>", "l3down": "2, 13, 4457473>", "l3c1s_bnl": "4, 6, 9503232>", "l2c1s": "8, 3, 1049088>"}
if len(sys.argv) > 2:
    pats = dict(a.split('=') for a in sys.argv[2:])
syms = [(i, l.split(':')[0]) for i, l in enumerate(lines) if re.match(r'^_ZN2gs\w+:', l)]
dem = subprocess.run(['c++filt'] + [s for _, s in syms], capture_output=True, text=True).stdout.split('\n')
for name, pat in pats.items():
    hit = [(i, s) for (i, s), d in zip(syms, dem) if pat in d]
    if not hit:
        print(name, 'not found'); continue
    i, s = hit[0]
    j = i
    while 's_endpgm' not in lines[j]: j += 1
    body = lines[i:j + 1]
    open(f[:-2] + '_' + name + '.s', 'w').write('\n'.join(body))
    txt = '\n'.join(lines[j:j + 120])
    m = re.search(r'next_free_vgpr (\d+)', txt); sp = re.search(r'private_segment_fixed_size (\d+)', txt)
    acc = re.search(r'accum_offset (\d+)', txt)
    c = lambda p: sum(1 for l in body if re.search(p, l))
    print("%-12s lines %5d mfma %4d vmcnt0 %2d waitcnt %3d v_mov %3d vgpr %s accoff %s scratch %s" % (name, len(body), c('v_mfma'), c(r'vmcnt\(0\)'), c('s_waitcnt'), c('v_mov_b32'), m and m.group(1), acc and acc.group(1), sp and sp.group(1)))
