#!/usr/bin/env python3
"""Static instruction audit of a trace kernel's per-surface loop, from the assembly hipcc
emits for gfx950 (no GPU needed): which of the wave-level instructions one intersection
costs are arithmetic the reference's formulas ask for, and which are the band tests,
selects, address arithmetic and scalar bookkeeping around them.

    python tools/isa_audit.py [--inst lean] [--mode 2] > profiles/r03_isa_audit_hits.json

The tile loop is the largest backward-branch region of the kernel (label ... `s_cbranch*
label` further down), the surface loop the largest region nested inside it; that region is
straight-line code with predicated
(`s_and_saveexec`) sub-regions, so its static instruction count is an upper bound of what
a wave issues per surface -- rarely taken slow paths (the fix-up wrappers of sqrt and /
outside the slim band, the Newton tail) are inside it.  The PMC figure to hold it against
is SQ_INSTS_VALU / intersections (profiles/valu_per_intersection.json)."""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-fno-fast-math']

CLASSES = [
    ('fp64 fma', r'v_fma_f64|v_fmac_f64'),
    ('fp64 mul', r'v_mul_f64'),
    ('fp64 add', r'v_add_f64'),
    ('fp64 div / sqrt seeds and scaling (rcp, rsq, div_scale, div_fmas, div_fixup, ldexp, frexp)',
     r'v_rcp_f64|v_rsq_f64|v_sqrt_f64|v_div_scale_f64|v_div_fmas_f64|v_div_fixup_f64|v_ldexp_f64|v_frexp'),
    ('fp64 compare / class', r'v_cmp\w*_f64|v_cmpx\w*_f64|v_cmp_class_f64'),
    ('fp64 min / max / other', r'v_(min|max|trunc|floor|ceil|rndne|fract)\w*_f64'),
    ('integer compare (band tests on exponent words, flags)', r'v_cmp\w*_[iu](16|32|64)|v_cmpx\w*_[iu](16|32|64)'),
    ('select (v_cndmask)', r'v_cndmask'),
    ('integer / bit ops on VGPRs (and, or, xor, shifts, bfe, add: band tests, sign handling, addresses)',
     r'v_(and|or|xor|not|lshl|lshr|ashr|bfe|bfi|add|sub|mul|mad|perm|alignbit)\w*_(b|u|i|co)'),
    ('moves / readlane / conversions', r'v_mov|v_readlane|v_readfirstlane|v_writelane|v_cvt|v_accvgpr|v_swap'),
    ('LDS reads', r'ds_read|ds_load'),
    ('LDS writes', r'ds_write|ds_store'),
    ('global / buffer stores', r'global_store|buffer_store|flat_store'),
    ('global / buffer / scalar loads', r'global_load|buffer_load|flat_load|s_load|s_buffer_load'),
    ('scalar ALU', r's_(and|or|xor|not|andn2|orn2|nand|nor|add|sub|mul|lshl|lshr|ashr|bfe|cmp|cselect|mov|cmov|'
                   r'ff1|flbit|bcnt|min|max|abs|sext|brev|bitset|movk|addk|mulk|cmpk|pack|getpc|setpc|swappc)\w*'),
    ('exec-mask handling (saveexec, wrexec)', r's_\w*saveexec|s_\w*wrexec'),
    ('branches', r's_cbranch|s_branch'),
    ('waits / barriers / nops', r's_waitcnt|s_barrier|s_nop|s_sleep|s_setprio|s_delay'),
]


def classify(op):
    for name, pat in CLASSES:
        if re.match(pat, op):
            return name
    return 'other: ' + op


def kernel_body(asm, mangled):
    lines = asm.splitlines()
    start = next(i for i, ln in enumerate(lines) if ln.startswith(mangled + ':'))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('.end_amdhsa_kernel')
               or lines[i].strip().startswith('s_endpgm'))
    # (the kernel may have several s_endpgm; take up to the .Lfunc_end label)
    for i in range(start, len(lines)):
        if lines[i].startswith('.Lfunc_end') or lines[i].strip().startswith('.section'):
            end = i
            break
    return lines[start + 1:end]


def instructions(body):
    """[(label or None, opcode, text)] in order"""
    out = []
    for ln in body:
        s = ln.split(';')[0].rstrip()
        if not s.strip():
            continue
        m = re.match(r'^(\.?\w+):', s)
        if m:
            out.append((m.group(1), None, s))
            continue
        t = s.strip()
        if t.startswith('.'):
            continue
        out.append((None, t.split()[0], t))
    return out


def loops(ins):
    """every backward-branch region: (instructions, first, last, label), largest first"""
    labels = {lab: i for i, (lab, _op, _t) in enumerate(ins) if lab}
    out = {}
    for i, (_lab, op, text) in enumerate(ins):
        if op and (op.startswith('s_cbranch') or op == 's_branch'):
            tgt = text.split()[-1]
            j = labels.get(tgt)
            if j is not None and j < i:
                n = sum(1 for k in range(j, i + 1) if ins[k][1])
                if tgt not in out or n > out[tgt][0]:
                    out[tgt] = (n, j, i, tgt)
    return sorted(out.values(), reverse=True)


def surface_loop(ins):
    """the tile loop is the outermost region; the surface loop is the largest region nested
    inside it"""
    ls = loops(ins)
    outer = ls[0]
    inner = [l for l in ls[1:] if l[1] >= outer[1] and l[2] <= outer[2]]
    return (inner[0] if inner else outer), outer, ls


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--inst', default='lean', help='feature instance (csrc/inst_<name>.hip)')
    ap.add_argument('--mode', type=int, default=2, help='ROX_OUT_* of the kernel (2 = HITS, 0 = FULL)')
    ap.add_argument('--feat', type=int, default=0, help='FEAT template value of the instance (lean 0, even 1, radial 2)')
    ap.add_argument('--workload', default=None, help='PMC record to quote beside the static count')
    ap.add_argument('--fast', action='store_true',
                    help='the tolerance-mode twin (csrc/fast_<name>.hip, FEAT | 64)')
    args = ap.parse_args()
    if args.fast:
        args.feat |= 64
    src = os.path.join(ROOT, 'ray-optics_amd', 'csrc', f"{'fast' if args.fast else 'inst'}_{args.inst}.hip")
    with tempfile.TemporaryDirectory() as td:
        subprocess.check_call(['hipcc'] + FLAGS + ['-I', os.path.join(ROOT, 'include'), '-S', '--cuda-device-only',
                                                   src, '-o', os.path.join(td, 'k.s')])
        asm = open(os.path.join(td, 'k.s')).read()
    mangled = f'_ZN3rox12trace_kernelILi{args.mode}ELi1ELb0ELi{args.feat}ELb0EEEvNS_9TraceArgsE'
    ins = instructions(kernel_body(asm, mangled))
    total = sum(1 for _l, op, _t in ins if op)
    (n, j, i, tgt), outer, all_loops = surface_loop(ins)
    loop = [(op, t) for _l, op, t in ins[j:i + 1] if op]
    counts = collections.Counter(classify(op) for op, _t in loop)
    valu = sum(c for k, c in counts.items() if k.startswith(('fp64', 'integer', 'select', 'moves')))
    salu = sum(c for k, c in counts.items() if k.startswith(('scalar', 'exec', 'branches')))
    arith = sum(c for k, c in counts.items() if k.startswith(('fp64 fma', 'fp64 mul', 'fp64 add', 'fp64 div')))
    pmc = None
    p = os.path.join(ROOT, 'profiles', 'valu_per_intersection.json')
    if os.path.exists(p):
        j_ = json.load(open(p))
        wl = args.workload or {'lean': 'dblgauss_c2', 'radial': 'cell_phone', 'even': 'nikkor_c3'}.get(args.inst)
        if wl in j_:
            key = 'valu_wave_insts_per_intersection' if args.mode == 2 else 'full_valu_wave_insts_per_intersection'
            pmc = {'workload': wl, 'valu_wave_insts_per_lane_intersection': j_[wl][key],
                   'valu_wave_insts_per_wave_surface': j_[wl][key] * 64,
                   'note': 'SQ_INSTS_VALU / intersections counts per ray; one wave instruction serves 64 rays, so '
                           'the per-wave-per-surface figure is 64 x (lanes that left the trace earlier lower it)'}
    rec = {'kernel': mangled, 'instance': args.inst, 'out_mode': args.mode,
           'static_instructions_in_kernel': total,
           'backward_branch_regions': [{'label': l[3], 'static_instructions': l[0]} for l in all_loops[:8]],
           'tile_loop_static_instructions': outer[0],
           'surface_loop': {'label': tgt, 'static_instructions': n, 'valu': valu, 'salu_and_control': salu,
                            'fp64_arithmetic_the_formulas_ask_for': arith,
                            'valu_overhead (compares, selects, integer band tests, moves)': valu - arith,
                            'by_class': dict(sorted(counts.items(), key=lambda kv: -kv[1]))},
           'pmc': pmc,
           'how_to_read': 'static count of the surface loop = an upper bound of the wave '
                          'instructions per surface (slow paths of sqrt and / outside the slim band and '
                          'predicated failure handling are inside it)'}
    print(json.dumps(rec, indent=1))


if __name__ == '__main__':
    sys.exit(main())
