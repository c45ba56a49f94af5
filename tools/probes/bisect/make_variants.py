"""dev tool: ISA-level bisection of the lane-quarter hazard.  Reads the device assembly of geometry_lane_probe.hip, writes variants
in which `s_nop` wait states are inserted after / before instruction classes inside geom_kernel<6> only."""
import re, sys
src = open(sys.argv[1]).read().split('\n')
out_dir = sys.argv[2]
name = '_Z11geom_kernelILi6EEvPKfS1_iiPyS2_Pi'
start = next(i for i, l in enumerate(src) if l.startswith(name + ':'))
end = next(i for i in range(start, len(src)) if 's_endpgm' in src[i])
body = list(range(start + 1, end))

def is_instr(l):
    t = l.strip()
    return bool(t) and not t.startswith(('.', ';', '//')) and not t.endswith(':') and l.startswith('\t')

def variant(tag, after=None, before=None, rng=None, nop='s_nop 1'):
    res = []
    k = 0
    guard = 0          # never split s_getpc_b64 from the two relocated adds that follow it
    for i, l in enumerate(src):
        ins = i in body_set and is_instr(l)
        op = l.strip().split()[0] if ins else ''
        inside = ins and (rng is None or rng[0] <= k < rng[1])
        if ins:
            k += 1
        if inside and guard == 0 and before and re.match(before, op):
            res.append('\t' + nop)
        res.append(l)
        if op == 's_getpc_b64':
            guard = 2
            continue
        if guard:
            guard -= 1
            continue
        if inside and after and re.match(after, op) and not op.startswith(('s_endpgm', 's_branch', 's_cbranch', 's_setpc')):
            res.append('\t' + nop)
    open('%s/%s.s' % (out_dir, tag), 'w').write('\n'.join(res))

body_set = set(body)
n_instr = sum(1 for i in body if is_instr(src[i]))
print('kernel body: %d instructions' % n_instr)
mode = sys.argv[3] if len(sys.argv) > 3 else 'classes'
if mode == 'classes':
    variant('Z_baseline')
    variant('A_after_vcmp', after=r'v_cmp')
    variant('B_after_readfirstlane', after=r'v_readfirstlane')
    variant('C_before_salu_mask_ops', before=r's_and_b64|s_and_saveexec_b64|s_cselect|s_or_b64|s_andn2')
    variant('D_after_vpk', after=r'v_pk_')
    variant('E_after_ds', after=r'ds_')
    variant('F_after_sload_waitcnt', after=r's_load|s_waitcnt')
    variant('G_after_every_instruction', after=r'.*')
    variant('H_after_every_valu', after=r'v_')
    variant('I_after_every_salu', after=r's_')
else:                     # ranges: after every instruction, but only inside [lo, hi) of the body
    lo, hi = int(sys.argv[4]), int(sys.argv[5])
    parts = int(sys.argv[6]) if len(sys.argv) > 6 else 4
    step = (hi - lo + parts - 1) // parts
    for p in range(parts):
        a, b = lo + p * step, min(hi, lo + (p + 1) * step)
        variant('R_%04d_%04d' % (a, b), after=r'.*', rng=(a, b))
