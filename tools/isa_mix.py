#!/usr/bin/env python3
"""Static instruction mix of one kernel in hipcc's assembly output, by instruction class and basic block:
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -DLG_LOGK=7 -S --cuda-device-only -o ntt7.s ligero_amd/csrc/ntt_inst.hip
    python tools/isa_mix.py ntt7.s _ZN2lg15ntt_rows_kernelILi7ELi0ELb1EEEvNS_7NttArgsE blocks
(profiles/r02_*_isa_mix.md were made with it)"""
import re, sys, collections
src=open(sys.argv[1]).read().splitlines()
kname=sys.argv[2]
# find function body: label "kname:" until ".Lfunc_end"
start=None
for i,l in enumerate(src):
    if l.startswith(kname+':'):
        start=i; break
end=None
for i in range(start,len(src)):
    if src[i].startswith('.Lfunc_end'):
        end=i; break
body=src[start:end]
# split into basic blocks; detect loops? report static counts per class, plus per-block counts
classes=collections.OrderedDict([
 ('v_mad_u64_u32','mad'),('v_mul_lo_u32','mul_lo'),('v_mul_hi_u32','mul_hi'),
])
def cls(op):
    if op.startswith('v_mad_u64_u32') or op.startswith('v_mad_i64_i32'): return 'v_mad_u64_u32'
    if op.startswith('v_mul_lo_u32'): return 'v_mul_lo_u32'
    if op.startswith('v_mul_hi_u32'): return 'v_mul_hi_u32'
    if op.startswith('v_lshrrev_b64') or op.startswith('v_lshlrev_b64') or op.startswith('v_ashrrev_i64'): return 'v 64-bit shift'
    if op.startswith('v_lshl_add_u64'): return 'v_lshl_add_u64'
    if op.startswith('v_alignbit'): return 'v_alignbit_b32'
    if op.startswith('v_and_or') or op.startswith('v_lshl_or') or op.startswith('v_lshl_add') or op.startswith('v_add_lshl') or op.startswith('v_add3') or op.startswith('v_bfe') or op.startswith('v_bfi') or op.startswith('v_and_or') or op.startswith('v_xad') or op.startswith('v_or3') or op.startswith('v_sub') and 'co' in op and False: return 'v 3-src int (add3/lshl_add/bfe/...)'
    if op.startswith('v_add_co') or op.startswith('v_addc') or op.startswith('v_sub_co') or op.startswith('v_subb') or op.startswith('v_subrev_co') or op.startswith('v_subbrev'): return 'v carry add/sub'
    if op.startswith('v_add_u32') or op.startswith('v_sub_u32') or op.startswith('v_subrev_u32') or op.startswith('v_add_nc'): return 'v_add/sub_u32'
    if op.startswith('v_and_b32') or op.startswith('v_or_b32') or op.startswith('v_xor_b32') or op.startswith('v_not'): return 'v and/or/xor'
    if op.startswith('v_lshrrev_b32') or op.startswith('v_lshlrev_b32') or op.startswith('v_ashrrev_i32'): return 'v 32-bit shift'
    if op.startswith('v_mov') or op.startswith('v_accvgpr'): return 'v_mov'
    if op.startswith('v_cndmask') or op.startswith('v_cmp'): return 'v cmp/cndmask'
    if op.startswith('v_readfirstlane') or op.startswith('v_readlane'): return 'v readlane'
    if op.startswith('v_'): return 'v other ('+op+')'
    if op.startswith('ds_'): return 'ds_* ('+op.split()[0]+')'
    if op.startswith('global_load') or op.startswith('buffer_load') or op.startswith('flat_load'): return 'global load'
    if op.startswith('global_store') or op.startswith('buffer_store') or op.startswith('flat_store'): return 'global store'
    if op.startswith('s_waitcnt'): return 's_waitcnt'
    if op.startswith('s_barrier'): return 's_barrier'
    if op.startswith('s_load') or op.startswith('s_buffer_load'): return 's_load'
    if op.startswith('s_nop'): return 's_nop'
    if op.startswith('s_cbranch') or op.startswith('s_branch'): return 's branch'
    if op.startswith('s_'): return 's other'
    return 'other'
blocks=[]; cur=('entry',collections.Counter())
for l in body[1:]:
    t=l.strip()
    if not t or t.startswith(';') or t.startswith('.') and not t.endswith(':'): continue
    if re.match(r'^[.\w$]+:', t):
        blocks.append(cur); cur=(t.rstrip(':'),collections.Counter()); continue
    op=t.split()[0]
    cur[1][cls(op)]+=1
blocks.append(cur)
tot=collections.Counter()
for n,c in blocks: tot.update(c)
print('kernel',kname)
print('static instruction mix, whole kernel (%d instructions, %d basic blocks):'%(sum(tot.values()),len(blocks)))
for k,v in tot.most_common():
    print('  %6d  %5.1f%%  %s'%(v,100.0*v/sum(tot.values()),k))
if len(sys.argv)>3:
    print('largest basic blocks:')
    for n,c in sorted(blocks,key=lambda b:-sum(b[1].values()))[:8]:
        print('  block',n,sum(c.values()),'instr:',', '.join('%s=%d'%(k,v) for k,v in c.most_common(8)))
