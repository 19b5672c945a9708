"""A census made on the FIRST reading of round 5's gfx950 packed-fp32 hazard (LAB_NOTES.md: "a late result"; the hand-assembled s_nop canaries later showed the result
is wrong, not late): per kernel of a disassembled translation unit, the sites where the result of a v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32 is read by a plain
(non-packed) VALU instruction one or two instructions later.
usage: bash tools/kernel_isa.sh mixermdm_amd/csrc/attn_f32.o . > /tmp/attn.s; python tools/pk_hazard_scan.py /tmp/attn.s [...]"""
import re,sys,collections
def regs(tok):
    out=set()
    for m in re.finditer(r'v\[(\d+):(\d+)\]|v(\d+)', tok):
        if m.group(1): out.update(range(int(m.group(1)), int(m.group(2))+1))
        else: out.add(int(m.group(3)))
    return out
for f in sys.argv[1:]:
    kern=None; lines=[]; res=collections.Counter(); tot=collections.Counter()
    for l in open(f):
        m=re.match(r'^[0-9a-f]+ <(.*)>:$', l.strip())
        if m: kern=m.group(1); lines=[]; continue
        t=l.strip().split('//')[0].strip()
        if not t: continue
        lines.append(t)
        if len(lines)>3: lines.pop(0)
        # check previous instructions for pk producers
        op=t.split()[0]
        if not op.startswith('v_') or op.startswith('v_pk_'): 
            continue
        ops=t[len(op):].split(',')
        srcs=set()
        for o in ops[1:]: srcs|=regs(o)
        for back in (1,2):
            if len(lines)>back:
                p=lines[-1-back]
                pop=p.split()[0]
                if re.match(r'v_pk_(mul|fma|add)_f32', pop):
                    dst=regs(p[len(pop):].split(',')[0])
                    if dst & srcs:
                        res[(kern,back)]+=1
    import subprocess
    agg=collections.Counter()
    for (k,b),n in res.items(): agg[k]+=n
    print(f, 'kernels with a packed-fp32 result read by a plain VALU op 1-2 slots later:', len(agg), 'sites:', sum(agg.values()))
    for k,n in agg.most_common(8):
        d=subprocess.run(['c++filt',k],capture_output=True,text=True).stdout.strip()
        print('   ', n, d[:110])
