#!/usr/bin/env python3
"""Diagnostic (round 6): how far ahead of their first use does a kernel's ISA issue its LDS reads?
    hipcc -O3 --offload-arch=gfx950 -Iinclude -Iartensor_amd/csrc -DARTN_TU_BITS=6 -DARTN_TU_HALF=1 --offload-device-only -S \
          artensor_amd/csrc/artn_kernels.hip -o /tmp/k6h1.s && python3 tools/isa_prefetch_depth.py /tmp/k6h1.s
Per artn_k_bits instantiation: MFMA count, and a histogram of "MFMAs issued between a ds_read and the first instruction that
reads its result" (0 = ds_read, s_waitcnt lgkmcnt(0), use: an exposed LDS round trip; 9 = nine or more).  About 60 zero-distance
reads are table look-ups outside the stages; a stage whose operand reads are sunk next to their uses shows as 30+ more
(the 6-bit 3M stage until round 6: profiles/r06_tile_loop.md)."""
import re,sys,collections
def kernels(path):
    s=open(path).read()
    for m in re.finditer(r'\n(_Z\w+):\s*; @', s):
        name=m.group(1); i=m.end(); j=s.index('.Lfunc_end',i)
        yield name, s[i:j].split('\n')
def regs(tok):
    m=re.match(r'v\[(\d+):(\d+)\]',tok)
    if m: return set(range(int(m.group(1)),int(m.group(2))+1))
    m=re.match(r'v(\d+)$',tok)
    if m: return {int(m.group(1))}
    return set()
def analyze(lines):
    # linear scan (ignores control flow): for each ds_read dest, MFMAs issued until first read of the dest by an mfma/valu
    pend=[]  # (dest regs, mfma count at issue)
    nm=0; dist=[]
    spill=sum(1 for l in lines if 'v_readlane' in l or 'v_writelane' in l)
    for l in lines:
        t=l.strip()
        if not t or t.startswith(';') or t.startswith('.'): continue
        ops=re.split(r'[ ,]+',t)
        op=ops[0]
        if op.startswith('ds_read'):
            pend.append((regs(ops[1]),nm)); continue
        srcs=set()
        for o in ops[2:]: srcs|=regs(o)
        if op.startswith('v_mfma'):
            for p in list(pend):
                if p[0]&srcs:
                    dist.append(nm-p[1]); pend.remove(p)
            nm+=1
        elif op.startswith('v_'):
            for p in list(pend):
                if p[0]&srcs:
                    dist.append(nm-p[1]); pend.remove(p)
    return nm, dist, spill
for path in sys.argv[1:]:
    for name,lines in kernels(path):
        if 'artn_k_bits' not in name: continue
        nm,dist,spill=analyze(lines)
        if nm<20: continue
        c=collections.Counter(min(d,9) for d in dist)
        m=re.search(r'artn_k_bitsILi(\d)ELi(\d)ELb(\d)ELi(\d)ELb(\d)ELb(\d)ELb(\d)ELb(\d)ELi(\d)',name)
        tag='k%s+%s big%s np%s g%s nt%s m3%s full%s n3%s'%m.groups() if m else name[:40]
        vg=None
        print(f"{tag:44s} mfma {nm:4d} reads {len(dist):4d} lanespill {spill:4d}  MFMAs between read and use: "+' '.join(f"{k}:{c[k]}" for k in sorted(c)))
