#!/usr/bin/env python3
"""Reaching-definitions analysis of the SGPR spill lanes of ONE kernel in a gfx950 code object (no GPU needed).

The compiler spills SGPRs into lanes of reserved VGPRs (v_writelane / v_readlane).  This tool builds the kernel's control-flow
graph from ``llvm-objdump -d`` output and reports, per spill reload, which spill stores can reach it: a reload reachable from
the kernel entry WITHOUT any store reads an undefined lane; a reload reached by several stores is normal for a scalar that is
updated on some paths only (one slot per virtual register), and is listed for inspection.

Written for the round-5 two-launch defect (one wave-uniform load base of pk_xall restored wrongly in
builds that spill SGPRs to VGPR lanes).  Result there: 253 stores into 85 slots of v254 / v255, 431 reloads, none reachable
without a store, 156 reached by more than one -- all of the conditionally-updated kind as far as inspected: NOT conclusive.

usage:  clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=model.hsaco --output=m.elf
        llvm-objdump --disassemble-symbols=pk_xall m.elf > k.s
        tools/spill_lane_flow.py k.s pk_xall [v254,v255]
(model.hsaco = zlib-inflated pockit_amd/_cache/<key>.hsacoz)"""
import re, sys, collections
fn = sys.argv[1]; base_name = sys.argv[2]
ins=[]   # (addr, op, args, target or None)
base=None
for l in open(fn):
    m=re.match(r"([0-9a-f]+) <%s>:" % base_name, l)
    if m: base=int(m.group(1),16); continue
    m=re.match(r"\s+(\S+)\s+(.*?)\s*//\s*([0-9A-Fa-f]+):\s*([0-9A-Fa-f ]+?)(?:\s+<%s\+0x([0-9a-fA-F]+)>)?\s*$" % base_name, l)
    if m:
        tgt = base+int(m.group(5),16) if m.group(5) else None
        ins.append((int(m.group(3),16), m.group(1), m.group(2), tgt))
addr2idx={a:i for i,(a,_,_,_) in enumerate(ins)}
n=len(ins)
spillregs={"v254","v255"} if len(sys.argv)<4 else set(sys.argv[3].split(","))
# leaders
leaders={0}
for i,(a,op,args,t) in enumerate(ins):
    if op.startswith(("s_cbranch","s_branch")):
        if t in addr2idx: leaders.add(addr2idx[t])
        if i+1<n: leaders.add(i+1)
    if op.startswith(("s_endpgm","s_setpc")) and i+1<n: leaders.add(i+1)
L=sorted(leaders); blk_of={}
blocks=[]
for k,s in enumerate(L):
    e=(L[k+1] if k+1<len(L) else n)
    blocks.append((s,e))
    for i in range(s,e): blk_of[i]=k
succ=collections.defaultdict(list); unknown=[]
for k,(s,e) in enumerate(blocks):
    a,op,args,t=ins[e-1]
    if op.startswith("s_branch"):
        if t in addr2idx: succ[k].append(blk_of[addr2idx[t]])
        else: unknown.append(hex(a))
    elif op.startswith("s_cbranch"):
        if t in addr2idx: succ[k].append(blk_of[addr2idx[t]])
        else: unknown.append(hex(a))
        if e<n: succ[k].append(blk_of[e])
    elif op.startswith("s_endpgm"): pass
    elif op.startswith("s_setpc"): unknown.append(hex(a))
    else:
        if e<n: succ[k].append(blk_of[e])
pred=collections.defaultdict(list)
for k,v in succ.items():
    for t in v: pred[t].append(k)
print("instructions", n, "blocks", len(blocks), "unresolved jumps", unknown)
# definitions
defs=[]   # (idx, reg, lane, sreg)
for i,(a,op,args,t) in enumerate(ins):
    if op.startswith("v_writelane"):
        d,s,lane=[x.strip() for x in args.split(",")]
        if d in spillregs: defs.append((i,d,lane,s))
slots=sorted({(d,lane) for _,d,lane,_ in defs})
print("spill slots", len(slots), "stores", len(defs))
UNDEF=-1
# per block gen/kill on slots: last def in block per slot
gen=[{} for _ in blocks]
for di,(i,d,lane,s) in enumerate(defs):
    gen[blk_of[i]][(d,lane)]=di          # later overwrites earlier: last def in block
IN=[{sl:set() for sl in slots} for _ in blocks]
OUT=[{sl:set() for sl in slots} for _ in blocks]
for sl in slots: IN[0][sl]={UNDEF}
changed=True
while changed:
    changed=False
    for k in range(len(blocks)):
        for sl in slots:
            if k==0: new_in={UNDEF}|set().union(*[OUT[p][sl] for p in pred[k]]) if pred[k] else {UNDEF}
            else: new_in=set().union(*[OUT[p][sl] for p in pred[k]]) if pred[k] else set()
            if new_in!=IN[k][sl]: IN[k][sl]=new_in; changed=True
            new_out={gen[k][sl]} if sl in gen[k] else new_in
            if new_out!=OUT[k][sl]: OUT[k][sl]=new_out; changed=True
# walk reads
multi=[]; undef=[]; reads=0
for k,(s,e) in enumerate(blocks):
    cur={sl:set(IN[k][sl]) for sl in slots}
    for i in range(s,e):
        a,op,args,t=ins[i]
        if op.startswith("v_writelane"):
            d,sr,lane=[x.strip() for x in args.split(",")]
            if (d,lane) in cur: cur[(d,lane)]={next(di for di,(ii,_,_,_) in enumerate(defs) if ii==i)}
        elif op.startswith("v_readlane"):
            sr,d,lane=[x.strip() for x in args.split(",")]
            if (d,lane) in cur:
                reads+=1
                r=cur[(d,lane)]
                if UNDEF in r: undef.append((hex(a),args,sorted(x for x in r if x!=UNDEF)))
                elif len(r)>1: multi.append((hex(a),args,[(hex(ins[defs[x][0]][0]),defs[x][3]) for x in sorted(r)]))
print("spill reloads", reads, "| reloads reached by >1 store:", len(multi), "| reloads reachable without any store:", len(undef))
for m in multi[:40]: print("  MULTI", m)
for u in undef[:40]: print("  UNDEF", u)
