"""Hot loops of a disassembled kernel (build/isa/k.s, written by tools/dump_isa.sh): innermost loops with >= 100 fp64
instructions, their instruction mix, and where the kernel's scratch (spill) instructions sit."""
import re, sys, collections
lines = open(sys.argv[1] if len(sys.argv) > 1 else 'build/isa/k.s').read().split('\n')
ins = []
for i, l in enumerate(lines):
    m = re.search(r'//\s*([0-9A-F]{12}):', l)
    if m: ins.append((int(m.group(1), 16), l.strip().split('//')[0].strip(), i))
a2i = {a: k for k, (a, _, _) in enumerate(ins)}
loops = []
for k, (a, t, i) in enumerate(ins):
    m = re.match(r's_cbranch_\w+\s+(\d+)', t) or re.match(r's_branch\s+(\d+)', t)
    if m:
        off = int(m.group(1))
        if off >= 32768:
            tgt = a + 4 + (off - 65536) * 4
            if tgt in a2i: loops.append((a2i[tgt], k))
isf = lambda b: re.match(r'v_(fma|mul|add|fmac|rcp).*f64', b[1])
for (s, e) in loops:
    body = ins[s:e + 1]
    if sum(1 for b in body if isf(b)) < 100: continue
    if any(s <= s2 and e2 <= e and (s2, e2) != (s, e) and sum(1 for b in ins[s2:e2 + 1] if isf(b)) >= 100 for (s2, e2) in loops): continue
    print("lines %d-%d n=%d valu=%d f64=%d ds=%d vmem=%d scratch=%d" % (ins[s][2] + 1, ins[e][2] + 1, len(body), sum(1 for b in body if b[1].startswith('v_')),
          sum(1 for b in body if isf(b)), sum(1 for b in body if b[1].startswith('ds_')), sum(1 for b in body if b[1].startswith(('global_', 'buffer_'))),
          sum(1 for b in body if b[1].startswith('scratch_'))))
sl = [b[2] + 1 for b in ins if b[1].startswith('scratch_')]
print(len(sl), "scratch instructions; per 500 lines:", sorted(collections.Counter(x // 500 * 500 for x in sl).items()))
