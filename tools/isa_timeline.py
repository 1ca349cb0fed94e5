"""Instruction-category timeline of one kernel in an assembly listing (hipcc -S):
   python tools/isa_timeline.py file.s <mangled-name-substring> [window]"""
import sys, collections
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
win = int(sys.argv[3]) if len(sys.argv) > 3 else 250
start = [i for i, l in enumerate(lines) if key in l and l.split(';')[0].rstrip().endswith(':')][0]
buf = []
for j in range(start, len(lines)):
    l = lines[j].strip()
    if l.startswith('.Lfunc_end'):
        break
    if not l or l.startswith(';') or (l.startswith('.') and not l.startswith('.LBB')):
        continue
    buf.append(l.split(';')[0].strip())
print(len(buf), 'instructions')
for k in range(0, len(buf), win):
    c = collections.Counter()
    for l in buf[k:k + win]:
        op = l.split()[0]
        if op.startswith('.LBB'): c['lbl'] += 1
        elif op.startswith('scratch_load'): c['sload'] += 1
        elif op.startswith('scratch_store'): c['sstore'] += 1
        elif op.startswith('global_load'): c['gload'] += 1
        elif op.startswith('global_store'): c['gstore'] += 1
        elif op.startswith('ds_'): c['ds'] += 1
        elif op.startswith('v_mfma'): c['mfma'] += 1
        elif op.startswith('v_readlane') or op.startswith('v_writelane'): c['lane'] += 1
        elif op.startswith('s_waitcnt'): c['wait'] += 1
        elif op.startswith('s_barrier'): c['bar'] += 1
    print(k, dict(c))
