"""Static instruction mix of the production trace kernel by loop depth (development aid).
    tools/kinfo.sh && python tools/isa_depth.py /tmp/prod.s
Depth 1 = the main loop (once per query round), deeper = the plane / sphere / node / leaf loops."""
import collections, re, sys
blocks = []; cur = None
for ln in open(sys.argv[1] if len(sys.argv) > 1 else '/tmp/prod.s'):
    m = re.match(r'^(\.LBB\d+_\d+):', ln)
    if m:
        cur = {'name': m.group(1), 'valu': 0, 'lane': 0, 'salu': 0, 'mem': 0, 'depth': 0}
        blocks.append(cur)
    if cur is None:
        cur = {'name': 'entry', 'valu': 0, 'lane': 0, 'salu': 0, 'mem': 0, 'depth': 0}; blocks.append(cur)
    m = re.search(r'in Loop: Header=(BB\d+_\d+) Depth=(\d+)', ln)
    if m and cur['depth'] == 0: cur['depth'] = int(m.group(2))
    m = re.search(r'Loop Header: Depth=(\d+)', ln)
    if m: cur['depth'] = int(m.group(1))
    m = re.match(r'\s+([a-z_0-9]+)', ln)
    if not m: continue
    op = m.group(1)
    if op.startswith(('v_readlane', 'v_writelane')): cur['lane'] += 1
    elif op.startswith('v_'): cur['valu'] += 1
    elif op.startswith('s_'): cur['salu'] += 1
    elif op.startswith(('global_', 'ds_', 'scratch_', 'flat_', 'buffer_')): cur['mem'] += 1
by = collections.defaultdict(lambda: [0, 0, 0, 0, 0])
for b in blocks:
    d = by[b['depth']]; d[0] += b['valu']; d[1] += b['lane']; d[2] += b['salu']; d[3] += b['mem']; d[4] += 1
print("depth: VALU  v_read/writelane  SALU  mem  blocks")
for k in sorted(by): print(k, by[k])
if len(sys.argv) > 2:
    for b in blocks:
        if b['lane'] >= int(sys.argv[2]): print(b)
