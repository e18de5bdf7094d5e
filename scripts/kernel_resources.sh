#!/bin/bash
# Per-kernel resource table of a built libhalart*.so: VGPRs, AGPRs, SGPRs, scratch (private segment) bytes per lane, static LDS, occupancy limit.
# usage: scripts/kernel_resources.sh [path/to/libhalart.so] [symbol filter regex]
LIB=${1:-hala-renderer_amd/lib/libhalart.so}
PAT=${2:-.}
TMP=$(mktemp -d)
# the fat binary holds one code object per translation unit: extract them all
python3 - "$LIB" "$TMP" <<'PY'
import sys, re
data = open(sys.argv[1], 'rb').read()
magic = b'__CLANG_OFFLOAD_BUNDLE__'
pos = 0; n = 0
while True:
    i = data.find(magic, pos)
    if i < 0: break
    import struct
    cnt = struct.unpack_from('<Q', data, i + 24)[0]
    off = i + 32
    for _ in range(cnt):
        eo, es, ts = struct.unpack_from('<QQQ', data, off); off += 24
        triple = data[off:off + ts].decode(); off += ts
        if 'gfx' in triple and es:
            open(f"{sys.argv[2]}/co{n}.elf", 'wb').write(data[i + eo:i + eo + es]); n += 1
    pos = i + 32
PY
for f in $TMP/co*.elf; do
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes $f 2>/dev/null
done | python3 -c "
import sys, re
txt = sys.stdin.read()
pat = re.compile(r'$PAT')
rows = []
for blk in txt.split('- .agpr_count:')[1:]:
    def g(k):
        m = re.search(r'\.' + k + r':\s+(\S+)', blk); return m.group(1) if m else '?'
    name = g('name')
    import subprocess
    rows.append((name, '.agpr_count: ' + blk.split('\n')[0].strip(), g('vgpr_count'), g('sgpr_count'), g('private_segment_fixed_size'), g('group_segment_fixed_size'), g('vgpr_spill_count'), g('sgpr_spill_count'), g('max_flat_workgroup_size')))
import shutil
for r in rows:
    dem = subprocess.run(['/usr/bin/c++filt', r[0]], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r'\(.*', '', dem)
    if not pat.search(dem): continue
    print(f'{dem:70s} vgpr {r[2]:>4s} sgpr {r[3]:>4s} scratch {r[4]:>5s} B  lds {r[5]:>6s} B  vspill {r[6]:>3s} sspill {r[7]:>3s} wg {r[8]}')
"
rm -rf $TMP
