# per-function VGPRs / scratch of one .hip source (kernels AND the noinline device functions they call): bash scripts/fn_usage.sh kernels_gn_team.hip [extra flags]
cd "$(dirname "$0")/../bpvo_amd/csrc"
src=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wall -Wno-unused-function "$@" -S --cuda-device-only "$src" -o /tmp/fn_usage.s 2>/tmp/fn_usage.err || { cat /tmp/fn_usage.err; exit 1; }
grep -E "warning|error" /tmp/fn_usage.err | head
python3 - <<'PY'
import re, subprocess
txt = open('/tmp/fn_usage.s').read()
blocks = re.split(r'^\s*\.type\s+(\S+),@function', txt, flags=re.M)
out = []
for i in range(1, len(blocks), 2):
    body = blocks[i + 1]
    g = lambda k: (re.search(r'; %s: (\d+)' % k, body) or [None, '?'])[1]
    out.append((blocks[i], g('NumVgprs'), g('ScratchSize'), g('Occupancy'), g('LDSByteSize')))
dem = subprocess.run(['c++filt'], input='\n'.join(o[0] for o in out), capture_output=True, text=True).stdout.splitlines()
for o, d in zip(out, dem):
    d = re.sub(r'\(.*', '', d).replace('bpvo_hip::', '').replace('void ', '').replace('bool ', '')
    print('%-84s vgpr %4s scratch %5s occ %2s lds %s' % (d[:84], o[1], o[2], o[3], o[4]))
PY
