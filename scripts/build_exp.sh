# experimental builds for scripts/exp_value.sh / shard_ab.py (BPVO_AB_LIB): bash scripts/build_exp.sh name "-DMACRO=.. -D.." [name2 "flags2" ...]
# (the product's own build function with extra flags; the objects of every variant go to a directory of their own)
mkdir -p bpvo_amd/csrc/exp
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  python3 -c "
import os, shlex, sys
sys.path.insert(0, os.getcwd())
import __graft_entry__ as g
g.build_hip(force=True, extra_flags=shlex.split('''$flags'''), out=os.path.join(g.CSRC, 'exp', 'libbpvo_hip_$name.so'))
print('built $name ($flags)')
" &
done
wait
