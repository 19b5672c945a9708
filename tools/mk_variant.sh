#!/bin/bash
# A variant build of ONE translation unit linked with the in-tree objects -> variants/libmmdm_<name>.so (select it with MMDM_LIB; tools/ab_lib.sh).
# usage: tools/mk_variant.sh <name> <source.hip> [extra hipcc flags ...]      e.g. tools/mk_variant.sh noepi gemm_fp8p.hip -DFP8P_EPI=0
set -e
NAME=$1; SRC=$2; shift 2
R=$(cd $(dirname $0)/.. && pwd); C=$R/mixermdm_amd/csrc
mkdir -p $R/variants /tmp/mmdm_var
OBJ=/tmp/mmdm_var/${SRC%.hip}_$NAME.o
FLAGS=$(python3 -c "
import sys; sys.path.insert(0, '$R')
from mixermdm_amd import build as b
u = [x for x in b.UNITS if x[0] == '$SRC'][0]
print(' '.join(b.BASE_FLAGS + list(u[2])))")
/opt/rocm/bin/hipcc $FLAGS "$@" -c $C/$SRC -o $OBJ 2>&1 | grep -v "not a recognized feature" || true
OBJS=$(python3 -c "
import sys; sys.path.insert(0, '$R')
from mixermdm_amd import build as b
print(' '.join('$OBJ' if (s, o) == ('$SRC', '${SRC%.hip}.o') else '$C/' + o for s, o, _ in b.UNITS))")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=$C/libmmdm.map -o $R/variants/libmmdm_$NAME.so $OBJS
ls -la $R/variants/libmmdm_$NAME.so
