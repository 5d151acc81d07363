#!/bin/bash
# Verification build of `lit`'s whole-bundle shadow decisions (rt_core.h: bundle_decide): compiled with the hooks of
# tools/decide_check_hooks.h (-DMCRT_KERNEL_HOOKS) every record's rays are traced even when the bundle was decided, and a
# decision the traced rays contradict is printed and counted.  The product library contains none of this.  usage: tools/decide_check.sh build   (here, no GPU needed)
#                                            tools/decide_check.sh run [first_seed count]   (on the GPU box)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
C=$R/minecraftskin_raytracer_amd/csrc
if [ "$1" = build ]; then
  mkdir -p $R/variants
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -I$R/include -I$C -I$R/tools '-DMCRT_KERNEL_HOOKS="decide_check_hooks.h"' \
    $C/render_kernels.hip $C/api.cpp $C/flatten.cpp $C/scene_builder.cpp $C/png_writer.cpp -o $R/variants/decide_check.so -lpthread
  exit $?
fi
first=${2:-900000}; count=${3:-2000}
export MCRT_LIB=$R/variants/decide_check.so MCRT_GRAPH=0
for mode in "" bundle wide; do
  timeout -k 10 900 python3 $R/tools/gpu_fuzz.py $first $count $mode 2>&1 | grep -a "DECIDE_CHECK\|fuzz:" | python3 -c "
import sys, re
tot = [0, 0, 0, 0]; bad = []
for line in sys.stdin:
    m = re.search(r'records: (\d+) undecided, (\d+) decided all-shadowed, (\d+) decided all-lit, (\d+) CONTRADICTED', line)
    if m: tot = [a + int(b) for a, b in zip(tot, m.groups())]
    elif 'decided' in line: bad.append(line.strip())
    elif line.startswith('fuzz:'): print(line.strip())
print('records: %d undecided, %d decided all-shadowed, %d decided all-lit, %d contradicted' % tuple(tot))
for b in bad[:20]: print(b)
sys.exit(1 if tot[3] else 0)" || exit 1
done
