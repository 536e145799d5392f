#!/bin/bash
# Per-kernel register / LDS / scratch figures straight from the compiler (-Rpass-analysis=kernel-resource-usage),
# i.e. the numbers occupancy follows.  Usage: bash scripts/kernel_resources.sh > profiles/rNN_kernel_resources.txt
set -e
cd "$(dirname "$0")/../synference_amd/csrc"
FLAGS="-O3 -std=c++17 --offload-arch=gfx950 -I../../include --cuda-device-only -Rpass-analysis=kernel-resource-usage"
fmt() {
  grep -E "Function Name|VGPRs:|AGPRs|SGPRs:|Spill|LDS Size|ScratchSize|Occupancy" |
    sed 's/.*remark: [^ ]* //; s/\[-Rpass-analysis=kernel-resource-usage\]//' | paste - - - - - - - - - |
    sed 's/Function Name: //; s/[ \t]\+/ /g'   # (names stay mangled: template arguments are readable as Li<N>E / Lb<0|1>E)
}
for f in sf_maf16.hip sf_kernels.hip sf_train.hip sf_mlp.hip sf_post.hip; do
  echo "## $f"; hipcc $FLAGS -c $f -o /dev/null 2>&1 | fmt
done
for k in 0 1; do for h in 2; do
  echo "## sf_inst.hip SF_KIND=$k SF_HT=$h (the tile count of H = 50)"; hipcc $FLAGS -DSF_KIND=$k -DSF_HT=$h -c sf_inst.hip -o /dev/null 2>&1 | fmt
done; done
echo "## sf_train_inst.hip SF_HT=2"; hipcc $FLAGS -DSF_HT=2 -c sf_train_inst.hip -o /dev/null 2>&1 | fmt
