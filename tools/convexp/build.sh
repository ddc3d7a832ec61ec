#!/bin/bash
# Builds the experiment libraries next to this script: libconv8p.so (4-phase im2col-by-DMA kernel), libconv8p_w.so (pixel
# window kernel = the product kernel), the stamp build and any ablation named on the command line (NOMFMA NODMA ...).
D=$(cd "$(dirname "$0")" && pwd)
cd "$D" || exit 1
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared"
rm -f libconv8p_w[A-Z]*.so
hipcc $F conv8p.hip -o libconv8p.so 2>&1 | grep -E "error"
hipcc $F -DUSE_CONV8W conv8p.hip -o libconv8p_w.so -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|VGPRs:|Scratch|LDS Size" | sort -u
hipcc $F -DUSE_CONV8W -DSGO_CONV8_STAMPS conv8p.hip -o libconv8p_stamps.so 2>&1 | grep -E "error"
for v in "$@"; do
  defs=""; for d in ${v//+/ }; do defs="$defs -DSGW_$d"; done
  hipcc $F -DUSE_CONV8W $defs conv8p.hip -o libconv8p_w$v.so 2>&1 | grep -E "error"
done
ls -1 "$D"/*.so
